#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel (demangled name, VGPRs, AGPRs, SGPRs,
scratch bytes per lane, static LDS, occupancy).  Usage: kernel_resources.py <remarks.txt> [--scratch-only]"""
import re
import subprocess
import sys


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" ")[0]] = val
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["demangled"] = re.sub(r"^void ", "", n)
    return rows


if __name__ == "__main__":
    rows = parse(sys.argv[1])
    only = "--scratch-only" in sys.argv
    for r in sorted(rows, key=lambda r: r["demangled"]):
        if only and r.get("ScratchSize", "0") == "0":
            continue
        print(f"{r['demangled'][:150]:150s} vgpr {r.get('VGPRs'):>4s} agpr {r.get('AGPRs'):>3s} sgpr {r.get('SGPRs'):>4s} scratch {r.get('ScratchSize'):>4s} lds {r.get('LDS'):>6s} occ {r.get('Occupancy')}")
    print(f"# {len(rows)} kernels, {sum(1 for r in rows if r.get('ScratchSize', '0') != '0')} with scratch")
