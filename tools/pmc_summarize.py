#!/usr/bin/env python3
"""Fold rocprofv3 PMC CSVs (separate FETCH_SIZE / WRITE_SIZE passes) into per-kernel HBM
bytes per launch and write profiles/pmc_traffic.json for bench.py's roofline.traffic.

    python tools/pmc_summarize.py <fetch_csv> <write_csv> "<workload string>" [out.json]

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for streaming stores.
"""
import collections
import csv
import hashlib
import json
import os
import sys


def kernel_source_digest() -> str:
    """sha256[:16] over the kernel sources (radix-sort_amd/csrc, sorted by name): bench.py prints a committed traffic figure only while
    this still matches (same function there) — a kernel that changed without a new PMC pass reports traffic null, not a stale number."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radix-sort_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode() + b"\0" + open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, workload = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    fetch, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    write, nw = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {
            "launches_sampled": [nf.get(k, 0), nw.get(k, 0)],
            "FETCH_SIZE_KiB_raw": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
            "hbm_read_bytes": int(2 * f_kib * 1024), "hbm_write_bytes": int(w_kib * 1024),
            "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024),
        }
    try:
        doc = json.load(open(out))
    except (OSError, ValueError):
        doc = {}
    # the graded kernel = the reorder variant launched most often (the fused look-ahead one in rsx_sort)
    cands = [v for k, v in kernels.items() if "reorder_kernel" in k or "reorder8_kernel" in k]
    reorder = max(cands, key=lambda v: v["launches_sampled"][0]) if cands else None
    doc[workload] = {
        "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB -> bytes, WRITE_SIZE exact",
        "reorder_hbm_bytes_per_launch": reorder["hbm_bytes_per_launch"] if reorder else None,
        "kernel_source_digest": kernel_source_digest(),
        "kernels": kernels,
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc[workload], indent=1))


if __name__ == "__main__":
    main()
