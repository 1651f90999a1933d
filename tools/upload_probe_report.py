"""Fold a rocprofv3 kernel trace of tools/upload_probe.py into one line per sort."""
import csv
import glob
import sys

root = sys.argv[1]
path = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sorts, cur, prev_end = [], None, None
for s, t, name in rows:
    short = "hist" if "histogram_kernel" in name else "reorder" if "reorder_kernel" in name else "scan" if "scan" in name else "paste" if "paste" in name else name[:12]
    if short == "hist":
        cur = {"gap_us": (s - prev_end) / 1e3 if prev_end else 0.0, "start": s, "k": []}
        sorts.append(cur)
    if cur is not None:
        cur["k"].append((short, (t - s) / 1e3, s))
    prev_end = t
for i, srt in enumerate(sorts):
    re = [d for k, d, _ in srt["k"] if k == "reorder"]
    hi = [d for k, d, _ in srt["k"] if k == "hist"]
    end = max(s + d * 1e3 for _, d, s in srt["k"])
    print(f"sort {i:2d}: idle before {srt['gap_us']:9.1f} us | hist {hi[0]:6.1f} | reorder " + " ".join(f"{d:6.1f}" for d in re) + f" | whole {(end - srt['start']) / 1e3:7.1f} us")
