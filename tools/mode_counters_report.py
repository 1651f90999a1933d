#!/usr/bin/env python3
"""mode_counters_report.py <counter_collection.csv> <kernel_trace.csv> <engines> [kernel substring]
Per engine of tools/mode_probe.py (the dispatches of the chosen kernel in launch order, cut into <engines> equal runs): mean launch
duration from the kernel trace of the SAME process and the mean of every collected counter over those launches."""
import collections
import csv
import sys

cc, kt, engines = sys.argv[1], sys.argv[2], int(sys.argv[3])
pat = sys.argv[4] if len(sys.argv) > 4 else "reorder8"
dur = {}
for r in csv.DictReader(open(kt)):
    if pat in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"]:
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted((d for d in dur if d in vals), key=lambda d: dur[d][0])
per = len(ids) // engines
names = sorted({n for d in ids for n in vals[d]})
print(f"{len(ids)} launches of *{pat}*, {per} per engine; counters are means per launch")
print("engine   launch us  " + "  ".join(f"{n[:34]:>34}" for n in names))
rows = []
for e in range(engines):
    mine = ids[e * per:(e + 1) * per]
    if not mine:
        continue
    us = sum(dur[d][1] for d in mine) / len(mine) / 1e3
    means = [sum(vals[d].get(n, 0.0) for d in mine) / len(mine) for n in names]
    rows.append((us, means))
    print(f"{e:6d}  {us:10.1f}  " + "  ".join(f"{v:34.4g}" for v in means))
if len(rows) >= 2:
    fast, slow = min(rows), max(rows)
    print(f"slowest / fastest engine: time x{slow[0] / fast[0]:.3f};  " + "  ".join(f"{n}: x{(s / f if f else float('nan')):.3f}" for n, s, f in zip(names, slow[1], fast[1])))
