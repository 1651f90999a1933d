"""Per-dispatch effective shader clock from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` run of
tools/upload_probe.py: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS note)."""
import csv
import glob
import sys

root = sys.argv[1]
cnt = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
rows = []
with open(cnt) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
rows.sort()
line, k = [], 0
for s, t, name, v in rows:
    if "histogram_kernel" in name and line:
        print(f"sort {k:2d}: " + " ".join(line)); line = []; k += 1
    if "reorder_kernel" in name or "histogram_kernel" in name:
        dur_us = (t - s) / 1e3
        line.append(f"{dur_us:6.1f}us@{v / 8 / dur_us / 1e3:4.2f}GHz")
if line:
    print(f"sort {k:2d}: " + " ".join(line))
