#!/bin/bash
# kernel_stats.sh OUTDIR TAG [bench args...]: rocprofv3 --kernel-trace --stats of one bench.py workload -> OUTDIR/TAG_kernel_stats.csv
# (per-kernel calls / average ns), OUTDIR/TAG_bench.json (the line of that same run), and — with PASSES=1 — OUTDIR/TAG_per_pass.txt:
# the reorder launches of the LAST timed sort one by one (which pass of a sorted input is the slow one).
set -o pipefail
O=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-verify "$@" > $O/${TAG}_bench.json 2> $O/${TAG}.err || exit 2
find $O/st_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
if [ -n "$PASSES" ]; then
  python3 - "$(find $O/st_$TAG -name '*kernel_trace.csv' | head -1)" > $O/${TAG}_per_pass.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "reorder" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -8:]
for r in last:
    print("%8.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:110]))
PY
fi
rm -rf $O/st_$TAG
python3 - $O/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print("%-100s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"].split("(")[0][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
