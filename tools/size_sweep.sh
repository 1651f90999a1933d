#!/bin/bash
# size_sweep.sh [out.jsonl] [sizes...]: one bench line per input size (no HIP events inside the timed region: at small sizes they cost as much as a launch)
OUT=${1:-gpurun_out/size_sweep.jsonl}; shift
SIZES=${@:-"8 10 12 14 16 18 20 22 24 26 28"}
: > "$OUT"
for p in $SIZES; do
  python bench.py $BENCH_ARGS --log2-keys $p --steps $([ $p -le 22 ] && echo 200 || echo 30) --warmup 5 --no-events --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT"
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"{d['config']['workload']:<48} {d['ms_per_step']:9.4f} ms/sort {d['value']:10.1f} Mkeys/s")
PY
