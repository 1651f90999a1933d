#!/bin/bash
# Bench matrix over BASELINE configs 2, 3, 5 (one JSON line per workload) -> $1 (default gpurun_out/matrix.jsonl)
OUT=${1:-gpurun_out/matrix.jsonl}
: > $OUT
for ds in Random Zeros Range InvertedRange RandomDistributed; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dataset $ds >> $OUT 2>> ${OUT%.jsonl}.err || echo "{\"failed\": \"u32 $ds\"}" >> $OUT
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dtype int32 --dataset RandomDistributed >> $OUT 2>> ${OUT%.jsonl}.err
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --payload --dataset Random >> $OUT 2>> ${OUT%.jsonl}.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype uint64 --dataset RandomDistributed >> $OUT 2>> ${OUT%.jsonl}.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype uint64 --payload --dataset RandomDistributed >> $OUT 2>> ${OUT%.jsonl}.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype int64 --payload --dataset Zeros >> $OUT 2>> ${OUT%.jsonl}.err
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l)
    if 'failed' in d: print(d); continue
    print("%-62s %9.1f Mkeys/s  %7.3f ms  reorder %.3f ms %5.1f%%  histo %.3f scan %.4f paste %.4f" % (d["config"]["workload"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], 100*d["roofline"]["frac"], d["phases_ms_per_launch"]["histogram"], d["phases_ms_per_launch"]["scan"], d["phases_ms_per_launch"]["paste"]))
PY
