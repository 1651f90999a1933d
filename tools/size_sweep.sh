#!/bin/bash
# Mkeys/s over sizes 2^16..2^30 (uint32 Random, device-resident) -> $1
OUT=${1:-gpurun_out/size_sweep.jsonl}
: > $OUT
for lg in 16 18 20 22 24 26 28 30; do
  steps=20; [ $lg -ge 30 ] && steps=5
  python bench.py --steps $steps --warmup 2 --no-cpu-baseline --log2-keys $lg >> $OUT 2>> ${OUT%.jsonl}.err || echo "{\"failed\": $lg}" >> $OUT
done
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l)
    if 'failed' in d: print(d); continue
    print("2^%-3d %10.1f Mkeys/s  %9.4f ms/sort  reorder %.4f ms (%4.1f%% of peak)" % (d['config']['keys_per_gpu'].bit_length()-1, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], 100*d['roofline']['frac']))
PY
