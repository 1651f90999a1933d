#!/bin/bash
# Interleaved A/B of one environment switch on ONE box: tools/ab_env.sh VAR "v1 v2 ..." [rounds] [bench args]
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq $ROUNDS); do for v in $VALS; do
env $VAR=$v python bench.py --no-cpu-baseline --steps 20 "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('r$r $VAR=$v', d['config']['workload'][:24], d['value'], d['ms_per_step'], 'reorder', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done; done
