#!/bin/bash
# A/B bench of alternative builds: tools/ab.sh "<lib1> <lib2> ..." [bench args]; prints one line per (lib, LA mode)
LIBS=$1; shift
for lib in $LIBS; do for la in 0 1; do
RSX_LIB=$PWD/radix-sort_amd/$lib RSX_LOOKAHEAD=$la python bench.py --no-cpu-baseline --steps 20 "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib LA=$la', d['config']['workload'][:28], d['value'], d['ms_per_step'], 'reorder', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'hist', d['phases_ms_per_launch']['histogram'])"
done; done
