#!/bin/bash
# Interleaved A/B bench of alternative builds on ONE box (boxes differ by a few %):
#   tools/ab.sh "<lib1> <lib2> ..." [rounds] [bench args]    prints one line per (round, lib, LA mode)
LIBS=$1; shift
ROUNDS=${1:-1}; shift
for r in $(seq $ROUNDS); do for lib in $LIBS; do for la in ${RSX_AB_MODES:-0 1}; do
RSX_LIB=$PWD/radix-sort_amd/$lib RSX_LOOKAHEAD=$la python bench.py --no-cpu-baseline --steps 20 "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('r$r %-28s LA=$la' % '$lib', d['config']['workload'][:24], d['value'], d['ms_per_step'], 'reorder', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'hist', d['phases_ms_per_launch']['histogram'])"
done; done; done
