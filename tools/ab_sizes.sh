#!/bin/bash
# ab_sizes.sh ENVVAR "sizes": ms/sort with ENVVAR=0 and =1, interleaved, two rounds per size (same box, same call)
VAR=$1; shift; SIZES=${@:-"24 26 28"}
for p in $SIZES; do
  line="2^$p:"
  for round in 1 2; do for v in 0 1; do
    ms=$(env $VAR=$v python bench.py --log2-keys $p --steps $([ $p -le 24 ] && echo 100 || echo 30) --warmup 5 --no-events --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    line="$line  $VAR=$v $ms"
  done; done
  echo "$line"
done
