#!/bin/bash
# ab_lib.sh VARIANT_SO -- "bench args" ["bench args" ...]: the product library against a variant build, interleaved, two rounds each
V=$1; shift; [ "$1" = "--" ] && shift
for args in "$@"; do
  line="[$args]"
  for round in $(seq 1 ${ROUNDS:-2}); do for lib in product variant; do
    if [ $lib = variant ]; then export RSX_LIB=$V; else unset RSX_LIB; fi
    r=$(python bench.py --no-cpu-baseline --no-verify $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms (reorder %.4f)' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    line="$line  $lib $r"
  done; done
  unset RSX_LIB
  echo "$line"
done
