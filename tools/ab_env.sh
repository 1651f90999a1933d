#!/bin/bash
# ab_env.sh VAR=VALUE [rounds] [bench args]: ms per sort of the product with and without one environment setting, interleaved, printed sorted
# (runs of one build are bimodal, 2-3 % apart: compare the sorted lists, not single runs)
S=$1; R=${2:-6}; shift 2
one () { python bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
p=""; v=""
for i in $(seq 1 $R); do
  p="$p $(one "$@")"
  v="$v $(env $S python bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
done
echo "[$*] default: $(echo $p | tr ' ' '\n' | sort -n | tr '\n' ' ')"
echo "[$*] $S: $(echo $v | tr ' ' '\n' | sort -n | tr '\n' ' ')"
