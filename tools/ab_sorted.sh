#!/bin/bash
# ab_sorted.sh VARIANT_SO [rounds] [bench args]: ms per sort of the product and a variant build, interleaved, printed sorted (the runs are bimodal, 2-3 % apart)
V=$1; R=${2:-6}; shift 2
p=""; v=""
for i in $(seq 1 $R); do
  unset RSX_LIB; p="$p $(python bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
  export RSX_LIB=$V; v="$v $(python bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
done
unset RSX_LIB
echo "product: $(echo $p | tr ' ' '\n' | sort -n | tr '\n' ' ')"
echo "$(basename $V): $(echo $v | tr ' ' '\n' | sort -n | tr '\n' ' ')"
