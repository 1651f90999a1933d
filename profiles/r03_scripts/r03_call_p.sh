#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03p; mkdir -p $O
echo "== fewer 8-bit scatter workgroups per CU (unused dynamic LDS): ms per scatter launch, fraction, HBM write bytes not measured here"
for w in "u32|" "u32pay|--payload" "u64|--dtype uint64 --dataset RandomDistributed" "u64pay|--dtype uint64 --payload --dataset RandomDistributed"; do
  tag=${w%%|*}; args=${w#*|}; line="[$tag]"
  for kb in 0 8 16 24 40; do
    r=$(RSX_R8_EXTRA_LDS_KB=$kb python bench.py --no-cpu-baseline --no-verify --radix-bits 8 --steps 10 --warmup 2 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f (%.3f)' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))")
    line="$line  +${kb}K $r"
  done
  echo "$line"
done 2>&1 | tee $O/extra_lds.txt
