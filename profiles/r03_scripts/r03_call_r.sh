#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03r; mkdir -p $O
echo "== 4-bit reorder with unused dynamic LDS (fewer workgroups per CU): ms per sort / per scatter launch (fraction)"
for w in "u64pay|--dtype uint64 --payload --dataset RandomDistributed --steps 6|0 8 24" "u64|--dtype uint64 --dataset RandomDistributed --steps 6|0 8 24" "u32pay|--payload --steps 10|0 8 16 32" "u32|--steps 20|0 8 16"; do
  IFS='|' read tag args kbs <<< "$w"; line="[$tag]"
  for round in 1 2; do for kb in $kbs; do
    r=$(RSX_REORDER_EXTRA_LDS_KB=$kb python bench.py --no-cpu-baseline --no-verify --warmup 2 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f / %.4f (%.3f)' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))")
    line="$line  +${kb}K $r"
  done; done
  echo "$line"
done 2>&1 | tee $O/extra_lds_4bit.txt
echo "== 8-bit uint32 Range / InvertedRange with fewer workgroups per CU"
for ds in Range InvertedRange; do line="[$ds]"; for kb in 0 4 8; do
  r=$(RSX_R8_EXTRA_LDS_KB=$kb python bench.py --no-cpu-baseline --no-verify --radix-bits 8 --steps 10 --warmup 2 --dataset $ds 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f (%.3f)' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))")
  line="$line  +${kb}K $r"; done; echo "$line"; done 2>&1 | tee $O/extra_lds_range.txt
