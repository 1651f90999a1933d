#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03q; mkdir -p $O
echo "== 8-bit scatter: policy (-1) against no extra LDS (0), interleaved, 2 rounds: ms per launch (fraction)"
for w in "u32pay|--payload" "u64|--dtype uint64 --dataset RandomDistributed" "u64pay|--dtype uint64 --payload --dataset RandomDistributed" "i64payz|--dtype int64 --payload --dataset Zeros" "u32payRange|--payload --dataset Range"; do
  tag=${w%%|*}; args=${w#*|}; line="[$tag]"
  for round in 1 2; do for kb in 0 -1; do
    r=$(RSX_R8_EXTRA_LDS_KB=$kb python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))")
    line="$line  [$kb] $r"
  done; done
  echo "$line"
done 2>&1 | tee $O/ab_policy.txt
echo "== 4-bit payload / 64-bit kernels with fewer workgroups per CU? (RSX_REORDER_EXTRA_LDS_KB not implemented: skipped)"
