#!/bin/bash
# 8-bit scatter as a staying grid with next-tile prefetch: parity test first, then A/B per variant and workgroups per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "staying_grid or kernel_variants or radix8 or eight_bit or 8bit" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
run() { # label, env..., -- bench args
  python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))"
}
{
echo "== 8-bit scatter: RSX_R8_STAY = workgroups per CU of the staying grid (0 = one workgroup per tile), RSX_R8_EXTRA_LDS_KB left to the policy unless named"
for v in "u32pay --payload" "u64 --dtype uint64 --dataset RandomDistributed" "u64pay --dtype uint64 --dataset RandomDistributed --payload" "u32"; do
  set -- $v; tag=$1; shift
  for stay in 0 2 3 0 2; do
    echo "[$tag] stay=$stay  $(RSX_R8_STAY=$stay run "$@")"
  done
done
echo "== staying grid with LDS that allows more workgroups per CU (extra 0) — does the prefetch stand in for occupancy?"
for v in "u32pay --payload" "u64 --dtype uint64 --dataset RandomDistributed" "u32"; do
  set -- $v; tag=$1; shift
  for stay in 3 4; do
    echo "[$tag] extra=0 stay=$stay  $(RSX_R8_EXTRA_LDS_KB=0 RSX_R8_STAY=$stay run "$@")"
  done
done
} 2>&1 | tee $O/ab_stay_tickets.txt
