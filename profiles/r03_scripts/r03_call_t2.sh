#!/bin/bash
# 4-bit reorder of 64-bit keys + payload on 512 x 8 (policy) against 256 x 16: tests, then config 3 and friends interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03t2; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -m gpu -x -q -k "either_workgroup or payload or harness or int64 or uint64" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
run() { python bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))"; }
{
echo "== 4-bit, 64-bit keys + payload: RSX_REORDER_WIDE=0 (256 x 16) against -1 (policy: 512 x 8)"
for v in "u64pay --dtype uint64 --dataset RandomDistributed --payload" "i64payz --dtype int64 --dataset Zeros --payload" "u64payrange --dtype uint64 --dataset Range --payload" "u64pay_2p26 --dtype uint64 --dataset RandomDistributed --payload --log2-keys 26" "u64pay_2p24 --dtype uint64 --dataset RandomDistributed --payload --log2-keys 24"; do
  set -- $v; tag=$1; shift
  for w in 0 -1 0 -1; do echo "[$tag] wide=$w  $(RSX_REORDER_WIDE=$w run "$@")"; done
done
} 2>&1 | tee $O/ab_4bit_wide_policy.txt
