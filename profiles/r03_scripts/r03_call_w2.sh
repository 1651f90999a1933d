#!/bin/bash
# 8-bit scatter on 512-thread workgroups (8 keys per thread, same tiles): parity first, then A/B per variant
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03w2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "512_thread" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
run() { python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))"; }
{
echo "== RSX_R8_WIDE: 0 = 256 threads x 16 keys (policy workgroups per CU), 1 = 512 threads x 8 keys (two per CU as LDS stands; three for uint32 keys)"
for v in "u64 --dtype uint64 --dataset RandomDistributed" "u32pay --payload" "u64pay --dtype uint64 --dataset RandomDistributed --payload" "u32" "u32range --dataset Range" "u64zeros --dtype uint64 --dataset Zeros"; do
  set -- $v; tag=$1; shift
  for w in 0 1 0 1; do echo "[$tag] wide=$w  $(RSX_R8_WIDE=$w run "$@")"; done
done
echo "== wide with other workgroups per CU (RSX_R8_EXTRA_LDS_KB pads LDS): uint32 at two per CU (16), uint64 at one per CU (32)"
echo "[u32] wide=1 extra=16  $(RSX_R8_WIDE=1 RSX_R8_EXTRA_LDS_KB=16 run)"
echo "[u64] wide=1 extra=32  $(RSX_R8_WIDE=1 RSX_R8_EXTRA_LDS_KB=32 run --dtype uint64 --dataset RandomDistributed)"
} 2>&1 | tee $O/ab_wide.txt
