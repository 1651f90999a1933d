#!/bin/bash
# uint64 + payload 8-bit scatter: 512-thread workgroups together with the merged key+payload image (variant build, RSX_R8_WIDE=1) against the default
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03w4; mkdir -p $O
RSX_R8_WIDE=1 RSX_LIB=tools/_variants/libradixsort_hip_merged.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "512_thread or kernel_variants" > $O/tests.log 2>&1; rc=$?; tail -2 $O/tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
run() { python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))"; }
{
for v in "u64pay --dtype uint64 --dataset RandomDistributed --payload" "i64payz --dtype int64 --dataset Zeros --payload"; do
  set -- $v; tag=$1; shift
  for round in 1 2; do
    echo "[$tag] default            $(run "$@")"
    echo "[$tag] wide               $(RSX_R8_WIDE=1 run "$@")"
    echo "[$tag] wide + merged      $(RSX_R8_WIDE=1 RSX_LIB=tools/_variants/libradixsort_hip_merged.so run "$@")"
  done
done
} 2>&1 | tee $O/ab_wide_merged.txt
