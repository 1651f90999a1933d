#!/bin/bash
# what bounds histogram8_kernel: probe builds with the LDS atomics replaced by plain stores / removed (counts are wrong: --no-verify), histogram ms per launch
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03z3; mkdir -p $O
hrun() { python bench.py --no-cpu-baseline --no-verify --radix-bits 8 --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms histo %.4f' % (d['ms_per_step'], d['phases_ms_per_launch']['histogram']))"; }
{
for v in "u32" "u64 --dtype uint64 --dataset RandomDistributed"; do
  set -- $v; tag=$1; shift
  for round in 1 2; do for lib in product h8store h8loads; do
    if [ $lib = product ]; then unset RSX_LIB; else export RSX_LIB=tools/_variants/libradixsort_hip_$lib.so; fi
    echo "[$tag] $lib  $(hrun "$@")"
  done; done; unset RSX_LIB
done
} 2>&1 | tee $O/h8_probe.txt
