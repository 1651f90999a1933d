#!/bin/bash
# XCD phase stagger of the 8-bit kernels (tuned for the 4-bit reorder in round 2: tiles_per_xcd / 8): a sweep, scatter ms per launch
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03ph; mkdir -p $O
run() { python bench.py --no-cpu-baseline --no-verify --radix-bits 8 --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  scatter %.4f  histogram %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['phases_ms_per_launch']['histogram']))"; }
{
echo "== RSX_XCD_PHASE (tiles; -1 = default tiles_per_xcd / 8 = 1024 at 2^28 keys), 8-bit digits"
for v in "u32" "u64 --dtype uint64 --dataset RandomDistributed" "u32pay --payload"; do
  set -- $v; tag=$1; shift
  for ph in -1 0 128 512 1024 2048 3072 777 -1; do echo "[$tag] phase=$ph  $(RSX_XCD_PHASE=$ph run "$@")"; done
done
} 2>&1 | tee $O/xcd_phase_8bit.txt
