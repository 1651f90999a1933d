#!/bin/bash
# do the key LOADS of the tiles in flight take L2 room from the scatter's open output lines?  non-temporal key loads (variant) against plain ones (product),
# 8-bit scatter of uint64 / uint32 keys at the policy's workgroups per CU and at what LDS admits
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03z2; mkdir -p $O
{
for kb in -1 0; do
  echo "== RSX_R8_EXTRA_LDS_KB=$kb"
  RSX_R8_EXTRA_LDS_KB=$kb ROUNDS=2 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_ntloads.so -- "--radix-bits 8 --dtype uint64 --dataset RandomDistributed" "--radix-bits 8" "--radix-bits 8 --dataset Range"
done
} 2>&1 | tee $O/ab_ntloads_8bit.txt
