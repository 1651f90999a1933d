#!/bin/bash
# keys and payloads in one LDS image (RSX_R8_MERGED_PAYLOAD=1, product) against separate trips (variant), 8-bit scatter with a separate payload array; tests first
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -m gpu -x -q -k "8bit or radix8 or eight or staying or kernel_variants or harness" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
{
ROUNDS=3 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_unmerged.so -- "--radix-bits 8 --dtype uint64 --dataset RandomDistributed --payload" "--radix-bits 8 --dtype int64 --dataset Zeros --payload" "--radix-bits 8 --dtype uint64 --dataset Range --payload"
echo "== uint32 keys with the payload kept apart (RSX_R8_PACKED=0)"
RSX_R8_PACKED=0 ROUNDS=2 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_unmerged.so -- "--radix-bits 8 --payload"
echo "== merged image at three workgroups per CU? (RSX_R8_EXTRA_LDS_KB: 0 = two per CU as the image stands; the variant's 16 = its two per CU, 0 = its three)"
for kb in 0 16; do RSX_R8_EXTRA_LDS_KB=$kb ROUNDS=1 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_unmerged.so -- "--radix-bits 8 --dtype uint64 --dataset RandomDistributed --payload"; done
} 2>&1 | tee $O/ab_merged_payload.txt
