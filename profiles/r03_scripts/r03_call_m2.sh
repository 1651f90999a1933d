#!/bin/bash
# merged key+payload image with key and payload of a slot stored together (variant) against separate trips (product, default)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03m2; mkdir -p $O
RSX_LIB=tools/_variants/libradixsort_hip_merged.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "staying or kernel_variants" > $O/tests_merged.log 2>&1; rc=$?; tail -2 $O/tests_merged.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
ROUNDS=3 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_merged.so -- "--radix-bits 8 --dtype uint64 --dataset RandomDistributed --payload" "--radix-bits 8 --dtype int64 --dataset Zeros --payload" "--radix-bits 8 --dtype uint64 --dataset Range --payload" 2>&1 | tee $O/ab_merged_interleaved.txt
