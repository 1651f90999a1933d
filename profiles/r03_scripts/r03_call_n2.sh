#!/bin/bash
# packed uint32 + payload 8-bit scatter: key and payload store of a slot together (product) against all key stores then all payload stores (variant)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03n2; mkdir -p $O
ROUNDS=3 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_splitstores.so -- "--radix-bits 8 --payload" "--radix-bits 8 --payload --dataset Range" "--radix-bits 8 --payload --dataset Zeros" 2>&1 | tee $O/ab_split_stores.txt
