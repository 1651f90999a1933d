#!/bin/bash
# table-row loads no longer waited for ahead of the key loads (4-bit reorder, packed 8-bit scatter): product against the build before, same box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03z; mkdir -p $O
ROUNDS=3 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_prerefactor.so -- "" "--payload" "--dtype uint64 --dataset RandomDistributed" "--dtype uint64 --dataset RandomDistributed --payload" "--dataset Zeros" "--dataset Range" "--log2-keys 24" "--log2-keys 26" "--radix-bits 8 --payload" "--radix-bits 8" 2>&1 | tee $O/ab_early_key_loads.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
