#!/bin/bash
# 4-bit chain on 512-thread x 8-key workgroups (same 4096-key tiles; probe build) against the product's 256 x 16
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03t; mkdir -p $O
V=tools/_variants/libradixsort_hip_t512k8.so
for a in "--dtype uint64 --dataset RandomDistributed" "--payload"; do
  RSX_LIB=$V python bench.py --no-cpu-baseline --steps 3 $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('probe build verified:', d['config']['verified'], d['ms_per_step'])"
done
ROUNDS=2 bash tools/ab_lib.sh $V -- "--dtype uint64 --dataset RandomDistributed" "--dtype uint64 --dataset RandomDistributed --payload" "" "--payload" "--dtype uint64 --dataset Zeros" 2>&1 | tee $O/ab_4bit_512x8.txt
