#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
STEPS="3a" bash tools/final_profiles_r03.sh 2>&1 | grep -E "pmc|done|rror" 
echo "== non-temporal scatter stores: product vs ntstore"
ROUNDS=2 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_ntstore.so -- "--steps 20" "--steps 10 --payload" 2>&1 | tee gpurun_out/final3/ab_ntstore.txt
timeout -k 10 400 python tools/soak.py 250 7 2>&1 | tail -4 | tee gpurun_out/final3/soak_250.txt
