#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/final3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final3/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/final3/gpu_tests.log
STEPS="3b" bash tools/final_profiles_r03.sh 2>&1 | grep -E "pmc|rror"
cp gpurun_out/final3/pmc_traffic.json profiles/pmc_traffic.json
sed 's/python bench.py/python bench.py --radix-bits 8/' tools/run_matrix.sh > /tmp/run_matrix8.sh && bash /tmp/run_matrix8.sh gpurun_out/final3/matrix_8bit.jsonl > gpurun_out/final3/matrix_8bit.txt 2>&1; cat gpurun_out/final3/matrix_8bit.txt
