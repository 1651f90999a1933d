#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/final3
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/final3/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -9 gpurun_out/final3/gpu_tests.log
STEPS="1 2" bash tools/final_profiles_r03.sh 2>&1 | tail -8
