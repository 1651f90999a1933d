#!/bin/bash
# final artefacts of round 3 on the final build: soak, then bench + rocprof stats + PMC for every matrix row + the matrices
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final3; mkdir -p $O
timeout -k 10 500 python tools/soak.py 300 31 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak.txt
bash tools/final_profiles_r03.sh 2>&1 | grep -v "^pmc " 
