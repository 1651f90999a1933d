#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
STEPS="3b 4" bash tools/final_profiles_r03.sh 2>&1 | grep -E "pmc|done|rror|Mkeys"
