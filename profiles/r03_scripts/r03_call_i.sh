#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$(pwd)/gpurun_out/r03i; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "inline_scan or fused_scan or self_scan or lookahead_histogram" 2>&1 | tail -5 | tee $O/tests.txt
echo "== inline scan off / on, ms per sort (4-bit, uint32 Random)"; bash tools/ab_sizes.sh RSX_INLINE_SCAN 22 23 24 25 26 27 2>&1 | tee $O/ab_inline.txt
echo "== VMM allocation probe"
for mode in 2 1; do RSX_DEBUG_ALLOC=1 RSX_ALLOC_MODE=$mode timeout -k 10 150 python tools/vmm_alloc_probe.py 2>&1 | tail -12 | tee -a $O/vmm_probe.txt; done
echo "== engine spread by allocation mode (u32+payload 8-bit)"
export MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=8 MODE_SORTS=10
for cfg in "0 0" "2 32" "1 32" "2 2" "2 256"; do set -- $cfg
  echo "-- RSX_ALLOC_MODE=$1 chunk=$2 MB" | tee -a $O/modes_alloc.txt; RSX_ALLOC_MODE=$1 RSX_ALLOC_CHUNK_MB=$2 timeout -k 10 240 python tools/mode_probe.py 2>&1 | grep -E "engine|rror" | tee -a $O/modes_alloc.txt
done
