#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03e; mkdir -p $O
(cd /tmp && rocprofv3 -L > $OLDPWD/$O/counters_all.txt 2>&1); grep -i -E "TCC_EA0|UTCL|TLB|TCC_.*STALL|TCC_REQ|TCC_HIT|TCC_MISS|TCP_.*MISS|MALL|TCC_BUBBLE|TCC_TAG" $O/counters_all.txt | sort -u | head -150 > $O/counters_mem.txt; wc -l $O/counters_all.txt $O/counters_mem.txt
python -m pytest tests -m gpu -x -q --durations=8 -k "peer_store_exchange or config4" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -14 $O/tests.log
hipcc --offload-arch=gfx950 -O3 tools/probes/lookback_probe.hip -o $O/lookback_probe || exit 1
for cfg in "16384 108" "65536 27"; do for x in 0 1; do echo "== lookback_probe $cfg per_xcd=$x"; timeout -k 10 120 $O/lookback_probe $cfg $x; done; done 2>&1 | tee $O/lookback.txt
rm -f $O/lookback_probe
echo "== modes: u32+payload 8-bit, 6 engines"; MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=8 MODE_SORTS=12 python tools/mode_probe.py 2>&1 | tee $O/modes_u32pay8.txt
echo "== modes: u64+payload 4-bit, 5 engines"; MODE_ENGINES=5 MODE_PAYLOAD=1 MODE_DTYPE=uint64 MODE_SORTS=8 python tools/mode_probe.py 2>&1 | tee $O/modes_u64pay4.txt
