#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03t; mkdir -p $O
export MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=8 MODE_SORTS=10
for kb in 0 16; do echo "-- uint64+payload 8-bit, RSX_R8_EXTRA_LDS_KB=$kb" | tee -a $O/modes_u64pay8.txt; MODE_DTYPE=uint64 MODE_SORTS=6 RSX_R8_EXTRA_LDS_KB=$kb python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_u64pay8.txt; done
for kb in 0 8; do echo "-- uint32+payload 8-bit (packed), RSX_R8_EXTRA_LDS_KB=$kb" | tee -a $O/modes_u32pay8.txt; RSX_R8_EXTRA_LDS_KB=$kb python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_u32pay8.txt; done
echo "-- uint32+payload 8-bit, unpacked, 0" | tee -a $O/modes_u32pay8.txt; RSX_R8_PACKED=0 RSX_R8_EXTRA_LDS_KB=0 python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_u32pay8.txt
