#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03w; mkdir -p $O
export MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=4 MODE_SORTS=10
for kb in 0 16; do echo "-- uint32+payload 4-bit, RSX_REORDER_EXTRA_LDS_KB=$kb" | tee -a $O/modes_u32pay4.txt; RSX_REORDER_EXTRA_LDS_KB=$kb python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_u32pay4.txt; done
echo "-- uint32 keys 8-bit, policy" | tee -a $O/modes_u32_8.txt; MODE_PAYLOAD=0 MODE_BITS=8 python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_u32_8.txt
echo "-- uint32 keys 8-bit, 5 workgroups per CU" | tee -a $O/modes_u32_8.txt; RSX_R8_EXTRA_LDS_KB=0 MODE_PAYLOAD=0 MODE_BITS=8 python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_u32_8.txt
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['verified'])"
