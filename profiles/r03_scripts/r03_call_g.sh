#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03g; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/probes/l2_atomics_probe.hip -o $O/l2a && { $O/l2a 16384 2048 512; $O/l2a 4096 256 256; $O/l2a 65536 32768 64; } 2>&1 | tee $O/l2_atomics.txt; rm -f $O/l2a
for n in 268435456 265269248 250000000 201326592; do
  echo "== u32+payload 8-bit, n=$n, 6 engines"; MODE_N=$n MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=8 MODE_SORTS=10 python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_n.txt
done
echo "== u32+payload 4-bit n=2^28 / 250e6"; for n in 268435456 250000000; do MODE_N=$n MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_SORTS=10 python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_n4.txt; done
echo "== u32 keys-only 8-bit n=2^28 / 250e6"; for n in 268435456 250000000; do MODE_N=$n MODE_ENGINES=6 MODE_BITS=8 MODE_SORTS=10 python tools/mode_probe.py 2>&1 | grep engine | tee -a $O/modes_n8k.txt; done
