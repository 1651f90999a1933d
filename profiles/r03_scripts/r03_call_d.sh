#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "peer or straight_to_destination or config4 or fused_scan or timeout or refuses_tables or graph_capture or two_engines" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
export RSX_REORDER8_V=1
echo "== v1: product (row image in round 1) vs r8pad (padded image in round 1)"
ROUNDS=2 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_r8pad.so -- "--radix-bits 8 --steps 10" "--radix-bits 8 --steps 10 --dataset Range" "--radix-bits 8 --steps 10 --dataset Zeros" "--radix-bits 8 --steps 10 --payload" "--radix-bits 8 --steps 5 --dtype uint64 --dataset RandomDistributed" "--radix-bits 8 --steps 5 --dtype uint64 --payload --dataset RandomDistributed" 2>&1 | tee $O/ab_r8pad.txt
unset RSX_REORDER8_V
echo "== 4-bit headline: product (no trailing barrier in the raking scan) vs trail (round 2's barrier)"
ROUNDS=3 bash tools/ab_lib.sh tools/_variants/libradixsort_hip_trail.so -- "--steps 20" "--steps 10 --payload" "--steps 5 --dtype uint64 --payload --dataset RandomDistributed" 2>&1 | tee $O/ab_trail.txt
for v in 1 2; do for ds in Random Range; do
  RSX_REORDER8_V=$v bash tools/pmc_sq.sh r8v${v}_$ds --radix-bits 8 --dataset $ds 2>&1 | tee -a $O/sq8.txt
done; done
rm -rf gpurun_out/sq_r8v*
