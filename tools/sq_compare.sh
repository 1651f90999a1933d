#!/bin/bash
# SQ counters of the fused reorder: product build vs the round-1 reorder kernel (tools/_variants/libradixsort_hip_r1reorder.so)
bash tools/pmc_sq.sh r2product
RSX_LIB=tools/_variants/libradixsort_hip_r1reorder.so bash tools/pmc_sq.sh r1reorder
