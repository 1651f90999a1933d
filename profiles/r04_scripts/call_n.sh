#!/bin/bash
# round 4, call n: the C++ driver with a pool of sort engines (one rank, real RCCL, 2^27 keys)
set -o pipefail
O=gpurun_out/r04n; mkdir -p $O
B=radix-sort_amd/host/bin/rsx_tests
for cfg in "peer-stores 1 3" "peer-stores 4 5" "peer-stores 2 4" "all-to-all 1 3" "all-to-all 4 5"; do
  set -- $cfg
  for rb in 4 8; do
    timeout -k 10 300 $B --sharded --comm rccl --exchange $1 --sort-engines $2 --partition-bits $3 --radix-bits $rb --num-elements 134217728 --skip-cpu --perf-csv-to-stdout > $O/cpp_$1_e$2_b$3_r$rb.log 2>&1 || echo "FAILED cpp $cfg $rb"
    echo "C++ one rank, RCCL, $1, engines $2, B $3, radix bits $rb:"; grep -E "^134217728,uint32_t,Random" $O/cpp_$1_e$2_b$3_r$rb.log | cut -d, -f1-3,8
  done
done
