#!/bin/bash
# round 4, call c: the C++ sharded engine (RadixSortMultiGPU) behind rsx_tests — rank threads on the one GPU (loopback), one rank through real RCCL
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
B=radix-sort_amd/host/bin/rsx_tests
timeout -k 10 300 $B -v --ranks 8 --num-elements 300000 --exchange peer-stores --with-permutation > $O/ranks8.log 2>&1; echo "ranks8 p2p rc=$?"
tail -3 $O/ranks8.log
timeout -k 10 300 $B -v --sharded --comm rccl --num-elements 400000 > $O/rccl1.log 2>&1; echo "rccl1 rc=$?"
tail -3 $O/rccl1.log
timeout -k 10 900 python -m pytest tests/test_gpu_harness.py -x -q -m gpu > $O/harness_tests.log 2>&1; echo "rc=$?" >> $O/harness_tests.log
tail -5 $O/harness_tests.log
timeout -k 10 600 $B --ranks 8 --num-elements 134217728 --skip-cpu --perf-to-stdout --exchange peer-stores > $O/ranks8_2p27_p2p.log 2>&1; echo "2p27 p2p rc=$?"
grep -E "total:|validated|FAILED" $O/ranks8_2p27_p2p.log | head -30
