#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the rows whose scatter kernel changed shape this round (which kernel runs by default, and its average duration)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03ks; mkdir -p $O
bash $R/tools/kernel_stats.sh $O u64_8bit --radix-bits 8 --dtype uint64 --dataset RandomDistributed --steps 20 && \
bash $R/tools/kernel_stats.sh $O u64pay_4bit --dtype uint64 --dataset RandomDistributed --payload --steps 10 && \
bash $R/tools/kernel_stats.sh $O u32pay_8bit --radix-bits 8 --payload --steps 20 && \
bash $R/tools/kernel_stats.sh $O u32pay_4bit --payload --steps 20
