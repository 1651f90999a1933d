#!/bin/bash
# mode_counters.sh OUTDIR: what differs between a slow and a fast engine of one process?  tools/mode_probe.py (several engines alive at
# once, same device-resident input, the scatter kernels of each timed by rocprofv3's kernel trace) under one --pmc pass per counter
# group; every pass is a process of its own (its own allocations, so its own slow and fast engines) and is reported by
# tools/mode_counters_report.py: per engine the mean scatter-launch time next to the counters of those same launches.
set -o pipefail
O=$(realpath -m $1); shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MODE_ENGINES=${MODE_ENGINES:-6} MODE_PAYLOAD=${MODE_PAYLOAD:-1} MODE_BITS=${MODE_BITS:-8} MODE_SORTS=${MODE_SORTS:-3}
i=0
for group in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUBBLE_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
             "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/pass$i -- python3 $R/tools/mode_probe.py > $O/pass$i.txt 2> $O/pass$i.err || { echo "pass $i failed"; tail -3 $O/pass$i.err; continue; }
  python3 $R/tools/mode_counters_report.py "$(find $O/pass$i -name '*counter_collection.csv' | head -1)" "$(find $O/pass$i -name '*kernel_trace.csv' | head -1)" $MODE_ENGINES ${MODE_KERNEL:-reorder8} | tee $O/report$i.txt
  rm -rf $O/pass$i
done
