#!/bin/bash
# Regenerates the judged artefacts of a round on the GPU box: default bench line, rocprofv3 kernel
# stats of the same command, FETCH_SIZE / WRITE_SIZE passes, workload matrix.  Output: gpurun_out/final/
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err || exit 2
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $O/pmc_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $O/pmc_write.err || exit 4
echo "pmc done"
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summarize.py "$F" "$W" "2^28 uint32 Random, 4-bit digits, 8 passes" $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
cd $R && bash tools/run_matrix.sh $O/matrix.jsonl > $O/matrix.txt 2>&1
echo "matrix done"
# the raw traces are large; keep the summaries only
rm -rf $O/stats/*/*kernel_trace.csv $O/pmc_fetch $O/pmc_write
ls -la $O
