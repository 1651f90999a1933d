#!/bin/bash
# Round-3 judged artefacts, regenerated on the GPU box -> gpurun_out/final3/ (copy the summaries to profiles/ as r03_*).
#  1. default bench line (N=1, BASELINE config 2), with cpu_baseline and bit-exact verification
#  2. rocprofv3 --kernel-trace --stats of the SAME workload with 300 timed steps
#  3. FETCH_SIZE / WRITE_SIZE passes (separate runs) for EVERY row of the workload matrix, 4-bit and 8-bit -> pmc_traffic.json
#  4. the workload matrix itself, 4-bit and 8-bit (after 3, so that every line carries its roofline.traffic)
# Steps can be selected: STEPS="1 2 3a 3b 4" (default all).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final3
STEPS=${STEPS:-"1 2 3a 3b 4"}
mkdir -p $O
has () { [[ " $STEPS " == *" $1 "* ]]; }
if has 1; then
  python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
  echo "bench done"
fi
if has 2; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 300 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err || exit 2
  find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
  rm -rf $O/stats
  python3 - <<PY
import csv, json
line = json.load(open("$O/bench_under_rocprof.json"))
rows = {r["Name"]: r for r in csv.DictReader(open("$O/kernel_stats.csv"))}
fused = next(v for k, v in rows.items() if "reorder_kernel<unsigned int, 256, 16, false, true, false," in k)
plain = next(v for k, v in rows.items() if "reorder_kernel<unsigned int, 256, 16, false, false, false," in k)
fa, pa = float(fused["AverageNs"]) * 1e-6, float(plain["AverageNs"]) * 1e-6
w = (7 * fa + pa) / 8
open("$O/rocprof_vs_events.txt", "w").write(
    "rocprofv3 --stats over %s fused + %s plain launches: fused %.4f ms, plain %.4f ms, launch-weighted %.4f ms = %.4f of 8 TB/s\n" % (fused["Calls"], plain["Calls"], fa, pa, w, 2 * 2**28 * 4 / (w * 1e-3) / 8e12)
    + "bench line of the same run (HIP events): avg_launch_ms %.4f, frac %.4f\n" % (line["roofline"]["avg_launch_ms"], line["roofline"]["frac"]))
print(open("$O/rocprof_vs_events.txt").read())
PY
  echo "kernel stats done"
fi
[ -f $R/profiles/pmc_traffic.json ] && [ ! -f $O/pmc_traffic.json ] && cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
pmc () {   # tag, bench args...  (the workload string is taken from the bench line of the FETCH pass itself)
  local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > $O/pf_$tag.json 2> $O/pf_$tag.err || return 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > /dev/null 2> $O/pw_$tag.err || return 4
  local wl=$(python3 -c "import json; print(json.load(open('$O/pf_$tag.json'))['config']['workload'])")
  python3 $R/tools/pmc_summarize.py "$(find $O/pf_$tag -name '*counter_collection.csv' | head -1)" "$(find $O/pw_$tag -name '*counter_collection.csv' | head -1)" "$wl" $O/pmc_traffic.json > $O/pmc_$tag.txt 2>&1
  rm -rf $O/pf_$tag $O/pw_$tag $O/pf_$tag.json
  echo "pmc $tag done: $wl"
}
rows () {   # suffix, extra bench args (digit width)
  local sfx=$1; shift
  for ds in Random Zeros Range InvertedRange RandomDistributed; do pmc u32_${ds}_$sfx --dataset $ds "$@" || return 3; done
  pmc i32_uni_$sfx --dtype int32 --dataset RandomDistributed "$@" || return 3
  pmc u32pay_$sfx --payload "$@" || return 3
  pmc u64_$sfx --dtype uint64 --dataset RandomDistributed "$@" || return 3
  pmc u64pay_$sfx --dtype uint64 --payload --dataset RandomDistributed "$@" || return 3
  pmc i64payz_$sfx --dtype int64 --payload --dataset Zeros "$@" || return 3
}
if has 3a; then rows 4bit || exit 3; fi
if has 3b; then rows 8bit --radix-bits 8 || exit 3; fi
if has 4; then
  cd $R
  cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json       # (on the box only: bench.py reads roofline.traffic from there)
  bash tools/run_matrix.sh $O/matrix_4bit.jsonl > $O/matrix_4bit.txt 2>&1
  sed 's/python bench.py/python bench.py --radix-bits 8/' tools/run_matrix.sh > /tmp/run_matrix8.sh && bash /tmp/run_matrix8.sh $O/matrix_8bit.jsonl > $O/matrix_8bit.txt 2>&1
  echo "matrix done"; cat $O/matrix_4bit.txt $O/matrix_8bit.txt
fi
ls $O
