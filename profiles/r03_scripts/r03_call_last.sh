#!/bin/bash
# last call of round 3: full GPU suite on the final build, then the two 4-bit matrix rows whose reorder kernel changed shape (64-bit keys + payload), with their PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03last; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
pmc () {
  local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pf_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > $R/$O/pf_$tag.json 2> $R/$O/pf_$tag.err || return 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pw_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > /dev/null 2> $R/$O/pw_$tag.err || return 4
  local wl=$(python3 -c "import json; print(json.load(open('$R/$O/pf_$tag.json'))['config']['workload'])")
  python3 $R/tools/pmc_summarize.py "$(find $R/$O/pf_$tag -name '*counter_collection.csv' | head -1)" "$(find $R/$O/pw_$tag -name '*counter_collection.csv' | head -1)" "$wl" $R/$O/pmc_traffic.json > $R/$O/pmc_$tag.txt 2>&1
  rm -rf $R/$O/pf_$tag $R/$O/pw_$tag $R/$O/pf_$tag.json
  echo "pmc $tag done: $wl"
}
pmc u64pay_4bit --dtype uint64 --payload --dataset RandomDistributed
pmc i64payz_4bit --dtype int64 --payload --dataset Zeros
cd $R; cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --no-cpu-baseline --steps 5 --dtype uint64 --payload --dataset RandomDistributed > $O/row_u64pay_4bit.json 2>/dev/null; cut -c1-200 $O/row_u64pay_4bit.json
python bench.py --no-cpu-baseline --steps 5 --dtype int64 --payload --dataset Zeros > $O/row_i64payz_4bit.json 2>/dev/null; cut -c1-200 $O/row_i64payz_4bit.json
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
