#!/bin/bash
# after the 512-thread policy for 64-bit keys: the 8-bit tests, the policy A/B on more datasets, the matrix rows it changes with their PMC passes
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03w3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -m gpu -x -q -k "8bit or radix8 or eight or staying or kernel_variants or harness or 512" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit 1; }
run() { python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))"; }
{
echo "== uint64 / int64 keys without payload, RSX_R8_WIDE=0 against the policy (-1)"
for v in "u64range --dtype uint64 --dataset Range" "u64inv --dtype uint64 --dataset InvertedRange" "i64uni --dtype int64 --dataset RandomDistributed" "u64rand --dtype uint64 --dataset Random" "u64_2p24 --dtype uint64 --dataset RandomDistributed --log2-keys 24" "u64_2p26 --dtype uint64 --dataset RandomDistributed --log2-keys 26"; do
  set -- $v; tag=$1; shift
  for w in 0 -1 0 -1; do echo "[$tag] wide=$w  $(RSX_R8_WIDE=$w run "$@")"; done
done
} 2>&1 | tee $O/ab_wide_policy.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/profiles/pmc_traffic.json $R/$O/pmc_traffic.json
tag=u64_8bit
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pf_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --radix-bits 8 --dtype uint64 --dataset RandomDistributed > $R/$O/pf_$tag.json 2> $R/$O/pf_$tag.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pw_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --radix-bits 8 --dtype uint64 --dataset RandomDistributed > /dev/null 2> $R/$O/pw_$tag.err
wl=$(python3 -c "import json; print(json.load(open('$R/$O/pf_$tag.json'))['config']['workload'])")
python3 $R/tools/pmc_summarize.py "$(find $R/$O/pf_$tag -name '*counter_collection.csv' | head -1)" "$(find $R/$O/pw_$tag -name '*counter_collection.csv' | head -1)" "$wl" $R/$O/pmc_traffic.json > $R/$O/pmc_$tag.txt 2>&1
rm -rf $R/$O/pf_$tag $R/$O/pw_$tag
cd $R; cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --no-cpu-baseline --radix-bits 8 --steps 5 --dtype uint64 --dataset RandomDistributed > $O/row_u64_8bit.json 2>/dev/null; cut -c1-300 $O/row_u64_8bit.json
tail -5 $O/pmc_$tag.txt
