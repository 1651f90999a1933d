#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$(pwd)/gpurun_out/final3; mkdir -p $O
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
pmc () { local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > $O/pf_$tag.json 2> $O/pf_$tag.err || return 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > /dev/null 2> $O/pw_$tag.err || return 4
  local wl=$(python3 -c "import json; print(json.load(open('$O/pf_$tag.json'))['config']['workload'])")
  python3 $R/tools/pmc_summarize.py "$(find $O/pf_$tag -name '*counter_collection.csv' | head -1)" "$(find $O/pw_$tag -name '*counter_collection.csv' | head -1)" "$wl" $O/pmc_traffic.json > $O/pmc_$tag.txt 2>&1
  rm -rf $O/pf_$tag $O/pw_$tag $O/pf_$tag.json; echo "pmc $tag done: $wl"; }
pmc u64pay_8bit --dtype uint64 --payload --dataset RandomDistributed --radix-bits 8
pmc i64payz_8bit --dtype int64 --payload --dataset Zeros --radix-bits 8
cd $R; cp $O/pmc_traffic.json profiles/pmc_traffic.json
for a in "--dtype uint64 --payload --dataset RandomDistributed" "--dtype int64 --payload --dataset Zeros"; do python bench.py --radix-bits 8 --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | tee -a $O/matrix_8bit_rows_redo.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'])"; done
echo "== forced exchange, one rank, 2^27 keys: RCCL waves vs peer stores (sharded_phases_ms)"
for st in waves waves-p2p; do RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$st python bench.py --gpus 1 --log2-keys 27 --steps 10 --warmup 2 --cpu-sample-log2 22 2>/dev/null | tee $O/forced_exchange_2p27_$st.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$st', d['ms_per_step'], d['sharded_phases_ms'], d['config']['verified'][:40])"; done
