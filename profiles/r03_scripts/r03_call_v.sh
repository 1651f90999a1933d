#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final3; mkdir -p $O
for st in waves waves-p2p; do RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$st python bench.py --gpus 1 --log2-keys 27 --steps 10 --warmup 2 --cpu-sample-log2 22 > $O/forced_exchange_2p27_$st.json 2> $O/forced_$st.err; echo "rc=$?"; tail -5 $O/forced_$st.err; python -c "import json,sys; d=json.load(open('$O/forced_exchange_2p27_$st.json')); print('$st', d['ms_per_step'], d['sharded_phases_ms'], d['config']['verified'][:40])"; done
