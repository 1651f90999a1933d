#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03x; mkdir -p $O
echo "== 4-bit uint32+payload: policy (three workgroups per CU) against four, interleaved"
for ds in Random Zeros Range RandomDistributed; do line="[$ds]"; for round in 1 2; do for kb in 0 -1; do
  r=$(RSX_REORDER_EXTRA_LDS_KB=$kb python bench.py --no-cpu-baseline --payload --steps 10 --warmup 2 --dataset $ds 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms %.4f (%.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))")
  line="$line  [$kb] $r"; done; done; echo "$line"; done 2>&1 | tee $O/ab_4bit_payload_policy.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
