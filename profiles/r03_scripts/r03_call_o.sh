#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "8bit or harness" 2>&1 | tail -4 | tee $O/tests.txt
echo "== uint32+payload, 8-bit: (key, payload) as one 64-bit element (RSX_R8_PACKED=1, default) against apart (=0)"
for args in "--payload" "--payload --dtype int32 --dataset RandomDistributed" "--payload --dataset Zeros" "--payload --dataset Range"; do
  line="[$args]"
  for round in 1 2 3; do for v in 0 1; do
    r=$(RSX_R8_PACKED=$v python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms (reorder %.4f = %.3f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))")
    line="$line  packed=$v $r"
  done; done
  echo "$line"
done 2>&1 | tee $O/ab_packed.txt
echo "== histogram kernels with and without the XCD tile mapping (ms per launch)"
for v in 1 0; do for bits in 4 8; do
  RSX_XCD_REMAP=$v python bench.py --no-cpu-baseline --no-verify --radix-bits $bits --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('XCD_REMAP=$v bits=$bits: %.3f ms/sort, histogram %.4f ms, reorder %.4f ms' % (d['ms_per_step'], d['phases_ms_per_launch']['histogram'], d['phases_ms_per_launch']['reorder']))"
done; done 2>&1 | tee $O/hist_remap.txt
