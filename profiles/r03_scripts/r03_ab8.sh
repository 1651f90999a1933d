#!/bin/bash
# reorder8 v1 (two trips) against v2 (one trip), interleaved, 8-bit digits, five workloads: ms per sort and per scatter launch
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in "u32rand|" "u32range|--dataset Range" "u32zeros|--dataset Zeros" "u32pay|--payload" "u64|--dtype uint64 --dataset RandomDistributed" "u64pay|--dtype uint64 --payload --dataset RandomDistributed"; do
  tag=${w%%|*}; args=${w#*|}
  line="[$tag]"
  for round in 1 2; do for v in ${VERSIONS:-1 2}; do
    r=$(RSX_REORDER8_V=$v python bench.py --no-cpu-baseline --radix-bits 8 --steps 10 --warmup 2 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms (reorder %.4f = %.3f; hist %.4f) %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['phases_ms_per_launch']['histogram'], 'ok' if d['config']['verified'] else 'UNVERIFIED'))")
    line="$line  v$v $r"
  done; done
  echo "$line"
done
