#!/bin/bash
# round 4, call m: the pool of sort engines (waves sorted side by side) — tests, soak, forced-exchange lines: doubling groups on one engine vs B deeper on four
set -o pipefail
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_harness.py -x -q -m gpu -k "pool or peer_store or ranks_on_one_gpu or sharded_harness" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout -k 10 400 python tools/soak_sharded.py 300 21 > $O/soak_sharded.txt 2>&1; tail -2 $O/soak_sharded.txt
for cfg in "waves-p2p 1 3" "waves-p2p 4 5" "waves-p2p 3 5" "waves-p2p 2 4" "waves 1 3" "waves 4 5"; do
  set -- $cfg
  for rb in 4 8; do
    RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$1 timeout -k 10 300 python bench.py --gpus 1 --log2-keys 27 --sort-engines $2 --partition-bits $3 --radix-bits $rb --no-events --no-cpu-baseline --steps 20 --warmup 3 > $O/forced_$1_e$2_b$3_r$rb.json 2> $O/forced_$1_e$2_b$3_r$rb.err || echo "FAILED $cfg $rb"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04m/forced_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d['config']['parallelism'][:110])
    except Exception as e:
        print(f, 'unreadable', e)
PY
