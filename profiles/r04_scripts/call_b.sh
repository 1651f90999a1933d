#!/bin/bash
# round 4, call b: the parametrised pipelined exchange (msd path) — GPU tests of the new code, then forced-exchange lines (one rank talking
# to itself through real RCCL / through peer stores) without events in the timed region
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_gpu_msd.py tests/test_gpu_sharded.py tests/test_gpu_experiments.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for cfg in "waves 27 4" "waves-p2p 27 4" "waves 28 4" "waves-p2p 28 4" "waves 27 3" "waves-p2p 27 3"; do
  set -- $cfg
  RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$1 timeout -k 10 300 python bench.py --gpus 1 --log2-keys $2 --partition-bits $3 --no-events --no-cpu-baseline --steps 20 --warmup 3 > $O/forced_$1_2p$2_b$3.json 2> $O/forced_$1_2p$2_b$3.err || echo "FAILED $cfg"
  RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$1 timeout -k 10 300 python bench.py --gpus 1 --log2-keys $2 --partition-bits $3 --radix-bits 8 --no-events --no-cpu-baseline --steps 20 --warmup 3 > $O/forced_$1_2p$2_b$3_r8.json 2> $O/forced_$1_2p$2_b$3_r8.err || echo "FAILED r8 $cfg"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04b/forced_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d['config']['parallelism'], d.get('sharded_phases_ms'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
