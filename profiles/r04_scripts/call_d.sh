#!/bin/bash
# round 4, call d: after the issue-order change of the Python driver (exchange issued two waves ahead from a side stream) — sharded GPU tests, forced-exchange
# lines again, and the same step through the C++ driver (one rank through real RCCL / peer stores to itself)
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_msd.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for cfg in "waves 27 3" "waves-p2p 27 3" "waves 27 4" "waves-p2p 27 4" "waves 28 4" "waves-p2p 28 4"; do
  set -- $cfg
  for rb in 4 8; do
    RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$1 timeout -k 10 300 python bench.py --gpus 1 --log2-keys $2 --partition-bits $3 --radix-bits $rb --no-events --no-cpu-baseline --steps 20 --warmup 3 > $O/forced_$1_2p$2_b$3_r$rb.json 2> $O/forced_$1_2p$2_b$3_r$rb.err || echo "FAILED $cfg $rb"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04d/forced_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d.get('sharded_phases_ms'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
B=radix-sort_amd/host/bin/rsx_tests
for ex in all-to-all peer-stores; do for rb in 4 8; do
  timeout -k 10 300 $B --sharded --comm rccl --exchange $ex --partition-bits 3 --radix-bits $rb --num-elements 134217728 --skip-cpu --perf-csv-to-stdout > $O/cpp_rccl_${ex}_r$rb.log 2>&1 || echo "FAILED cpp $ex $rb"
  echo "C++ one rank, RCCL, $ex, radix bits $rb: NumElements,Datatype,Dataset,avgHistogram,avgScan,avgPaste,avgReorder,avgTotalGPU(step ms),..."; grep -E "^134217728" $O/cpp_rccl_${ex}_r$rb.log | cut -d, -f1-3,7,8 | head -20
done; done
