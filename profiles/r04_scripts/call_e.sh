#!/bin/bash
# round 4, call e: doubling wave groups — sharded / msd / harness GPU tests, then forced-exchange lines (one rank to itself): single waves vs doubling groups
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_harness.py -k "ranks_on_one_gpu or config4 or peer_store or sharded_harness" -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for grp in doubling; do for cfg in "waves 27 3" "waves-p2p 27 3" "waves-p2p 28 4"; do
  set -- $cfg
  for rb in 4 8; do
    RSX_WAVE_GROUPING=$grp RSX_FORCE_EXCHANGE=1 RSX_STRATEGY=$1 timeout -k 10 300 python bench.py --gpus 1 --log2-keys $2 --partition-bits $3 --radix-bits $rb --no-events --no-cpu-baseline --steps 20 --warmup 3 > $O/forced_${grp}_$1_2p$2_b$3_r$rb.json 2> $O/forced_${grp}_$1_2p$2_b$3_r$rb.err || echo "FAILED $grp $cfg $rb"
  done
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04f/forced_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d.get('sharded_phases_ms'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
B=radix-sort_amd/host/bin/rsx_tests
for ex in all-to-all peer-stores; do for rb in 4 8; do
  timeout -k 10 300 $B --sharded --comm rccl --exchange $ex --partition-bits 3 --radix-bits $rb --num-elements 134217728 --skip-cpu --perf-csv-to-stdout > $O/cpp_rccl_${ex}_r$rb.log 2>&1 || echo "FAILED cpp $ex $rb"
  echo "C++ one rank, RCCL, $ex, radix bits $rb"; grep -E "^134217728,uint32_t" $O/cpp_rccl_${ex}_r$rb.log | cut -d, -f1-3,7,8
done; done
