set -e
python -m pytest tests -x -q -m gpu > gpurun_out/phase_tests_full.log 2>&1 || { tail -40 gpurun_out/phase_tests_full.log; exit 1; }
tail -2 gpurun_out/phase_tests_full.log
bash tools/ab_env.sh RSX_XCD_PHASE=0 4
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --radix-bits 8
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --dtype uint64 --payload --dataset RandomDistributed
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --dataset Zeros
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --dataset InvertedRange
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --log2-keys 24
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --log2-keys 26
bash tools/ab_env.sh RSX_XCD_PHASE=0 3 --log2-keys 27
