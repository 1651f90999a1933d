set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or lookahead or uint32" > gpurun_out/phase_tests.log 2>&1 || { tail -30 gpurun_out/phase_tests.log; exit 1; }
tail -2 gpurun_out/phase_tests.log
for ph in 0 37 150 257 585 1024 1170; do
  echo "== phase $ph" 
  RSX_XCD_PHASE=$ph python tools/mode_probe.py 2>&1 | grep engine
done
echo "== skew 150"
RSX_XCD_SKEW=150 python tools/mode_probe.py 2>&1 | grep engine
