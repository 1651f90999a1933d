O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
for w in "u32rand|" "u32range|--dataset Range" "u32pay|--payload" "u64pay|--dtype uint64 --payload --dataset RandomDistributed" "u64|--dtype uint64 --dataset RandomDistributed"; do
  tag=${w%%|*}; args=${w#*|}
  echo "== 8-bit $tag"; PASSES=1 bash tools/kernel_stats.sh $O r8_$tag --radix-bits 8 --steps 10 --warmup 2 $args
done
cat $O/r8_u32range_per_pass.txt
