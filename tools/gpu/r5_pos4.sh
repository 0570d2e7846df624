cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pos
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_conv_gpu.py -x -q -k "position_split" > $O/tests3.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests3.log
for v in 0 -1; do
  NNL_WINO2_POS=$v timeout -k 10 300 python tools/bench_small_batch.py --bs 8,16,32 --steps 40 > $O/sb_pos$v.log 2>&1; echo "pos=$v rc=$?"
  cut -c1-400 $O/sb_pos$v.log | grep -v amdgpu
done
