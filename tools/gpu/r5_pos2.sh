cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pos
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -k "position_split or winograd_2d_debug" > $O/tests2.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests2.log
for bs in 8 16 32; do
  timeout -k 10 300 python tools/bench_conv.py --bs $bs --only l1_3x3,l2_3x3,l3_3x3,l4_3x3 --ab "NNL_CONV_WINO+NNL_WINO2_POS=0+0,1+0,3+0,3+1,3+2,3+4" > $O/pos2_bs$bs.log 2>&1; echo "bs$bs rc=$?"
  grep -v "s2 \|wgrad" $O/pos2_bs$bs.log
done
