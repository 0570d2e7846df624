# round 5: the position-split 2-D Winograd plan: correctness, then per-layer A/B at 8 / 16 / 32 images
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pos
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -k "position_split or winograd_2d_debug or graph_replay_under_data or two_input or times_out" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 300 python -m pytest tests/test_graph_gpu.py -x -q > $O/tests_graph.log 2>&1; echo "graph tests rc=$?"; tail -5 $O/tests_graph.log
for bs in 8 16 32; do
  timeout -k 10 300 python tools/bench_conv.py --bs $bs --only l1_3x3,l2_3x3,l3_3x3,l4_3x3 --ab "NNL_CONV_WINO+NNL_WINO2_POS=0+0,1+0,3+0,3+1,3+2,3+4,3+8" > $O/pos_bs$bs.log 2>&1; echo "bs$bs rc=$?"
  cat $O/pos_bs$bs.log
done
