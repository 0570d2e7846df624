cd $GRAFT_REPO_ROOT
O=gpurun_out/r5wg2d
mkdir -p $O
export NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_abhooks.so NNL_AB=1
timeout -k 10 300 python tools/bench_conv.py --bs 64 --only l2_3x3,l3_3x3,l4_3x3 --ab "NNL_WGRAD_WINO2D_BIG_E6=500,100,100000" > $O/tile_bs64.log 2>&1; echo "rc=$?"; grep "wgrad\|layer\|total" $O/tile_bs64.log
timeout -k 10 300 python tools/bench_conv.py --net r50 --bs 16 --only 3x3_128,3x3_256,3x3_512,fpn_3x3,head_ --ab "NNL_WGRAD_WINO2D_BIG_E6=500,100,100000" > $O/tile_r50.log 2>&1; echo "rc=$?"; grep "wgrad\|layer\|total" $O/tile_r50.log
