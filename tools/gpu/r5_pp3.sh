cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pp
mkdir -p $O
timeout -k 10 300 python tools/bench_conv.py --bs 64 --only l3_3x3,l4_3x3 --ab "NNL_CONV_WINO+NNL_WINO2_PP+NNL_WINO_BALANCE+NNL_WINO2_POS+NNL_WINO2_PRIO=3+1+0+0+0,3+1+0+0+4,3+1+0+0+8,3+1+0+0+16,3+1+0+0+12,3+1+0+0+20,3+1+0+0+24,3+1+0+0+28" > $O/pp_abl_bs64.log 2>&1; echo "rc=$?"; grep -v "s2 \|wgrad\|amdgpu" $O/pp_abl_bs64.log
