cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pp
mkdir -p $O
timeout -k 10 300 python tools/bench_conv.py --bs 64 --only l2_3x3,l3_3x3,l4_3x3 --ab "NNL_CONV_WINO+NNL_WINO2_PP+NNL_WINO_BALANCE+NNL_WINO2_POS=3+1+0+0,3+0+0+0" > $O/pp_nobal_bs64.log 2>&1; echo "rc=$?"; grep -v "s2 \|wgrad\|amdgpu" $O/pp_nobal_bs64.log
timeout -k 10 300 python tools/bench_conv.py --bs 32 --only l2_3x3,l3_3x3,l4_3x3 --ab "NNL_CONV_WINO+NNL_WINO2_PP+NNL_WINO_BALANCE+NNL_WINO2_POS=3+1+0+0,3+0+0+0" > $O/pp_nobal_bs32.log 2>&1; echo "rc=$?"; grep -v "s2 \|wgrad\|amdgpu" $O/pp_nobal_bs32.log
