cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pp
mkdir -p $O
NNL_WINO2_PP=1 timeout -k 10 400 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "wino" > $O/pp_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/pp_tests.log
for bs in 64 8; do
timeout -k 10 300 python tools/bench_conv.py --bs $bs --only l1_3x3,l2_3x3,l3_3x3,l4_3x3 --ab "NNL_WINO2_PP+NNL_WINO2_POS=0+-1,1+0,0+0" > $O/pp_bs$bs.log 2>&1; echo "rc=$?"; grep -v "s2 \|wgrad\|amdgpu" $O/pp_bs$bs.log
done
timeout -k 10 300 python tools/bench_conv.py --bs 64 --only l1_3x3,l2_3x3,l3_3x3,l4_3x3 --ab "NNL_WINO2_PP+NNL_WINO_PLAN_KS=1+1,1+2,1+4,0+1,0+2,0+4" > $O/pp_ks_bs64.log 2>&1; echo "rc=$?"; grep -v "s2 \|wgrad\|amdgpu" $O/pp_ks_bs64.log
