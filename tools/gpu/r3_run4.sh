# round 3, GPU call 4: k-slicing sweep of the balanced conv schedule at 8 / 16 / 32 images (workspace re-queried per plan)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 100 python -m pytest tests/test_optim_gpu.py -q > gpurun_out/r3_t4.log 2>&1; tail -2 gpurun_out/r3_t4.log
for bs in 8 16 32; do
timeout -k 10 400 python tools/bench_conv.py --bs $bs --ab NNL_IGEMM_PLAN_S=0,1,2,3,4,6,8,12,16 > gpurun_out/r3_conv_bs${bs}_S.log 2>&1; grep -v wgrad gpurun_out/r3_conv_bs${bs}_S.log | cut -c1-230
done
timeout -k 10 300 python tools/bench_conv.py --bs 8 --ab NNL_WGRAD_SPLITS=0,2,4,8,16,32 > gpurun_out/r3_conv_bs8_W.log 2>&1; grep "wgrad\|layer" gpurun_out/r3_conv_bs8_W.log | cut -c1-200
