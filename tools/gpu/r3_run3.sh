# round 3, GPU call 3: conv plan sweeps at 8 images (strong-scaling shapes) and the per-layer table at 64 and 8
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 200 python -m pytest tests/test_optim_gpu.py -q > gpurun_out/r3_t3.log 2>&1; tail -2 gpurun_out/r3_t3.log
timeout -k 10 200 python tools/bench_conv.py --bs 64 > gpurun_out/r3_conv_bs64.log 2>&1; tail -14 gpurun_out/r3_conv_bs64.log
timeout -k 10 200 python tools/bench_conv.py --bs 8 > gpurun_out/r3_conv_bs8.log 2>&1; tail -14 gpurun_out/r3_conv_bs8.log
timeout -k 10 400 python tools/bench_conv.py --bs 8 --ab NNL_IGEMM_PLAN_S=0,2,4,8,9,16,18,32 > gpurun_out/r3_conv_bs8_S.log 2>&1; cat gpurun_out/r3_conv_bs8_S.log
timeout -k 10 300 python tools/bench_conv.py --bs 8 --ab NNL_WGRAD_SPLITS=0,4,8,16,32,64 > gpurun_out/r3_conv_bs8_W.log 2>&1; grep wgrad gpurun_out/r3_conv_bs8_W.log
