# round 3, GPU call 6: affine tap mode (no table walk in the conv prologue): parity, timestamps, per-layer A/B at 64 and 8 images
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py tests/test_text.py tests/test_tabular.py -m gpu -q -x > gpurun_out/r3_t6.log 2>&1; tail -3 gpurun_out/r3_t6.log
NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so NNL_IGEMM_BALANCE=0 timeout -k 10 200 python tools/conv_timing.py > gpurun_out/r3_conv_timing2.log 2>&1; cat gpurun_out/r3_conv_timing2.log
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_IGEMM_AFFINE=0,1 > gpurun_out/r3_affine_bs64.log 2>&1; grep -v wgrad gpurun_out/r3_affine_bs64.log
timeout -k 10 300 python tools/bench_conv.py --bs 8 --ab NNL_IGEMM_AFFINE=0,1 > gpurun_out/r3_affine_bs8.log 2>&1; grep -v wgrad gpurun_out/r3_affine_bs8.log
