# round 3, GPU call 10: wgrad kernel timelines (timing build) + conv tests with the new BK rule
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/stamps_wgrad
NNL_TIMING_DUMP=gpurun_out/stamps_wgrad NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so timeout -k 10 200 python tools/conv_timing.py wgrad 2>&1 | grep -v amdgpu.ids
NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so timeout -k 10 200 python tools/conv_timing.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python -m pytest tests/test_conv_gpu.py -m gpu -q 2>&1 | tail -2
