# round 3, GPU call 7: per-workgroup timestamps + CU ids of plain and balanced launches (dumped for offline timelines)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/stamps_plain gpurun_out/stamps_bal
NNL_TIMING_DUMP=gpurun_out/stamps_plain NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so NNL_IGEMM_BALANCE=0 timeout -k 10 200 python tools/conv_timing.py > gpurun_out/r3_conv_timing3.log 2>&1; cat gpurun_out/r3_conv_timing3.log
timeout -k 10 600 python -m pytest tests/test_vision_gpu.py -m gpu -q -k g13b > gpurun_out/r3_t7.log 2>&1; tail -2 gpurun_out/r3_t7.log
