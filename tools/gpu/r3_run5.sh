# round 3, GPU call 5: per-workgroup timestamps of the conv kernel + the LM / RetinaNet curve fixtures (G14, G15)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so NNL_IGEMM_BALANCE=0 timeout -k 10 200 python tools/conv_timing.py > gpurun_out/r3_conv_timing.log 2>&1; cat gpurun_out/r3_conv_timing.log
timeout -k 10 900 python -m pytest tests -m gpu -q -k "g14 or g15 or g13b" > gpurun_out/r3_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t5.log
grep -n "rel |\|losses  \|gradient norms\|passed\|failed\|Error" gpurun_out/r3_t5.log | cut -c1-260
