# round 3, GPU call 26: phases of a timestep of the persistent LSTM forward (timing build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so timeout -k 10 200 python tools/lstm_timing.py 2>&1 | grep -v amdgpu.ids
