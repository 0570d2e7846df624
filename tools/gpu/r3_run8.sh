# round 3, GPU call 8: steady-state k-loop rate per tile shape (timing build, plain launches, forced tiles)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for tile in 3 1 2 0; do
echo "=== NNL_IGEMM_TILE=$tile (0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64)"
mkdir -p gpurun_out/stamps_tile$tile
NNL_IGEMM_TILE=$tile NNL_TIMING_DUMP=gpurun_out/stamps_tile$tile NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so NNL_IGEMM_BALANCE=0 timeout -k 10 200 python tools/conv_timing.py 2>&1 | grep -v amdgpu.ids
done
