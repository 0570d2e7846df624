cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5bn
for bs in 64 8; do timeout -k 10 200 python tools/bench_bn.py --bs $bs 2>&1 | grep -v amdgpu | tee gpurun_out/r5bn/bn_bw_bs$bs.log; done
