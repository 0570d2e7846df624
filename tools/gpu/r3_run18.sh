# round 3, GPU call 18: per-layer table of the RetinaNet R50-FPN convolutions at 512x512, 16 images
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python tools/bench_conv.py --net r50 --bs 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_conv_r50.log
