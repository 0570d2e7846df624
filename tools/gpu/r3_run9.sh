# round 3, GPU call 9: BK=16 vs BK=32 for the 64x64 taps kernel after the prologue / k-loop clean-ups (per layer, 64 images)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_IGEMM_BK32=-1,0,1 2>&1 | grep -v "wgrad\|amdgpu.ids"
