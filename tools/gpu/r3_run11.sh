# round 3, GPU call 11: wgrad — workgroups per CU (split count) A/B at 64 images
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python tools/bench_conv.py --bs 64 --ab NNL_WGRAD_WGPCU10=0,10,15,20,30,40,50 2>&1 | grep "wgrad\|layer\|total"
