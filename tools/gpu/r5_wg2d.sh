cd $GRAFT_REPO_ROOT
O=gpurun_out/r5wg2d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "wgrad" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_conv.py --bs 64 > $O/conv_bs64.log 2>&1; echo "rc=$?"; grep -v amdgpu $O/conv_bs64.log | tail -16
timeout -k 10 300 python tools/bench_conv.py --net r50 --bs 16 --only 3x3,head,fpn > $O/conv_r50.log 2>&1; echo "rc=$?"; grep -v amdgpu $O/conv_r50.log | tail -22
