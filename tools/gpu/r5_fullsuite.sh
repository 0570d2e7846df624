cd $GRAFT_REPO_ROOT
O=gpurun_out/r5suite
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/gpu_tests.log | tail -15
