cd $GRAFT_REPO_ROOT
O=gpurun_out/r5dp
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_dp_replay_two_ranks_gpu.py tests/test_graph_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/tests.log | tail -30
timeout -k 10 400 python tools/cpu_threads_probe.py --legs 16,32,64,128 --bs 64 > $O/cpu_probe.log 2>&1; echo "probe rc=$?"; cat $O/cpu_probe.log
