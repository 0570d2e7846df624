cd $GRAFT_REPO_ROOT
O=gpurun_out/r5suite
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/gpu_tests.log | tail -8
timeout -k 10 120 python tools/prof_retina_aten.py > $O/retina_aten.log 2>&1; grep -v "amdgpu\|Warning\|warn" $O/retina_aten.log | head -45
timeout -k 10 200 python tools/bench_heads.py lm --steps 10 2>&1 | grep -v amdgpu | tail -3
