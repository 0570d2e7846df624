cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
timeout -k 10 900 python -m pytest tests/test_collab_gpu.py tests/test_graph_gpu.py tests/test_e2e_gpu.py tests/test_step_loss_parity.py -x -q -m gpu > gpurun_out/r4tab/test.log 2>&1; rc=$?; tail -3 gpurun_out/r4tab/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
for k in 1 0 1 0; do NNL_EMBDOT_SCAN=$k timeout -k 10 200 python tools/bench_heads.py collab --steps 400 --graphs 2>/dev/null | tail -2 | head -1 | cut -c1-140; done
