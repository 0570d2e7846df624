cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
timeout -k 10 900 python -m pytest tests/test_bn_gpu.py tests/test_tabular.py tests/test_step_loss_parity.py tests/test_fcnet_fit_curves.py tests/test_e2e_gpu.py tests/test_graph_gpu.py tests/test_syncbn_gpu.py -x -q -m gpu > gpurun_out/r4tab/test.log 2>&1; rc=$?; tail -3 gpurun_out/r4tab/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
for k in 1 0 1 0; do NNL_BN_FUSE_FINALIZE=$k timeout -k 10 200 python tools/bench_heads.py tabular --steps 300 --graphs 2>/dev/null | tail -1 | cut -c1-140; done
for k in 1 0; do NNL_BN_FUSE_FINALIZE=$k timeout -k 10 200 python tools/bench_small_batch.py --bs 8 2>/dev/null | tail -2 | cut -c1-200; done
