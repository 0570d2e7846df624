cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
timeout -k 10 900 python -m pytest tests/test_tabular.py tests/test_conv_gpu.py tests/test_vision_gpu.py tests/test_e2e_gpu.py -x -q -m gpu -k "not winograd and not wgrad_wino and not g13" > gpurun_out/r4tab/test.log 2>&1; rc=$?; tail -3 gpurun_out/r4tab/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
for k in 1 2 3; do timeout -k 10 200 python tools/bench_heads.py tabular --steps 300 --graphs 2>/dev/null | tail -1 | cut -c1-140; done
