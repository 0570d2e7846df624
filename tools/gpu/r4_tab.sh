cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_tabular.py tests/test_step_loss_parity.py tests/test_fcnet_fit_curves.py -x -q -m gpu -k "conv2d_fwd_bwd or tabular or g16 or g2 or g3 or g4 or step or fit" > gpurun_out/r4tab/test.log 2>&1; rc=$?; tail -5 gpurun_out/r4tab/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
for k in 1 0 1 0; do NNL_IGEMM_KTAIL=$k timeout -k 10 200 python tools/bench_heads.py tabular --steps 200 --graphs 2>/dev/null | tail -1 | cut -c1-140; done
