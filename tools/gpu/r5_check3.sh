cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_abi.py tests/test_bbox_inference.py tests/test_bn_gpu.py tests/test_collab_gpu.py tests/test_device_data.py tests/test_dp_replay_two_ranks_gpu.py tests/test_fcnet_fit_curves.py tests/test_optim_gpu.py tests/test_pool_gpu.py tests/test_step_loss_parity.py tests/test_syncbn_gpu.py tests/test_tabular.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/tests.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
