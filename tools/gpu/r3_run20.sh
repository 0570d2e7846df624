# round 3, GPU call 20: launch-bound heads after the node-count work (one fill + one segment-sum launch, HIP MSE, scaled sigmoid, Linear->ReLU->BN epilogue statistics)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_collab_gpu.py tests/test_tabular.py tests/test_fcnet_fit_curves.py tests/test_e2e_gpu.py tests/test_graph_gpu.py tests/test_step_loss_parity.py tests/test_vision_gpu.py tests/test_optim_gpu.py -m gpu -q 2>&1 | tail -4 | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs collab,tabular > gpurun_out/r3_bench_heads.json.log 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_heads.json.log').read().strip().splitlines()[-1])
for k,v in d['configs'].items(): print(k, v['ms_per_step'], 'eager', v['eager_step']['ms_per_step'], 'graph', v['hipgraph_step']['ms_per_step'], 'kernel ms', v['roofline']['kernel_ms_per_step'], v['roofline']['by_kind'])
PY
