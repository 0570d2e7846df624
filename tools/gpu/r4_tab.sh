cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_optim_gpu.py tests/test_fcnet_fit_curves.py tests/test_host_logic.py tests/test_e2e_gpu.py -x -q -m gpu > gpurun_out/r4tab/test.log 2>&1; rc=$?; tail -3 gpurun_out/r4tab/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs collab,tabular 2>gpurun_out/r4tab/bench.err | tail -1 > gpurun_out/r4tab/bench_heads.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4tab/bench_heads.json').read())
for k,c in d['configs'].items():
    print(k, c.get('ms_per_step'), c.get('value'), {x:c.get(x) for x in ('eager_mean_ms','replay_mean_ms','replay_in_fit_loop_mean_ms')}, c.get('error'))
PY
cd tools && timeout -k 10 300 python prof_fit_host.py collab 2>&1 | sed -n 2,14p
