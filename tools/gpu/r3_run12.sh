# round 3, GPU call 12: the plain bench line (as the driver runs it) + rocprofv3 kernel stats of the headline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_a.json.log 2> gpurun_out/r3_bench_a.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3_bench_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_a.json.log').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['by_kind'])
print('proxy', {k:(v['ms_per_step'],v['hipgraph_ms_per_step']) for k,v in d.get('strong_scaling_proxy',{}).items() if isinstance(v,dict)})
for k,v in d['configs'].items(): print(k, v['ms_per_step'], v.get('median_ms_per_step'), v['value'], v['roofline'].get('frac'), v.get('eager_step'), v.get('hipgraph_step'))
print('cpu', d.get('cpu_baseline'))
PY
