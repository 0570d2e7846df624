# the plain bench line of the final tree + small-batch kernel stats of the final build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5final2
mkdir -p $O
timeout -k 10 900 python3 bench.py > $O/bench_final.json.log 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_final.json.log') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'busy', d['roofline']['mfma_busy'], d['roofline']['counters_source'][:60])
p=d.get('strong_scaling_proxy'); print('proxy', {k:(v['hipgraph_ms_per_step'], v['hipgraph_t64_over_t']) for k,v in p.items() if isinstance(v, dict)})
for k,c in d['configs'].items(): print(k, c.get('ms_per_step'), c.get('value'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
bash tools/gpu/r5_small_batch_stats.sh r5f > $O/sb.log 2>&1; grep -c wrote $O/sb.log
