cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5quick
mkdir -p $O
timeout -k 10 600 python3 bench.py --no-counters --no-cpu-baseline > $O/bench.json.log 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json.log') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'conv ms', d['roofline'].get('conv_ms_per_step'))
p=d.get('strong_scaling_proxy'); print('proxy', {k:(v['hipgraph_ms_per_step'], v['hipgraph_t64_over_t']) for k,v in p.items() if isinstance(v, dict)})
for k,c in d['configs'].items(): print(k, c.get('ms_per_step'), c.get('value'))
PY
