# driver-like final check: the whole GPU suite in ONE process, smoke(), then the plain bench line with its wall time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5last
mkdir -p $O
S=$(date +%s)
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc $(( $(date +%s) - S )) s"; tail -3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
S=$(date +%s)
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(( $(date +%s) - S )) s"; tail -1 $O/smoke.log
[ $rc -eq 0 ] || exit 1
S=$(date +%s)
timeout -k 10 900 python3 bench.py > $O/bench.json.log 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json.log') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'busy', d['roofline']['mfma_busy'], d['roofline']['counters_source'][:60])
p=d.get('strong_scaling_proxy'); print('proxy', {k:(v['hipgraph_ms_per_step'], v['hipgraph_t64_over_t']) for k,v in p.items() if isinstance(v, dict)})
for k,c in d['configs'].items(): print(k, c.get('ms_per_step'), c.get('value'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
