# round 5 final evidence on ONE build: the plain bench line, rocprofv3 kernel stats of the headline / RetinaNet / LM, the committed counter passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5final
mkdir -p $O
timeout -k 10 900 python3 bench.py > $O/bench_final.json.log 2> $O/bench_final.err; echo "bench rc=$?"
bash tools/gpu/r5_counters.sh > $O/counters.log 2>&1; echo "counters rc=$?"; tail -3 $O/counters.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_rl -o p -- python3 bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --no-counters --configs retinanet > $O/bench_ret.log 2>$O/prof_ret.err; echo "rocprof retina rc=$?"
for f in $(find $O/prof_rl -name "*.db" | head -1); do python tools/stats_csv.py $f $O/r5_retinanet_kernel_stats.csv; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_lm -o p -- python3 tools/bench_heads.py lm --steps 10 > $O/prof_lm.log 2>&1; echo "lm rocprof rc=$?"
for f in $(find $O/prof_lm -name "*.db" | head -1); do python tools/stats_csv.py $f $O/r5_lm_kernel_stats.csv; done
find $O gpurun_out/r5p -name "*.db" -delete 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_final.json.log') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'busy', d['roofline']['mfma_busy'])
print('proxy', json.dumps(d.get('strong_scaling_proxy'))[:700])
for k,c in d['configs'].items(): print(k, c.get('ms_per_step'), c.get('value'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
