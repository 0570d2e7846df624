cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5rl
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_detection.py tests/test_text.py tests/test_collab_gpu.py tests/test_tabular.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-counters --configs lm,retinanet > $O/bench_rl.log 2> $O/bench_rl.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_rl.log') if l.startswith('{')][-1])
for k in ('lm','retinanet'):
    c=d['configs'][k]; print(k, c.get('ms_per_step'), c.get('median_ms_per_step'), c.get('value'), {a:b.get('ms_per_step') for a,b in c['roofline'].get('by_kind',{}).items()})
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_ret -o p -- python3 bench.py --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-counters --configs retinanet > /dev/null 2>$O/prof_ret.err; echo "rocprof rc=$?"
for f in $(find $O/prof_ret -name "*.db" | head -1); do python tools/stats_csv.py $f $O/r5_retinanet_kernel_stats.csv; done
find $O -name "*.db" -delete
