cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_detection.py tests/test_text.py tests/test_graph_gpu.py tests/test_vision_gpu.py tests/test_e2e_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/tests.log | tail -4
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_rl -o p -- python3 bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --no-counters --configs retinanet,lm > $O/bench_rl.log 2>$O/prof.err; echo "rocprof rc=$?"
for f in $(find $O/prof_rl -name "*.db" | head -1); do python tools/stats_csv.py $f $O/kernel_stats.csv; done
find $O -name "*.db" -delete
python - <<PY
import json, csv
d=json.loads([l for l in open('$O/bench_rl.log') if l.startswith('{')][-1])
for k,c in d['configs'].items(): print(k, c.get('ms_per_step'), c.get('median_ms_per_step'), c.get('value'))
print('headline', d['value'], d['ms_per_step'])
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
for r in rows:
    nm=r['Name']
    if any(k in nm for k in ('CUDAFunctor_add','sum_tensors','copyBuffer','FillFunctor<float>','fillBuffer','elementwise_kernel_manual','CatArray')):
        print(nm[:80], r['Calls'], r['AverageUs'])
PY
