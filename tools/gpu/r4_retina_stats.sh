cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4ret
rm -rf gpurun_out/r4ret/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r4ret/prof -o r -- python3 tools/bench_heads.py retina --steps 8 > gpurun_out/r4ret/bench.log 2> gpurun_out/r4ret/err.log; echo rc=$?
for f in $(find gpurun_out/r4ret/prof -name "*.db" | head -1); do python tools/stats_csv.py $f gpurun_out/r4ret/kernel_stats.csv; done
find gpurun_out/r4ret -name "*.db" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4ret/kernel_stats.csv')))
steps=14   # 3 warm-up + 8 timed + 3 profiled
tot=sum(float(r['TotalDurationUs']) for r in rows)
conv=('igemm_','wino','splitk_reduce','slab_reduce','weight_transpose')
c=sum(float(r['TotalDurationUs']) for r in rows if any(k in r['Name'] for k in conv))
print('total kernel ms/step %.3f conv family %.3f' % (tot/steps/1e3, c/steps/1e3))
for r in rows:
    if not any(k in r['Name'] for k in conv):
        ms=float(r['TotalDurationUs'])/steps/1e3
        if ms>0.04: print('%-110s calls/step %6.1f avg %8.2f us  ms/step %.3f' % (r['Name'][:110], float(r['Calls'])/steps, float(r['AverageUs']), ms))
PY
