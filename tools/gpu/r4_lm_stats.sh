cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4lm
rm -rf gpurun_out/r4lm/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4lm/prof -o lm -- python3 tools/bench_heads.py lm --steps 10 > gpurun_out/r4lm/bench.log 2> gpurun_out/r4lm/err.log; echo rc=$?
for f in $(find gpurun_out/r4lm/prof -name "*.db" | head -1); do python tools/stats_csv.py $f gpurun_out/r4lm/kernel_stats.csv; done
find gpurun_out/r4lm -name "*.db" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4lm/kernel_stats.csv')))
steps=16   # 3 warm-up + 10 timed + 3 profiled
tot=sum(float(r['TotalDurationUs']) for r in rows)
print('total kernel ms/step %.3f' % (tot/steps/1e3))
for r in rows[:28]:
    print('%-100s calls %6s avg %8.2f us  ms/step %.3f' % (r['Name'][:100], r['Calls'], float(r['AverageUs']), float(r['TotalDurationUs'])/steps/1e3))
PY
