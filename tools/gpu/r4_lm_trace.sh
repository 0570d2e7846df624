cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4lm
rm -rf gpurun_out/r4lm/prof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4lm/prof -o lm -- python3 tools/bench_heads.py lm --steps 4 > gpurun_out/r4lm/bench.log 2> gpurun_out/r4lm/err.log; echo rc=$?
f=$(find gpurun_out/r4lm/prof -name "*kernel_trace.csv" | head -1); cp $f gpurun_out/r4lm/trace_lm.csv; rm -rf gpurun_out/r4lm/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4lm/trace_lm.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[r for r in rows if (int(r['End_Timestamp'])-int(r['Start_Timestamp']))>150000]
last=big[-40:]
for r in last:
    print('%8.1f us grid %8s wg %4s  %s' % ((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X', r.get('Grid_Size','?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size','?')), r['Kernel_Name'][:80]))
PY
