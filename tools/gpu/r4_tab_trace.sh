cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4tab
for head in tabular collab; do
  rm -rf gpurun_out/r4tab/prof_$head
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4tab/prof_$head -o t -- python3 tools/bench_heads.py $head --steps 20 --graphs > gpurun_out/r4tab/bench_$head.log 2> gpurun_out/r4tab/err_$head.log; echo rc=$?
  f=$(find gpurun_out/r4tab/prof_$head -name "*kernel_trace.csv" | head -1)
  python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('dispatches', len(rows))
PY
  cp $f gpurun_out/r4tab/trace_$head.csv
  rm -rf gpurun_out/r4tab/prof_$head
done
