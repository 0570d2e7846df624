cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
timeout -k 10 300 python -m pytest tests/test_tabular.py -x -q -m gpu > gpurun_out/r4tab/test.log 2>&1; rc=$?; tail -3 gpurun_out/r4tab/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
for k in 1 2; do timeout -k 10 200 python tools/bench_heads.py tabular --steps 200 --graphs 2>/dev/null | tail -1 | cut -c1-140; done
rm -rf gpurun_out/r4tab/prof_t; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4tab/prof_t -o t -- python3 tools/bench_heads.py tabular --steps 20 --graphs > /dev/null 2>&1
f=$(find gpurun_out/r4tab/prof_t -name "*kernel_trace.csv" | head -1); cp $f gpurun_out/r4tab/trace_tabular.csv; rm -rf gpurun_out/r4tab/prof_t
