cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4tab
rm -rf gpurun_out/r4tab/prof_t; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4tab/prof_t -o t -- python3 tools/bench_heads.py tabular --steps 20 --graphs > /dev/null 2>&1
f=$(find gpurun_out/r4tab/prof_t -name "*kernel_trace.csv" | head -1); cp $f gpurun_out/r4tab/trace_tabular.csv; rm -rf gpurun_out/r4tab/prof_t
