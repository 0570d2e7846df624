# round 3, GPU call 13: kernel trace of the LM step -> where does the GPU idle?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r3_lm -o lm -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --configs lm > gpurun_out/r3_lm_trace.log 2>&1; echo rc=$?
f=$(find gpurun_out/prof_r3_lm -name "*kernel_trace.csv" | head -1); echo $f; wc -l $f
python tools/gap_analysis.py $f
tail -c 600 gpurun_out/r3_lm_trace.log
find gpurun_out/prof_r3_lm -size +20M -delete
