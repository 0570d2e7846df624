# round 4: rocprofv3 kernel stats of the headline bench with the staged 2-D Winograd kernel on / off (same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s
for v in 1 0; do
  export NNL_CONV_WINO2S=$v
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r4s/prof_w2s$v -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none > gpurun_out/r4s/bench_w2s$v.json.log 2> gpurun_out/r4s/rocprof_w2s$v.err; echo "rocprof rc=$?"
  for f in $(find gpurun_out/r4s/prof_w2s$v -name "*.db" | head -1); do python tools/stats_csv.py $f gpurun_out/r4s/kernel_stats_w2s$v.csv; done
  find gpurun_out/r4s/prof_w2s$v -name "*.db" -size +10M -delete
done
head -25 gpurun_out/r4s/kernel_stats_w2s1.csv
echo ----
head -25 gpurun_out/r4s/kernel_stats_w2s0.csv
