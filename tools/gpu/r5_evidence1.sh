# round 5 evidence, part 1: tests of the latest changes, small-batch proxy + rocprofv3 kernel stats at 8 / 16 / 32 images, LM kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5ev
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_graph_gpu.py tests/test_vision_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/gpu/r5_small_batch_stats.sh r5b > $O/sb.log 2>&1; tail -4 $O/sb.log | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_lm -o p -- python3 tools/bench_heads.py lm --steps 10 > $O/prof_lm.log 2>&1; echo "lm rocprof rc=$?"
for f in $(find $O/prof_lm -name "*.db" | head -1); do python tools/stats_csv.py $f $O/r5_lm_kernel_stats.csv; done
find $O -name "*.db" -delete
head -14 $O/r5_lm_kernel_stats.csv | cut -c1-200
