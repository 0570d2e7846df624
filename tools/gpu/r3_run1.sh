set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not g13b and not g14 and not g15" > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3_bs8 -o bs8 -- python3 tools/bench_small_batch.py --bs 8 --steps 30 > gpurun_out/r3_bs8.log 2>&1; echo "prof rc=$?"
tail -2 gpurun_out/r3_bs8.log
ls gpurun_out/prof_r3_bs8 | head
for f in $(find gpurun_out/prof_r3_bs8 -name "*.db" | head -1); do python tools/stats_csv.py $f gpurun_out/r3_bs8_kernel_stats.csv; done
find gpurun_out/prof_r3_bs8 -name "*.db" -size +30M -delete
