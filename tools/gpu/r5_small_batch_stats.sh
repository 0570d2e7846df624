# round 5: rocprofv3 kernel stats of the replayed ResNet-34 step at 8 / 16 / 32 images (the strong-scaling proxy), one run per batch size
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r5}
O=gpurun_out/${TAG}sb
mkdir -p $O
for bs in 8 16 32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_bs$bs -o sb -- python3 tools/bench_small_batch.py --bs $bs --steps 40 > $O/bs$bs.json.log 2> $O/rocprof_bs$bs.err; echo "rocprof bs$bs rc=$?"
  for f in $(find $O/prof_bs$bs -name "*.db" | head -1); do python tools/stats_csv.py $f $O/${TAG}_bs${bs}_kernel_stats.csv; done
  find $O/prof_bs$bs -name "*.db" -size +10M -delete
  cat $O/bs$bs.json.log
done
python3 tools/bench_small_batch.py --bs 8,16,32,64 --steps 40 > $O/plain.json.log 2>&1; cat $O/plain.json.log
