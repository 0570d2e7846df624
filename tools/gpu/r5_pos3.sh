cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5pos
mkdir -p $O
for bs in 8 32; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_pos$bs -o p -- python3 tools/bench_conv.py --bs $bs --only l2_3x3,l3_3x3,l4_3x3 --ab "NNL_CONV_WINO+NNL_WINO2_POS=3+1,0+0" > $O/prof_pos$bs.log 2>&1; echo "rc=$?"
for f in $(find $O/prof_pos$bs -name "*.db" | head -1); do python tools/stats_csv.py $f $O/pos${bs}_kernel_stats.csv; done
head -12 $O/pos${bs}_kernel_stats.csv | cut -c1-230
done
