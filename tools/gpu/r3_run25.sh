# round 3, GPU call 25: planner fix-up cost with the slab bandwidth term at 16 TB/s: extra iterations 2 / 1 / 0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for bs in 64 32 16 8; do
echo "== bs $bs"
timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_IGEMM_PLAN_EXTRA=2,1,0 2>&1 | grep "total"
done
timeout -k 10 300 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_IGEMM_PLAN_EXTRA=2,1,0 2>&1 | grep "total"
