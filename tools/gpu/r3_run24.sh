# round 3, GPU call 24: re-calibration of the balanced planner's fix-up cost terms after the chunked slab loads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for bs in 64 16 8; do
echo "== bs $bs: NNL_IGEMM_PLAN_EXTRA"
timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_IGEMM_PLAN_EXTRA=2,1,0 2>&1 | grep "l._3x3 \|total" | grep -v wgrad
echo "== bs $bs: NNL_IGEMM_PLAN_BW"
timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_IGEMM_PLAN_BW=4000,8000,16000 2>&1 | grep "l._3x3 \|total" | grep -v wgrad
done
