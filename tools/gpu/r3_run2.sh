# round 3, GPU call 2: the new fused ops + the parity fixtures that exist so far
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -k "not g14 and not g15" > gpurun_out/r3_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t2.log
tail -25 gpurun_out/r3_t2.log | cut -c1-300
