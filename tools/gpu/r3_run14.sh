# round 3, GPU call 14: the whole -m gpu suite + smoke, as the driver runs them
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/r3_t14.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3_t14.log | cut -c1-300
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r3_smoke.log
