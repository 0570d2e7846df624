#!/bin/bash
# k-loop probe with the wave-k-split variant, then the full GPU suite + smoke on the current build
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 120 ./tools/kloop_probe > gpurun_out/r3_kloop_ks.log 2>&1; echo "probe rc=$?"
cat gpurun_out/r3_kloop_ks.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3_t29.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3_t29.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r3_smoke29.log 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/r3_smoke29.log
