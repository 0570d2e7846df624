#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_linear.py > gpurun_out/r3_linear.log 2>&1; echo "rc=$?"
cat gpurun_out/r3_linear.log
