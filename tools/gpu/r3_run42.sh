#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_wino.py --bs 64 > gpurun_out/r3_wino.log 2>&1; echo "rc=$?"
cat gpurun_out/r3_wino.log | cut -c1-700
