#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python tools/wino_wgrad_debug.py > gpurun_out/r3_wwg_dbg.log 2>&1; echo "rc=$?"
cat gpurun_out/r3_wwg_dbg.log | cut -c1-300
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WGRAD_WINO=0,1 > gpurun_out/r3_wwg_bs64.log 2>&1
grep -E "wgrad|total" gpurun_out/r3_wwg_bs64.log
