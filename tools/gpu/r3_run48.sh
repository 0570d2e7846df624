#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO_BALANCE=0,1 > gpurun_out/r3_wino_bal.log 2>&1
grep -E "3x3 |total" gpurun_out/r3_wino_bal.log | grep -v "s2" | grep -E "fwd|dgrad|total"
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO_BK=16,32 > gpurun_out/r3_wino_bk.log 2>&1
grep -E "3x3 |total" gpurun_out/r3_wino_bk.log | grep -v "s2" | grep -E "fwd|dgrad|total"
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO_PLAN_S=0,2,4,8 > gpurun_out/r3_wino_S.log 2>&1
grep -E "3x3 |total" gpurun_out/r3_wino_S.log | grep -v "s2" | grep -E "fwd|dgrad|total"
