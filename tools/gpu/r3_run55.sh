#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
for bs in 64 32 16; do
timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_WGRAD_WINO=0,1 > gpurun_out/r3_wwg2_bs$bs.log 2>&1
grep -E "3x3 .*wgrad|total" gpurun_out/r3_wwg2_bs$bs.log | grep -v s2
done
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_WGRAD_WINO=0,1 > gpurun_out/r3_wwg2_r50.log 2>&1
grep -E "wgrad|total" gpurun_out/r3_wwg2_r50.log | grep -E "3x3|head|out_|total" | grep -v s2 | head -40
