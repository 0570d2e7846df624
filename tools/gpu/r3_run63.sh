#!/bin/bash
# 2-D Winograd F(2x2,3x3) probe: correctness + per-stage time against the 1-D kernel
set -e
mkdir -p gpurun_out
echo "== 1-D ==" > gpurun_out/r63.log
timeout -k 10 240 python tools/bench_wino.py >> gpurun_out/r63.log 2>&1
echo "== 2-D occ4 ==" >> gpurun_out/r63.log
timeout -k 10 240 python tools/bench_wino.py --two-d >> gpurun_out/r63.log 2>&1
echo "== 2-D occ3 ==" >> gpurun_out/r63.log
NNL_WINO2_OCC=3 timeout -k 10 240 python tools/bench_wino.py --two-d >> gpurun_out/r63.log 2>&1
echo "== 2-D bk32 ==" >> gpurun_out/r63.log
NNL_WINO2_BK=32 timeout -k 10 240 python tools/bench_wino.py --two-d >> gpurun_out/r63.log 2>&1
tail -5 gpurun_out/r63.log
