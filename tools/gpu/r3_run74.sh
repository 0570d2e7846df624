#!/bin/bash
# fitted 2-D planner + model-driven dispatch: conv tests, per-layer A/B at 64 / 32 / 16 / 8 images and RetinaNet R50 bs16
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/r74_tests.log 2>&1 || { tail -30 gpurun_out/r74_tests.log; exit 1; }
tail -2 gpurun_out/r74_tests.log
for bs in 64 32 16 8; do
  timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_CONV_WINO2=0,1 > gpurun_out/r74_ab_bs$bs.log 2>&1
  tail -1 gpurun_out/r74_ab_bs$bs.log
done
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_CONV_WINO2=0,1 > gpurun_out/r74_ab_r50.log 2>&1
tail -1 gpurun_out/r74_ab_r50.log
