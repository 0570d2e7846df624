#!/bin/bash
# 2-D Winograd dispatched from ~500 quad tiles: conv tests, per-layer A/B on ResNet-34 bs64 and RetinaNet R50 bs16
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/r66_tests.log 2>&1 || { tail -30 gpurun_out/r66_tests.log; exit 1; }
tail -2 gpurun_out/r66_tests.log
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_CONV_WINO2=0,1 > gpurun_out/r66_ab_bs64.log 2>&1
tail -12 gpurun_out/r66_ab_bs64.log
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_CONV_WINO2=0,1 > gpurun_out/r66_ab_r50.log 2>&1
tail -12 gpurun_out/r66_ab_r50.log
