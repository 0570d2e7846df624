#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/r75_tests.log 2>&1 || { tail -30 gpurun_out/r75_tests.log; exit 1; }
tail -2 gpurun_out/r75_tests.log
for bs in 64 32; do
  timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_WINO2_CHUNK=0,64,128 > gpurun_out/r75_ab_bs$bs.log 2>&1
  grep -E "3x3 +(fwd|dgrad)|total" gpurun_out/r75_ab_bs$bs.log
done
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_WINO2_CHUNK=0,64 > gpurun_out/r75_ab_r50.log 2>&1
tail -1 gpurun_out/r75_ab_r50.log
