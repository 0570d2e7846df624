#!/bin/bash
# G12 slice ratios with both kernels; the balanced-schedule tests for both kernels; per-layer A/B at 8 / 16 / 32 images
set -x
cd /root/repo; export TMPDIR=/tmp
for v in 0 1; do
  NNL_CONV_WINO=$v timeout -k 10 400 python -m pytest tests/test_detection.py -m gpu -q -s -k g12_objectdetectionnet_hip > gpurun_out/r3_g12_w$v.log 2>&1; echo "g12 wino=$v rc=$?"
  grep "G12 eval gradient" gpurun_out/r3_g12_w$v.log | cut -c1-400
done
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x > gpurun_out/r3_t46.log 2>&1; echo "conv tests rc=$?"
tail -3 gpurun_out/r3_t46.log
for bs in 8 16 32; do
  timeout -k 10 300 python tools/bench_conv.py --bs $bs --ab NNL_CONV_WINO=0,1 > gpurun_out/r3_wino_bs$bs.log 2>&1
  grep -E "3x3 |total" gpurun_out/r3_wino_bs$bs.log | grep -v "s2" | grep -E "fwd|dgrad|total"
done
