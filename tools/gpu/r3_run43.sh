#!/bin/bash
# Winograd path integrated (NNL_CONV_WINO default 1): conv / vision / detection / graph suites, per-layer A/B, headline A/B
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py tests/test_bn_gpu.py tests/test_graph_gpu.py -m gpu -x -q > gpurun_out/r3_t43.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3_t43.log
timeout -k 10 400 python tools/bench_conv.py --bs 64 --ab NNL_CONV_WINO=0,1 > gpurun_out/r3_wino_bs64.log 2>&1; echo "ab rc=$?"
tail -36 gpurun_out/r3_wino_bs64.log
for v in 0 1; do
  echo "== NNL_CONV_WINO=$v headline" >> gpurun_out/r3_wino_head.log
  NNL_CONV_WINO=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none 2>&1 | grep '^{' | cut -c1-330 >> gpurun_out/r3_wino_head.log
done
cat gpurun_out/r3_wino_head.log
