#!/bin/bash
# wgrad on a side stream: correctness (conv / vision / graph / e2e / dist suites), then A/B at 8/16/32/64 images
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py tests/test_graph_gpu.py tests/test_e2e_gpu.py tests/test_detection.py -m gpu -x -q > gpurun_out/r3_t30.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_t30.log
for v in 0 1; do
  for bs in 8 16 32; do
    echo "== NNL_WGRAD_STREAM=$v bs=$bs" >> gpurun_out/r3_side_ab.log
    NNL_WGRAD_STREAM=$v timeout -k 10 200 python tools/bench_small_batch.py --bs $bs --steps 40 >> gpurun_out/r3_side_ab.log 2>&1
  done
  echo "== NNL_WGRAD_STREAM=$v headline" >> gpurun_out/r3_side_ab.log
  NNL_WGRAD_STREAM=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none >> gpurun_out/r3_side_ab.log 2>&1
done
grep -v "^\[" gpurun_out/r3_side_ab.log | cut -c1-400
