#!/bin/bash
# 2-D vs 1-D Winograd at other grid sizes (32 / 128 / 16 images)
set -e
mkdir -p gpurun_out
: > gpurun_out/r65.log
for bs in 16 32 128; do
  echo "== bs $bs 1-D ==" >> gpurun_out/r65.log
  timeout -k 10 200 python tools/bench_wino.py --bs $bs 2>&1 | grep -o '"layer": "[a-z0-9]*"\|"wino_us": [0-9.]*' | paste - - >> gpurun_out/r65.log
  echo "== bs $bs 2-D ==" >> gpurun_out/r65.log
  timeout -k 10 200 python tools/bench_wino.py --bs $bs --two-d 2>&1 | grep -o '"layer": "[a-z0-9]*"\|"wino_us": [0-9.]*' | paste - - >> gpurun_out/r65.log
done
cat gpurun_out/r65.log
