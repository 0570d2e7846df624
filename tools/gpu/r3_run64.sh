#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/r64.log 2>&1
tail -5 gpurun_out/r64.log
