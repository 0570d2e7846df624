#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_vision_gpu.py -q -m gpu -s -k test_resnet34_full_baseline_size_forward_backward_vs_oracle > gpurun_out/r71.log 2>&1
grep -A8 "closest to the bound" gpurun_out/r71.log; grep -A6 "head gates" gpurun_out/r71.log; tail -3 gpurun_out/r71.log
