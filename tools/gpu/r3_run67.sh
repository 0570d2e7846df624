#!/bin/bash
# the full-size ResNet-34 parity test with the 2-D Winograd dispatch off / on: per-tensor margins
mkdir -p gpurun_out
NNL_CONV_WINO2=0 timeout -k 10 400 python -m pytest tests/test_vision_gpu.py -q -m gpu -s -k test_resnet34_full_baseline_size_forward_backward_vs_oracle > gpurun_out/r67_off.log 2>&1
grep -A8 "closest to the bound" gpurun_out/r67_off.log; tail -2 gpurun_out/r67_off.log
timeout -k 10 400 python -m pytest tests/test_vision_gpu.py -q -m gpu -s -k test_resnet34_full_baseline_size_forward_backward_vs_oracle > gpurun_out/r67_on.log 2>&1
grep -A8 "closest to the bound" gpurun_out/r67_on.log; tail -2 gpurun_out/r67_on.log
exit 0
