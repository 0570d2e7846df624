#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_vision_gpu.py tests/test_detection.py -q -m gpu > gpurun_out/r79.log 2>&1; echo "vision+detection rc=$?"; tail -3 gpurun_out/r79.log
