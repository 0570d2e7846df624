#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/wino2_accuracy.py > gpurun_out/r68.log 2>&1
cat gpurun_out/r68.log
