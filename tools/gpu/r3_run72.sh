#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python tools/wino2_plan_sweep.py > gpurun_out/r72.log 2>&1
cat gpurun_out/r72.log
