#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r73.log
for bs in 16 32 64; do timeout -k 10 300 python tools/wino2_plan_sweep.py --bs $bs >> gpurun_out/r73.log 2>&1; done
grep -c bk gpurun_out/r73.log
