#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python tools/wino2_head_flip_probe.py > gpurun_out/r70.log 2>&1
tail -12 gpurun_out/r70.log
