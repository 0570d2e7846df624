#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/wino2_net_probe.py > gpurun_out/r69.log 2>&1
cat gpurun_out/r69.log | tail -30
