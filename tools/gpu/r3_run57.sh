#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WGRAD_WINO_TILE=-1,0,3 > gpurun_out/r3_wwg_tile_bs64.log 2>&1
grep -E "3x3 .*wgrad|total" gpurun_out/r3_wwg_tile_bs64.log | grep -v s2
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_WGRAD_WINO_TILE=-1,0,3 > gpurun_out/r3_wwg_tile_r50.log 2>&1
grep -E "wgrad|total" gpurun_out/r3_wwg_tile_r50.log | grep -E "3x3|head_(64|32|16)|out_|total" | grep -v "s2_3x3s2\|s3_3x3s2\|s4_3x3s2" | head -20
