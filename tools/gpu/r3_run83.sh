#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO2_BK=0,16,32 > gpurun_out/r83_ab_bs64.log 2>&1
grep -E "3x3 +(fwd|dgrad)|total" gpurun_out/r83_ab_bs64.log
