#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WGRAD_KG=-1,2,4 > gpurun_out/r3_wwg_kg_bs64.log 2>&1
grep -E "3x3 .*wgrad|total" gpurun_out/r3_wwg_kg_bs64.log | grep -v s2
