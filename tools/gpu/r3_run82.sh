#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "winograd or wino" > gpurun_out/r82_tests.log 2>&1 || { tail -20 gpurun_out/r82_tests.log; exit 1; }
tail -1 gpurun_out/r82_tests.log
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO2_FOLD_SKIP=0,1 > gpurun_out/r82_ab_bs64.log 2>&1
grep -E "3x3 +(fwd|dgrad)|total" gpurun_out/r82_ab_bs64.log
