#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_bn_gpu.py tests/test_tabular.py -m gpu -q -x > gpurun_out/r3_t61.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_t61.log | cut -c1-300
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_IGEMM_EPI4=0,1 > gpurun_out/r3_epi4_r50.log 2>&1
grep -E "1x1|ds_|lat|s2_3x3s2|total" gpurun_out/r3_epi4_r50.log | grep -E "fwd|dgrad|total" | head -60
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_IGEMM_EPI4=0,1 > gpurun_out/r3_epi4_bs64.log 2>&1
grep -E "total|s2 |stem" gpurun_out/r3_epi4_bs64.log | head
