#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
for kg in 1 2; do
NNL_WGRAD_WINO=2 NNL_WGRAD_KG=$kg timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 > gpurun_out/r3_wwg_r50_kg$kg.log 2>&1
echo "== forced wino, KG=$kg"; grep -E "wgrad" gpurun_out/r3_wwg_r50_kg$kg.log | grep -E "3x3_|fpn_3x3" | grep -v s2
done
