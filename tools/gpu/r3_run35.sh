#!/bin/bash
# two tiles in flight (NNL_IGEMM_PF2): correctness with the variant forced, per-layer A/B, headline A/B
set -x
cd /root/repo; export TMPDIR=/tmp
NNL_IGEMM_PF2=3 timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -m gpu -x -q > gpurun_out/r3_t35.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_t35.log
timeout -k 10 400 python tools/bench_conv.py --bs 64 --ab NNL_IGEMM_PF2=0,1,2,3 > gpurun_out/r3_pf2_bs64.log 2>&1; echo "ab rc=$?"
tail -45 gpurun_out/r3_pf2_bs64.log
for v in 0 2 3; do
  echo "== NNL_IGEMM_PF2=$v headline" >> gpurun_out/r3_pf2_head.log
  NNL_IGEMM_PF2=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none 2>&1 | grep '^{' | cut -c1-330 >> gpurun_out/r3_pf2_head.log
done
cat gpurun_out/r3_pf2_head.log
