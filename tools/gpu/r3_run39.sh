#!/bin/bash
# cross-barrier MFMA split (NNL_IGEMM_VARIANT=2): correctness with the variant forced, per-layer A/B at 64 and 8 images, headline A/B
set -x
cd /root/repo; export TMPDIR=/tmp
NNL_IGEMM_VARIANT=2 timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py -m gpu -x -q > gpurun_out/r3_t39.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_t39.log
timeout -k 10 400 python tools/bench_conv.py --bs 64 --ab NNL_IGEMM_VARIANT=1,2 > gpurun_out/r3_xb_bs64.log 2>&1; echo "ab rc=$?"
tail -36 gpurun_out/r3_xb_bs64.log
timeout -k 10 400 python tools/bench_conv.py --bs 8 --ab NNL_IGEMM_VARIANT=1,2 > gpurun_out/r3_xb_bs8.log 2>&1; echo "ab rc=$?"
tail -2 gpurun_out/r3_xb_bs8.log
for v in 1 2; do
  echo "== NNL_IGEMM_VARIANT=$v headline" >> gpurun_out/r3_xb_head.log
  NNL_IGEMM_VARIANT=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none 2>&1 | grep '^{' | cut -c1-330 >> gpurun_out/r3_xb_head.log
done
cat gpurun_out/r3_xb_head.log
