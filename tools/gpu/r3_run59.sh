#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -m gpu -q > gpurun_out/r3_t59.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_t59.log | cut -c1-300
grep -E "^E " gpurun_out/r3_t59.log | head -5
