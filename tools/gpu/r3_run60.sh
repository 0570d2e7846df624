#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_bn_gpu.py tests/test_conv_gpu.py -m gpu -q -x > gpurun_out/r3_t60.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_t60.log | cut -c1-300
grep -E "^E " gpurun_out/r3_t60.log | head -8
for v in 0 1; do
NNL_BN_BWD_EPI=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EPI=$v headline',d['value'],d['ms_per_step'],d['roofline']['by_kind']['elementwise'], d['roofline']['by_kind']['conv_dgrad'])"
done
