#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py -m gpu -q > gpurun_out/r3_t56.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r3_t56.log | cut -c1-200
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WGRAD_WINO=0,1 > gpurun_out/r3_wwg3_bs64.log 2>&1
grep -E "3x3 .*wgrad|total" gpurun_out/r3_wwg3_bs64.log | grep -v s2
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs retinanet 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline',d['value'],d['ms_per_step'],d['roofline']['frac'],'retinanet',d['configs']['retinanet']['ms_per_step'])"
