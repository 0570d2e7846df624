#!/bin/bash
# one-launch BatchNorm backward for small tensors: tests, then bs8 / tabular / headline numbers
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_bn_gpu.py tests/test_tabular.py tests/test_vision_gpu.py tests/test_graph_gpu.py -m gpu -x -q > gpurun_out/r3_t41.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_t41.log
for v in 0 2048; do
  echo "== NNL_BN_SMALL_ROWS=$v" >> gpurun_out/r3_bnsmall.log
  NNL_BN_SMALL_ROWS=$v timeout -k 10 200 python tools/bench_small_batch.py --bs 8 --steps 40 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if k!='by_kind'}, d['by_kind'].get('elementwise'))" >> gpurun_out/r3_bnsmall.log
  NNL_BN_SMALL_ROWS=$v timeout -k 10 200 python tools/bench_small_batch.py --bs 16 --steps 40 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if k!='by_kind'}, d['by_kind'].get('elementwise'))" >> gpurun_out/r3_bnsmall.log
  NNL_BN_SMALL_ROWS=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs tabular 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline',d['value'],d['ms_per_step'],'tabular',d['configs']['tabular']['ms_per_step'])" >> gpurun_out/r3_bnsmall.log
done
cat gpurun_out/r3_bnsmall.log
