#!/bin/bash
# final build sanity: conv + vision + graph tests, smoke, one bench line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py tests/test_graph_gpu.py tests/test_detection.py -q -m gpu > gpurun_out/r77_tests.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r77_tests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r77_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-sweep --configs none > gpurun_out/r77_bench.log 2>gpurun_out/r77_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r77_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'])
PY
