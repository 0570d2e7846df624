#!/bin/bash
# PF2 (BK16) default on: headline + side configs + the conv/vision/text/detection suites
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r3_bench36.json.log 2>gpurun_out/r3_bench36.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3_bench36.json.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print('headline',d['value'],d['ms_per_step'],d['roofline']['frac'])
        print({k:(v['ms_per_step']) for k,v in d['configs'].items()})
        print({k:v.get('hipgraph_ms_per_step') for k,v in d['strong_scaling_proxy'].items() if isinstance(v,dict)})
PY
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3_t36.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_t36.log
