#!/bin/bash
# final-candidate build: whole GPU suite + smoke + bench
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_t52.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3_t52.log | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/r3_t52.log | head
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r3_smoke52.log 2>&1; echo "smoke rc=$?"
tail -1 gpurun_out/r3_smoke52.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench52.json.log 2>gpurun_out/r3_bench52.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3_bench52.json.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print('headline',d['value'],d['ms_per_step'],d['roofline']['frac'], {k:v for k,v in d['roofline']['by_kind'].items() if k.startswith('conv')})
        print({k:(v['ms_per_step']) for k,v in d['configs'].items()})
        print({k:v.get('hipgraph_ms_per_step') for k,v in d['strong_scaling_proxy'].items() if isinstance(v,dict)})
PY
