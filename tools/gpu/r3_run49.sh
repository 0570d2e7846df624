#!/bin/bash
# Winograd PF2 (BK16): kernel checks, conv tests, A/B
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_wino.py --bs 64 > gpurun_out/r3_wino2.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3_wino2.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['layer'], {k:(round(v,8) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('diff') or k.startswith('bn') or k in ('wino_us',)})
PY
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x > gpurun_out/r3_t49.log 2>&1; echo "conv tests rc=$?"
tail -2 gpurun_out/r3_t49.log
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO_PF2=0,1 > gpurun_out/r3_wino_pf2.log 2>&1
grep -E "3x3 |total" gpurun_out/r3_wino_pf2.log | grep -v "s2" | grep -E "fwd|dgrad|total"
