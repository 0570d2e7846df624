#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_wino.py --bs 64 > gpurun_out/r3_wino3.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3_wino3.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['layer'], {k:(round(v,9) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('diff') or k.startswith('bn') or k in ('wino_us',)})
PY
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_vision_gpu.py -m gpu -q -x > gpurun_out/r3_t62.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r3_t62.log | cut -c1-200
timeout -k 10 300 python tools/bench_conv.py --bs 64 --ab NNL_WINO_EPI4=0,1 > gpurun_out/r3_wino_epi4_bs64.log 2>&1
grep -E "3x3 |total" gpurun_out/r3_wino_epi4_bs64.log | grep -v "s2" | grep -E "fwd|dgrad|total"
timeout -k 10 400 python tools/bench_conv.py --net r50 --bs 16 --ab NNL_WINO_EPI4=0,1 > gpurun_out/r3_wino_epi4_r50.log 2>&1
grep -E "total|head_64|fpn_3x3_64|s2_3x3_128|s3_3x3_256" gpurun_out/r3_wino_epi4_r50.log | grep -E "fwd|dgrad|total"
