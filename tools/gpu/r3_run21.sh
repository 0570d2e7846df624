# round 3, GPU call 21: BPTT cell kernel with batched slab loads — text parity + LM bench + LSTM microbench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_text.py tests/test_conv_gpu.py -m gpu -q 2>&1 | tail -3 | cut -c1-300
timeout -k 10 300 python tools/bench_lstm.py 2>&1 | tail -12
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs lm > gpurun_out/r3_bench_lm.json.log 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_lm.json.log').read().strip().splitlines()[-1])
v=d['configs']['lm']; print('lm', v['ms_per_step'], v['median_ms_per_step'], v['value'], v['roofline']['by_kind'])
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_kind']['conv_wgrad'])
PY
