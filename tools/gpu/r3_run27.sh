# round 3, GPU call 27: single-chunk k loop of the persistent LSTM forward: parity, phases, microbench A/B, LM bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_text.py -m gpu -q 2>&1 | tail -2 | cut -c1-200
NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so timeout -k 10 200 python tools/lstm_timing.py 2>&1 | grep -v amdgpu.ids
for v in 0 1; do echo "NNL_LSTM_SINGLE=$v"; NNL_LSTM_SINGLE=$v timeout -k 10 120 python tools/bench_lstm.py 2>&1 | grep fwd_ms; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs lm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); v=d['configs']['lm']; print('lm', v['ms_per_step'], v['value'], v['roofline']['by_kind']['lstm'])"
