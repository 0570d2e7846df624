cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5dp
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_dp_replay_two_ranks_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/tests.log | tail -30
timeout -k 10 900 python3 bench.py --steps 5 --warmup 2 --no-sweep --configs collab --counters > $O/bench_counters.log 2> $O/bench_counters.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_counters.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'].get('measured_counters'))); print(d['roofline']['traffic'], d['roofline']['mfma_busy'], d['roofline']['counters_source'][:80]); print(json.dumps(d['cpu_baseline'])[:600]); print(json.dumps(d['configs']['collab'].get('cpu_baseline'))[:400])
PY
