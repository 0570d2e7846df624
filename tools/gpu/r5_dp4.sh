cd $GRAFT_REPO_ROOT
O=gpurun_out/r5dp
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_syncbn_gpu.py -x -q -k replays > $O/tests4.log 2>&1; echo "tests rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/tests4.log | tail -30
NNL_BENCH_FORCE_DIST=1 NNL_DIST_FORCE_ALLREDUCE=1 timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-counters --configs tabular > $O/bench_tab_dist.log 2> $O/bench_tab_dist.err; echo "bench rc=$?"; tail -3 $O/bench_tab_dist.err
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_tab_dist.log') if l.startswith('{')][-1])
print(json.dumps(d['configs']['tabular'])[:1500])
PY
