# round 3, GPU call 19: full -m gpu suite + smoke + the N > 1 bench legs rehearsed at world size 1 (RCCL path)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/r3_t19.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t19.log | cut -c1-300
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r3_smoke2.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_smoke2.log
NNL_BENCH_FORCE_DIST=1 NNL_DIST_FORCE_ALLREDUCE=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_bench_forcedist.json.log 2> gpurun_out/r3_bench_forcedist.err; echo "forced-dist bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_forcedist.json.log').read().strip().splitlines()[-1])
print('dp', json.dumps(d.get('dp'))[:900])
print('strong', json.dumps(d.get('strong'))[:600])
print({k:v['ms_per_step'] for k,v in d['configs'].items()})
PY
tail -c 400 gpurun_out/r3_bench_forcedist.err
