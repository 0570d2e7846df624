#!/bin/bash
# rehearsal of the N>1 bench path on the final build: forced process group at world size 1 (RCCL call path, GradSync, DP graph step),
# and the torchrun launch line the driver uses with one rank
set -x
cd /root/repo; export TMPDIR=/tmp
NNL_BENCH_FORCE_DIST=1 NNL_DIST_FORCE_ALLREDUCE=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_bench_forcedist2.json.log 2> gpurun_out/r3_bench_forcedist2.err; echo "forced-dist bench rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3_bench_forcedist2.json.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['n_gpus'], d.get('ranks_seen'), json.dumps(d.get('dp'))[:600])
PY
tail -3 gpurun_out/r3_bench_forcedist2.err
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --configs none > gpurun_out/r3_bench_torchrun1.json.log 2> gpurun_out/r3_bench_torchrun1.err; echo "torchrun bench rc=$?"
grep '^{' gpurun_out/r3_bench_torchrun1.json.log | cut -c1-300
