#!/bin/bash
# DP rehearsal on the final build: forced process group at world size 1 (RCCL), and torchrun with one rank
mkdir -p gpurun_out
NNL_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --configs none > gpurun_out/r81_force.log 2>gpurun_out/r81_force.err; echo "force-dist rc=$?"
tail -1 gpurun_out/r81_force.log | cut -c1-400
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --configs none > gpurun_out/r81_trun.log 2>gpurun_out/r81_trun.err; echo "torchrun rc=$?"
tail -1 gpurun_out/r81_trun.log | cut -c1-300
