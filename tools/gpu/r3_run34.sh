#!/bin/bash
# memory latency / hit-rate passes over the conv kernels (3 bench steps each)
set -x
cd /root/repo; export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none"
i=5
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/prof_r3_stall$i -o s$i -- $CMD > gpurun_out/r3_stall$i.log 2>&1; echo "pass $i rc=$?"
  tail -2 gpurun_out/r3_stall$i.log | cut -c1-300
done
ls gpurun_out/prof_r3_stall*/ | head -40
