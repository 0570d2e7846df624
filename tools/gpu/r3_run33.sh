#!/bin/bash
# which SQ counters exist on gfx950, then stall attribution passes over the conv kernels (3 bench steps each)
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 -L > gpurun_out/r3_counters_avail.log 2>&1; echo "list rc=$?"
grep -o "SQ_[A-Z0-9_]*" gpurun_out/r3_counters_avail.log | sort -u > gpurun_out/r3_sq_counters.txt
wc -l gpurun_out/r3_sq_counters.txt
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/prof_r3_stall$i -o s$i -- $CMD > gpurun_out/r3_stall$i.log 2>&1; echo "pass $i rc=$?"
  find gpurun_out/prof_r3_stall$i -name "*counter_collection.csv" | head -2
done
ls -la gpurun_out/prof_r3_stall*/ | head -40
