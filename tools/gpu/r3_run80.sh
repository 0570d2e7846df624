#!/bin/bash
# memory-side traffic of the headline with the channel-chunked k order of the 2-D kernel (NNL_WINO2_CHUNK=64), for the record
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export NNL_WINO2_CHUNK=64
mkdir -p gpurun_out/r80
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r80/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none > gpurun_out/r80/fetch.json.log 2>gpurun_out/r80/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r80/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none > /dev/null 2>gpurun_out/r80/pmc_write.err; echo "pmc write rc=$?"
F=$(find gpurun_out/r80/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r80/pmc_write -name "*counter_collection.csv" | head -1)
python - <<PY
import sys, os, shutil
sys.path.insert(0,'tools')
import pmc_traffic
pmc_traffic.main('$F','$W','r3chunk64')
for n in ('r3chunk64_traffic.json','r3chunk64_pmc_fetch_size_summary.csv','r3chunk64_pmc_write_size_summary.csv'):
    if os.path.exists(os.path.join('profiles',n)): shutil.copy(os.path.join('profiles',n), 'gpurun_out/r80/'+n)
PY
find gpurun_out/r80 -name "*.csv" -size +5M -delete
cat gpurun_out/r80/r3chunk64_traffic.json
