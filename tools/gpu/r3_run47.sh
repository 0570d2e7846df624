# round 3, GPU call 47: evidence for profiles/ on the build with the Winograd path — bench line, rocprofv3 kernel stats (headline + side configs), PMC traffic passes, MFMA busy
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
timeout -k 10 600 python bench.py > gpurun_out/r3p/bench_final.json.log 2> gpurun_out/r3p/bench_final.err; echo "bench rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --configs none > gpurun_out/r3p/bench_under_rocprof.json.log 2>gpurun_out/r3p/rocprof_bench.err; echo "rocprof rc=$?"
for f in $(find gpurun_out/r3p/prof_bench -name "*.db" | head -1); do python tools/stats_csv.py $f gpurun_out/r3p/bench_kernel_stats.csv; done
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof_cfg -o cfg -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --configs lm,retinanet,tabular,collab > gpurun_out/r3p/configs_under_rocprof.json.log 2>gpurun_out/r3p/rocprof_cfg.err; echo "rocprof cfg rc=$?"
for f in $(find gpurun_out/r3p/prof_cfg -name "*.db" | head -1); do python tools/stats_csv.py $f gpurun_out/r3p/configs_kernel_stats.csv; done
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r3p/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none > /dev/null 2>gpurun_out/r3p/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r3p/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none > /dev/null 2>gpurun_out/r3p/pmc_write.err; echo "pmc write rc=$?"
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r3p/pmc_mfma -o m -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none > /dev/null 2>gpurun_out/r3p/pmc_mfma.err; echo "pmc mfma rc=$?"
ls -la gpurun_out/r3p/pmc_fetch gpurun_out/r3p/pmc_write gpurun_out/r3p/pmc_mfma | head -30
F=$(find gpurun_out/r3p/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r3p/pmc_write -name "*counter_collection.csv" | head -1)
mkdir -p gpurun_out/r3p/profiles_out
python - <<PY
import sys, os, shutil
sys.argv=['pmc','$F','$W','r3']
sys.path.insert(0,'tools')
import pmc_traffic
pmc_traffic.main('$F','$W','r3')
for n in ('r3_traffic.json','r3_pmc_fetch_size_summary.csv','r3_pmc_write_size_summary.csv'):
    shutil.copy(os.path.join('profiles',n), 'gpurun_out/r3p/profiles_out/'+n)
PY
find gpurun_out/r3p -name "*.db" -size +20M -delete; find gpurun_out/r3p -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r3p
