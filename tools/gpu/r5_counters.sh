# round 4: the evidence bench.py's roofline block points at — bench line, rocprofv3 kernel stats, PMC traffic + MFMA-busy passes, all on ONE build
# whose source stamp is written into profiles/r5_traffic.json (bench.py prints the counter figures only while the loaded library matches it)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-counters --configs none > $O/bench_under_rocprof.json.log 2>$O/rocprof_bench.err; echo "rocprof rc=$?"
for f in $(find $O/prof_bench -name "*.db" | head -1); do python tools/stats_csv.py $f $O/bench_kernel_stats.csv; done
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-counters --configs none > /dev/null 2>$O/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-counters --configs none > /dev/null 2>$O/pmc_write.err; echo "pmc write rc=$?"
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-counters --configs none > /dev/null 2>$O/pmc_mfma.err; echo "pmc mfma rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); M=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
mkdir -p $O/profiles_out
python - <<PY
import sys, os, shutil
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import pmc_traffic
from neuralnetworklibrary_amd import _lib
pmc_traffic.main('$F', '$W', 'r5', '$M', _lib.source_stamp())
for n in ('r5_traffic.json', 'r5_pmc_fetch_size_summary.csv', 'r5_pmc_write_size_summary.csv', 'r5_pmc_mfma_busy.csv'):
    shutil.copy(os.path.join('profiles', n), '$O/profiles_out/' + n)
PY
find $O -name "*.db" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete
du -sh $O
