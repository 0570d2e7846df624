# round 4: the bench line of the final build (the driver's command), kept under profiles/r4_bench_final.json.log
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p
mkdir -p $O
timeout -k 10 1100 python bench.py > $O/bench_final.json.log 2> $O/bench_final.err; echo "bench rc=$?"
tail -c 600 $O/bench_final.json.log
