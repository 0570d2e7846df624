cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4bptt2
timeout -k 10 400 python -m pytest tests/test_text.py -x -q -m gpu -k "bptt2 or (full_size_recurrence and 5)" > gpurun_out/r4bptt2/test.log 2>&1; rc=$?; tail -3 gpurun_out/r4bptt2/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
export NNL_LSTM_PERSIST=5
NNL_LSTM_PERSIST=1 timeout -k 10 100 python tools/bench_bptt.py 2>/dev/null
for i in 1 2; do timeout -k 10 100 python tools/bench_bptt.py 2>/dev/null; done
for d in 2 3; do NNL_LSTM_BPTT2_DBG=$d timeout -k 10 100 python tools/bench_bptt.py 2>/dev/null; done
timeout -k 10 100 python tools/bench_bptt.py 400 2>/dev/null
for m in 5 5; do NNL_LSTM_PERSIST=$m timeout -k 10 200 python tools/bench_heads.py lm --steps 20 2>/dev/null | tail -1 | cut -c1-120; done
