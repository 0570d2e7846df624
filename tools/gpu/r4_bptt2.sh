cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4bptt2
timeout -k 10 400 python -m pytest tests/test_text.py -x -q -m gpu -k "bptt2 or (full_size_recurrence and 5)" > gpurun_out/r4bptt2/test.log 2>&1; rc=$?; tail -5 gpurun_out/r4bptt2/test.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
export NNL_LSTM_PERSIST=5
NNL_LSTM_PERSIST=1 timeout -k 10 100 python tools/bench_bptt.py 2>/dev/null
for cfg in "16 16" "9 24" "12 18" "18 12" "8 24" "16 15"; do set -- $cfg; NNL_LSTM_BPTT2_KG=$1 NNL_LSTM_BPTT2_NG=$2 timeout -k 10 100 python tools/bench_bptt.py 2>/dev/null; done
for d in 1 2 3; do NNL_LSTM_BPTT2_DBG=$d NNL_LSTM_BPTT2_KG=16 NNL_LSTM_BPTT2_NG=16 timeout -k 10 100 python tools/bench_bptt.py 2>/dev/null; done
NNL_LSTM_PERSIST=1 timeout -k 10 100 python tools/bench_bptt.py 400 2>/dev/null
timeout -k 10 100 python tools/bench_bptt.py 400 2>/dev/null
