cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4bptt2
timeout -k 10 600 python -m pytest tests/test_text.py -x -q -m gpu -k "bptt2 or (full_size_recurrence and 13)" > gpurun_out/r4bptt2/test_fwd2.log 2>&1; rc=$?; tail -3 gpurun_out/r4bptt2/test_fwd2.log; echo test_rc=$rc
[ $rc -eq 0 ] || exit $rc
for m in 5 13 5 13; do NNL_LSTM_PERSIST=$m timeout -k 10 100 python tools/bench_lstm.py 2>/dev/null; done
for cfg in "4 64" "4 58" "8 32" "2 116" "4 48"; do set -- $cfg; echo KG=$1 NG=$2; NNL_LSTM_PERSIST=13 NNL_LSTM_FWD2_KG=$1 NNL_LSTM_FWD2_NG=$2 timeout -k 10 100 python tools/bench_lstm.py 2>/dev/null | head -1; done
