cd $GRAFT_REPO_ROOT
for m in 5 1; do NNL_LSTM_PERSIST=$m timeout -k 10 200 python tools/bench_heads.py lm --steps 20 2>/dev/null | tail -1; done
timeout -k 10 100 python tools/bench_lstm.py 2>/dev/null
