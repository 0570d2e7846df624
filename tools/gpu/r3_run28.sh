# round 3, GPU call 28: chunk depth of the persistent LSTM forward k loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 12 6 9 18; do echo "NNL_LSTM_PD=$v"; NNL_LSTM_PD=$v timeout -k 10 120 python tools/bench_lstm.py 2>&1 | grep fwd_ms | cut -c1-110; done
