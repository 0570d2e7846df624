# round 3, GPU call 22: BPTT split-K workgroup budget after the cell-kernel fix
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for wg in 128 192 256 384 512 768 1024; do echo "NNL_LSTM_WG=$wg"; NNL_LSTM_WG=$wg timeout -k 10 120 python tools/bench_lstm.py 2>&1 | grep bwd_ms; done
