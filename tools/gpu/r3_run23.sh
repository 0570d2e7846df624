# round 3, GPU call 23: chunked slab loads in the split-tile fix-up + seq_reg batching: parity and per-layer table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_text.py -m gpu -q 2>&1 | tail -2 | cut -c1-200
timeout -k 10 300 python tools/bench_conv.py --bs 64 2>&1 | grep -v amdgpu.ids | tail -14
timeout -k 10 300 python tools/bench_conv.py --bs 8 2>&1 | grep -v amdgpu.ids | tail -3
