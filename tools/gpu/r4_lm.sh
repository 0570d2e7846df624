cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_decoder_gemm.py 2>/dev/null | head -2
timeout -k 10 600 python -m pytest tests/test_text.py -x -q -m gpu -k "language_model or g14 or g7 or softmax" 2>&1 | tail -2
for i in 1 2; do timeout -k 10 200 python tools/bench_heads.py lm --steps 20 2>/dev/null | tail -1 | cut -c1-330; done
