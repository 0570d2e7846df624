# round 3, GPU call 15: wgrad with KG wave groups per workgroup (slab traffic / KG): parity + A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_text.py tests/test_tabular.py tests/test_vision_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 400 python tools/bench_conv.py --bs 64 --ab NNL_WGRAD_KG=1,2,4,-1 2>&1 | grep "wgrad\|layer\|total"
timeout -k 10 400 python tools/bench_conv.py --bs 8 --ab NNL_WGRAD_KG=1,-1 2>&1 | grep "wgrad\|layer\|total"
