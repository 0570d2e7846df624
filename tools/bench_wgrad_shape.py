"""Time nnl_conv2d_wgrad on a 1x1 'conv' over N pixels (the dW GEMMs of Linear / LSTM layers): dW [K, C] = dy[N, K]^T x[N, C].
Usage: python tools/bench_wgrad_shape.py N C K [ENVVAR=v0,v1,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import _lib  # noqa: E402
from neuralnetworklibrary_amd._lib import lib, ptr, check, stream  # noqa: E402

N, C, K = (int(v) for v in sys.argv[1:4])
ab = sys.argv[4] if len(sys.argv) > 4 else None
var, vals = (ab.split('=')[0], ab.split('=')[1].split(',')) if ab else (None, [None])
dev = torch.device('cuda', 0)
x = torch.randn(N, C, device=dev)
dy = torch.randn(N, K, device=dev)
dw = torch.empty(K, C, device=dev)
g = _lib.ConvGeom(N, 1, 1, C, K, 1, 1, 1, 0, 1, 1)
for v in vals:
    if var:
        os.environ[var] = v
        lib.nnl_reload_env()
    wb = int(lib.nnl_conv2d_wgrad_workspace_bytes(g))
    ws = torch.empty(max(wb // 4, 1), device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(3):
        check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(ws), wb, stream()))
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(10):
        check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(ws), wb, stream()))
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print('N %d C %d K %d %s=%s: %.3f ms  %.1f TF/s' % (N, C, K, var, v, ms, 2.0 * N * C * K / ms / 1e9))
