"""The three GEMMs of the AWD-LSTM decoder (Text.py:572: nn.Linear(400, V = 47343) on 64 x 70 = 4480 rows) through the conv entry
points, with forced output tiles: NNL_IGEMM_TILE (forward / dgrad) and NNL_WGRAD_TILE (wgrad).  python tools/bench_decoder_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import _lib  # noqa: E402
from neuralnetworklibrary_amd._lib import lib, ptr, check, stream  # noqa: E402

dev = torch.device('cuda', 0)
M, C, K = 4480, 400, 47344
x = torch.randn(M, C, device=dev)
w = torch.randn(K, C, device=dev) * 0.05
wt = w.t().contiguous()
dy = torch.randn(M, K, device=dev)
y = torch.empty(M, K, device=dev)
dx = torch.empty(M, C, device=dev)
g = _lib.ConvGeom(M, 1, 1, C, K, 1, 1, 1, 0, 1, 1)
cnt = torch.zeros(int(lib.nnl_conv2d_tile_counters()), dtype=torch.int32, device=dev)


def timed(fn, n=8):
    for _ in range(2):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


for tile in ('-1', '0', '1', '2', '3'):
    os.environ['NNL_IGEMM_TILE'] = tile
    lib.nnl_reload_env()
    wsf = int(lib.nnl_conv2d_fwd_workspace_bytes(g)); wf = torch.empty(max(wsf // 4, 1), device=dev)
    wsd = int(lib.nnl_conv2d_dgrad_workspace_bytes(g)); wd = torch.empty(max(wsd // 4, 1), device=dev)
    tf = timed(lambda: check(lib.nnl_conv2d_fwd(ptr(x), ptr(w), None, ptr(y), g, 0, ptr(wf), wsf, ptr(cnt), None, None, None, stream())))
    td = timed(lambda: check(lib.nnl_conv2d_dgrad(ptr(dy), ptr(wt), ptr(dx), g, None, ptr(wd), wsd, ptr(cnt), stream())))
    fl = 2.0 * M * C * K / 1e9
    print('NNL_IGEMM_TILE=%s: fwd %.3f ms %.1f TF/s | dgrad %.3f ms %.1f TF/s' % (tile, tf, fl / tf, td, fl / td))

# weight gradient dw[K, C] = dy^T x through the C ABI (the dispatcher's own tile; NNL_WGRAD_TILE needs the A/B build), and what the
# vendor library (torch.mm -> hipBLASLt / rocBLAS) reaches on the same three products: the reference point for "how far is the tile loop
# from a tuned fp32 GEMM on this part"
os.environ.pop('NNL_IGEMM_TILE', None)
lib.nnl_reload_env()
dw = torch.empty(K, C, device=dev)
wsw = int(lib.nnl_conv2d_wgrad_workspace_bytes(g)); ww = torch.empty(max(wsw // 4, 1), device=dev)
tw = timed(lambda: check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(ww), wsw, stream())))
fl = 2.0 * M * C * K / 1e9
print('wgrad (default plan): %.3f ms %.1f TF/s' % (tw, fl / tw))
ref = dy.t() @ x
print('wgrad max |err| vs torch.mm: %.3g (of %.3g)' % ((dw - ref).abs().max().item(), ref.abs().max().item()))
for name, fn in (('fwd   y = x w^T', lambda: torch.mm(x, w.t(), out=y)), ('dgrad dx = dy w', lambda: torch.mm(dy, w, out=dx)),
                 ('wgrad dw = dy^T x', lambda: torch.mm(dy.t(), x, out=dw))):
    t = timed(fn)
    print('torch.mm %-18s %.3f ms %.1f TF/s' % (name, t, fl / t))
