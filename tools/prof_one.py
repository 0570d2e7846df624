"""Run ONE conv pass of one ResNet-34 layer geometry a few times (for rocprofv3 --pmc / --kernel-trace).
Usage: python tools/prof_one.py l2_3x3 fwd|dgrad|wgrad [--bs 64] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import ops  # noqa: E402
from neuralnetworklibrary_amd._lib import check, lib, ptr, stream  # noqa: E402
from bench_conv import LAYERS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('layer'); ap.add_argument('which')
ap.add_argument('--bs', type=int, default=64); ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
name, C, H, K, R, stride, pad, _ = [l for l in LAYERS if l[0] == a.layer][0]
dev = 'cuda'
N = a.bs
g = ops._geom(N, H, H, C, K, R, R, stride, pad)
x = torch.randn(N, H, H, C, device=dev); w = torch.randn(K, R, R, C, device=dev) * 0.05
y = torch.empty(N, g.P, g.Q, K, device=dev); dy = torch.randn(N, g.P, g.Q, K, device=dev)
wt = torch.empty(C, R, R, K, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
fwsb = int(lib.nnl_conv2d_fwd_workspace_bytes(g)); fws = torch.empty(max(fwsb // 4, 1), device=dev)
dwsb = int(lib.nnl_conv2d_dgrad_workspace_bytes(g)); dws = torch.empty(max(dwsb // 4, 1), device=dev)
wsb = int(lib.nnl_conv2d_wgrad_workspace_bytes(g)); ws = torch.empty(max(wsb // 4, 1), device=dev)
check(lib.nnl_conv2d_weight_transpose(ptr(w), ptr(wt), K, R, R, C, stream()))
for _ in range(a.iters):
    if a.which == 'fwd':
        check(lib.nnl_conv2d_fwd(ptr(x), ptr(w), None, ptr(y), g, 0, ptr(fws), fwsb, ptr(ops._tile_counters(x.device)), None, None, None, stream()))
    elif a.which == 'dgrad':
        check(lib.nnl_conv2d_dgrad(ptr(dy), ptr(wt), ptr(dx), g, None, ptr(dws), dwsb, ptr(ops._tile_counters(x.device)), stream()))
    else:
        check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(ws), wsb, stream()))
torch.cuda.synchronize()
print('done', name, a.which, 'flop', 2.0 * N * g.P * g.Q * K * R * R * C)
