"""BatchNorm(+residual)(+ReLU) kernels per ResNet-34 stage through the C ABI (no autograd / Python overhead between the launches): time and
achieved HBM-side bandwidth of nnl_bn_fwd (statistics + finalize + apply) and nnl_bn_bwd (reduce + finalize + apply).
Usage: python tools/bench_bn.py [--bs 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd._lib import check, lib, ptr, stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--bs', type=int, default=64)
a = ap.parse_args()
dev = 'cuda'


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


print('%-22s %8s | %9s %8s | %9s %8s' % ('shape', 'MB', 'fwd us', 'TB/s', 'bwd us', 'TB/s'))
for C, H in ((64, 56), (128, 28), (256, 14), (512, 7)):
    rows = a.bs * H * H
    x = torch.randn(rows, C, device=dev); res = torch.randn(rows, C, device=dev); dy = torch.randn(rows, C, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x); dres = torch.empty_like(x)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    mask = torch.empty((rows * C + 31) // 32 + 2, dtype=torch.int32, device=dev)
    wsb = int(lib.nnl_bn_workspace_bytes(rows, C)); ws = torch.empty(wsb // 4, device=dev)
    fwd = lambda: check(lib.nnl_bn_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(res), ptr(y), ptr(mean), ptr(invstd), ptr(rm), ptr(rv), rows, C, 1e-5, 0.1, 1, 1,
                                       None, ptr(mask), None, 0, None, None, ptr(ws), wsb, stream()))
    bwd = lambda: check(lib.nnl_bn_bwd(ptr(dy), None, ptr(mask), ptr(x), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx), ptr(dres), ptr(dg), ptr(db), rows, C, 1, 1,
                                       ptr(ws), wsb, stream()))
    tf = timeit(fwd); tb = timeit(bwd)
    elems = rows * C
    fb = elems * (4 + 8 + 4 + 0.125)               # stats read, apply read + write, residual read, mask write
    bb = elems * (8 + 0.125 + 8 + 0.125 + 4 + 4)   # reduce: dy, x, mask; apply: dy, x, mask, dx, dres
    print('%-22s %8.1f | %9.1f %8.2f | %9.1f %8.2f' % ('[%d,%d,%d,%d]' % (a.bs, C, H, H), elems * 4 / 1e6, tf, fb / tf / 1e6, tb, bb / tb / 1e6))
