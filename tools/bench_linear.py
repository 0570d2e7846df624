"""Times ops.linear forward / backward of one shape (default: the AWD-LSTM decoder 4480 x 400 -> 47343) with the library's own
per-kind HIP-event profile.  Usage (GPU box): python tools/bench_linear.py [M K N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import _lib, ops  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4480, 400, 47343)
x = torch.randn(M, K, device='cuda', requires_grad=True)
w = torch.randn(N, K, device='cuda', requires_grad=True)
b = torch.zeros(N, device='cuda', requires_grad=True)
dy = torch.randn(M, N, device='cuda')


def step():
    x.grad = w.grad = b.grad = None
    y = ops.linear(x, w, b)
    y.backward(dy)


for _ in range(3):
    step()
torch.cuda.synchronize()
_lib.prof_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n):
    step()
e1.record()
torch.cuda.synchronize()
_lib.prof_enable(False)
prof = _lib.prof_collect()
flop = 2.0 * M * K * N
print('linear %d x %d -> %d: %.3f ms per fwd+bwd (%.1f TF/s over 3 GEMMs)' % (M, K, N, e0.elapsed_time(e1) / n, 3 * flop / (e0.elapsed_time(e1) / n * 1e-3) / 1e12))
for k, v in prof.items():
    if v['launches']:
        print('  %-12s %7.3f ms/step  %s' % (k, v['ms'] / n, ('%.1f TF/s' % (v['work'] / (v['ms'] * 1e-3) / 1e12)) if k.startswith('conv') else ''))
