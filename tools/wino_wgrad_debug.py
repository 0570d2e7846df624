"""Winograd-domain weight gradient (NNL_WGRAD_WINO=1) against the direct wgrad kernel and float64 on a few shapes."""
import os, sys
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd import ops
from neuralnetworklibrary_amd._lib import lib

dev = 'cuda'
def run(N, C, H, W, K, wino):
    os.environ['NNL_WGRAD_WINO'] = str(wino); lib.nnl_reload_env()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dev).requires_grad_(True)
    dy = torch.randn(N, K, H, W, generator=g).to(dev)
    y = ops.conv2d(x, w, None, 1, 1, False)
    y.backward(dy)
    torch.cuda.synchronize()
    return w.grad.detach(), x.detach(), dy

for case in [(64, 64, 56, 56, 64), (64, 128, 28, 28, 128), (64, 256, 14, 14, 256), (16, 256, 64, 64, 256), (3, 64, 10, 6, 128), (8, 128, 28, 28, 256), (5, 64, 64, 64, 36)]:
    N, C, H, W, K = case
    dw0, x, dy = run(N, C, H, W, K, 0)
    dw1, _, _ = run(N, C, H, W, K, 1)
    ref = torch.nn.grad.conv2d_weight(x[:].double().cpu(), (K, C, 3, 3), dy.double().cpu(), padding=1) if N * H * W * C * K < 3e10 else None
    sc = dw0.abs().max().item()
    msg = 'wino-vs-direct %.3e (scale %.3e)' % ((dw1 - dw0).abs().max().item(), sc)
    if ref is not None:
        msg += ' | vs f64: direct %.3e wino %.3e' % ((dw0.double().cpu() - ref).abs().max().item(), (dw1.double().cpu() - ref).abs().max().item())
    print(case, msg, flush=True)
