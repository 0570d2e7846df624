import os, sys, json
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd import ops
from neuralnetworklibrary_amd._lib import lib

dev = 'cuda'
def run(N, C, H, W, K, relu, bias, wino):
    os.environ['NNL_CONV_WINO'] = str(wino); lib.nnl_reload_env()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(K, generator=g).to(dev).requires_grad_(True) if bias else None
    dy = torch.randn(N, K, H, W, generator=g).to(dev)
    y = ops.conv2d(x, w, b, 1, 1, relu)
    y.backward(dy)
    torch.cuda.synchronize()
    return y.detach(), x.grad.detach(), w.grad.detach()

for case in [(2, 256, 2, 2, 256, False, True), (2, 256, 2, 2, 256, True, True), (3, 256, 4, 4, 256, True, True), (1, 64, 2, 2, 64, False, False), (2, 256, 8, 8, 36, False, True), (2, 256, 8, 8, 180, False, True), (2, 256, 2, 2, 180, False, True), (2, 256, 64, 64, 256, True, True), (2, 256, 64, 64, 256, False, False), (2, 256, 64, 64, 256, False, True), (2, 128, 64, 64, 256, False, False), (2, 256, 64, 64, 128, False, False), (4, 256, 32, 32, 256, False, False)]:
    N, C, H, W, K, relu, bias = case
    y0, dx0, dw0 = run(N, C, H, W, K, relu, bias, 0)
    y1, dx1, dw1 = run(N, C, H, W, K, relu, bias, 1)
    e = (dx1 - dx0).abs()
    bad = (e > 1e-3).nonzero()
    print(case, 'y', (y1 - y0).abs().max().item(), 'dx', e.max().item(), 'dw', (dw1 - dw0).abs().max().item(), 'nbad', len(bad))
    if len(bad):
        print('  bad n', sorted(set(bad[:, 0].tolist()))[:8], 'c range', bad[:, 1].min().item(), bad[:, 1].max().item(),
              'h', sorted(set(bad[:, 2].tolist()))[:70], 'w', sorted(set(bad[:, 3].tolist()))[:70])
