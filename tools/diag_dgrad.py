"""Diagnostic: dx / dw error of single convolutions (vs torch CPU fp64) under the A/B switches of the conv planner."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralnetworklibrary_amd import ops  # noqa: E402

GEOMS = [(2, 256, 64, 64, 36), (2, 256, 32, 32, 36), (16, 256, 16, 16, 36), (2, 256, 64, 64, 48), (2, 256, 64, 64, 180),
         (2, 256, 64, 64, 256), (2, 256, 64, 64, 32), (2, 256, 64, 64, 16)]
SETTINGS = [{}, {'NNL_IGEMM_BALANCE': '0'}, {'NNL_DGRAD_PAD16': '0'}, {'NNL_IGEMM_TILE': '3'}, {'NNL_IGEMM_TILE': '1'}, {'NNL_IGEMM_VARIANT': '0'}]
for N, C, H, W, K in GEOMS:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 3, 3, generator=g) / (9 * C) ** .5
    dy = torch.randn(N, K, H, W, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xd, wd, None, padding=1).backward(dy.double())
    line = '%-22s' % str((N, C, H, W, K))
    for st in SETTINGS:
        for k in ('NNL_IGEMM_BALANCE', 'NNL_DGRAD_PAD16', 'NNL_IGEMM_TILE', 'NNL_IGEMM_VARIANT'):
            os.environ.pop(k, None)
        os.environ.update(st)
        ops.lib.nnl_reload_env()
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        ops.conv2d(xg, wg, None, 1, 1).backward(dy.cuda())
        e = ((xg.grad.cpu().double() - xd.grad).norm() / xd.grad.norm()).item()
        line += ' | %s dx %.1e' % (','.join('%s=%s' % (k[4:], v) for k, v in st.items()) or 'default', e)
    print(line)
