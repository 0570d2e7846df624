"""Accuracy of the three 3x3 / stride 1 kernels (direct, 1-D Winograd, 2-D Winograd) on a chain of conv -> BatchNorm (batch statistics)
-> ReLU layers with post-ReLU inputs, against float64 on the CPU: relative L2 error after each layer.   python tools/wino2_accuracy.py"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from neuralnetworklibrary_amd import ops
from neuralnetworklibrary_amd._lib import lib


def main():
    dev = torch.device('cuda:0')
    N, C, H, L = 8, 64, 56, 6
    g = torch.Generator().manual_seed(3)
    x0 = torch.relu(torch.randn(N, C, H, H, generator=g) + 0.3)
    ws = [torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5 for _ in range(L)]
    gam = [1 + 0.1 * torch.randn(C, generator=g) for _ in range(L)]
    bet = [0.1 * torch.randn(C, generator=g) for _ in range(L)]

    def chain(conv, x, dt):
        outs = []
        for l in range(L):
            y = conv(x, ws[l].to(x.device, dt))
            y = F.batch_norm(y, None, None, gam[l].to(x.device, dt), bet[l].to(x.device, dt), True, 0.1, 1e-5)
            x = torch.relu(y)
            outs.append(x.detach().double().cpu())
        return outs

    ref = chain(lambda x, w: F.conv2d(x, w, padding=1), x0.double(), torch.float64)
    cpu32 = chain(lambda x, w: F.conv2d(x, w, padding=1), x0.clone(), torch.float32)
    res = {'cpu32': [((a - b).norm() / b.norm()).item() for a, b in zip(cpu32, ref)]}
    for name, mode in (('direct', '0'), ('wino1d', '2'), ('wino2d', '3')):
        os.environ['NNL_CONV_WINO'] = mode; lib.nnl_reload_env()
        with torch.no_grad():
            xs = x0.to(dev).contiguous(memory_format=torch.channels_last)
            outs = chain(lambda x, w: ops.conv2d(x, w.contiguous(memory_format=torch.channels_last), None, 1, 1, False), xs, torch.float32)
        res[name] = [((a - b).norm() / b.norm()).item() for a, b in zip(outs, ref)]
        # single conv on the post-ReLU input, no BatchNorm
        with torch.no_grad():
            y = ops.conv2d(xs, ws[0].to(dev).contiguous(memory_format=torch.channels_last), None, 1, 1, False).double().cpu()
        r = F.conv2d(x0.double(), ws[0].double(), padding=1)
        res[name + '_conv_only'] = [((y - r).norm() / r.norm()).item(), ((y - r).abs().max() / r.abs().max()).item()]
    os.environ['NNL_CONV_WINO'] = '1'; lib.nnl_reload_env()
    for k, v in res.items():
        print(k, ' '.join('%.2e' % e for e in v))


if __name__ == '__main__':
    main()
