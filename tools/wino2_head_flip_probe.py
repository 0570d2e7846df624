"""Is the head-gradient difference of the full-size ResNet-34 parity test a ReLU gate flip in the head?  Pre-activations of the head's
first Linear (product net on the GPU vs the float64 oracle): sign mismatches and their magnitudes.   python tools/wino2_head_flip_probe.py"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import synth, reference_nets as RNets
from neuralnetworklibrary_amd._lib import lib


def main():
    from neuralnetworklibrary_amd.Applications import Vision as V
    N, S = 64, 224

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(N, 3, S, S, generator=g), torch.randint(0, 2, (N,), generator=g)
    onet = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.), probe_sz=(S, S))
    synth.fill_module_(onet, seed=5)
    onet64 = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.)).double()
    onet64.load_state_dict({k: v.double() for k, v in onet.state_dict().items()})
    onet64.train()
    pre64 = {}
    lin64 = dict(onet64.named_modules())['head.2.lins.0.lin']
    lin64.register_forward_hook(lambda m, i, o: pre64.setdefault('p', o.detach()))
    onet64(x.double())
    p64 = pre64['p']
    print('f64 pre-activations: smallest |value| %.3e, count |v| < 1e-4: %d of %d' % (p64.abs().min().item(), int((p64.abs() < 1e-4).sum()), p64.numel()))
    for name, env in (('w1', {'NNL_CONV_WINO2': '0'}), ('w2', {}), ('direct', {'NNL_CONV_WINO': '0'})):
        for k in ('NNL_CONV_WINO', 'NNL_CONV_WINO2'):
            os.environ.pop(k, None)
        os.environ.update(env); lib.nnl_reload_env()
        net = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
        synth.fill_module_(net, seed=5)
        net = net.cuda().train()
        got = {}
        mods = dict(net.named_modules())
        mods['head.2'].register_forward_hook(lambda m, i, o: got.setdefault('f', i[0].detach().double().cpu()))
        net(x.cuda())
        # (the product head runs fused kernels that bypass the sub-modules' hooks: the pre-activations are recomputed here in float64
        # from the FEATURES the product net fed its head — the part the convolution kernels influence)
        bn, lin = mods['head.2.pre_bn'], mods['head.2.lins.0.lin']
        h = torch.nn.functional.batch_norm(got['f'], None, None, bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu(), True, 0.1, bn.eps)
        p = torch.nn.functional.linear(h, lin.weight.detach().double().cpu(), lin.bias.detach().double().cpu())
        flips = (p > 0) != (p64 > 0)
        idx = flips.nonzero()
        print(name, 'max |pre - f64| %.3e; gate flips: %d' % ((p - p64).abs().max().item(), int(flips.sum())),
              [(int(a), int(b), float(p64[a, b]), float(p[a, b])) for a, b in idx[:5]])


if __name__ == '__main__':
    main()
