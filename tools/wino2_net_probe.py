"""Where does the 2-D Winograd dispatch change the full-size ResNet-34 forward?  Block outputs of the product net under three kernel
settings, pairwise relative L2 differences.   python tools/wino2_net_probe.py"""
import os
import sys

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import synth
from neuralnetworklibrary_amd._lib import lib


def main():
    from neuralnetworklibrary_amd.Applications import Vision as V
    N, S = 64, 224

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, 3, S, S, generator=g).cuda()
    outs = {}
    for name, env in (('direct', {'NNL_CONV_WINO': '0'}), ('w1', {'NNL_CONV_WINO2': '0'}), ('w2', {})):
        for k in ('NNL_CONV_WINO', 'NNL_CONV_WINO2'):
            os.environ.pop(k, None)
        os.environ.update(env); lib.nnl_reload_env()
        net = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
        synth.fill_module_(net, seed=5)
        net = net.cuda().train()
        acts = []
        hooks = []
        for nm, m in net.named_modules():
            if nm.count('.') == 2 and nm.startswith('body.') or nm in ('body.0', 'body.3'):
                hooks.append(m.register_forward_hook(lambda mod, i, o, nm=nm: acts.append((nm, o.detach().float().clone()))))
        with torch.enable_grad():
            y = net(x)
        acts.append(('logits', y.detach().clone()))
        outs[name] = acts
        for h in hooks:
            h.remove()
    for (n0, a), (_, b), (_, c) in zip(outs['direct'], outs['w1'], outs['w2']):
        d1 = ((b - a).norm() / a.norm()).item(); d2 = ((c - a).norm() / a.norm()).item()
        print('%-12s |w1-direct| %.2e  |w2-direct| %.2e  max %.2e %.2e' % (n0, d1, d2, (b - a).abs().max().item(), (c - a).abs().max().item()))


if __name__ == '__main__':
    main()
