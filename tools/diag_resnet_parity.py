"""Diagnostic: ResNet-34 + head, 224x224, bs 64 — per-parameter gradient error of the HIP path and of the fp32 CPU oracle against
the fp64 CPU oracle (one forward / backward, BatchNorm in training mode), and the loss after ONE SGD step."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import reference_nets as RN, synth  # noqa: E402
from neuralnetworklibrary_amd.Applications import Vision as V  # noqa: E402

N, S = 64, 224
x = synth.synth_input((N, 3, S, S), 130)
y = (torch.arange(N) * 7) % 2


class D:
    sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'


o32 = RN.ImageClassificationNet(RN.resnet34(), 2, 512, drops=(0., 0.), probe_sz=(S, S))
synth.fill_module_(o32, seed=5)
o64 = RN.ImageClassificationNet(RN.resnet34(), 2, 512, drops=(0., 0.)).double()
o64.load_state_dict({k: v.double() for k, v in o32.state_dict().items()})
net = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
synth.fill_module_(net, seed=5)
net = net.cuda()
for m in (o32, o64, net):
    m.train()
lp = nn.CrossEntropyLoss()(net(x.cuda()), y.cuda()); lp.backward()
l32 = nn.CrossEntropyLoss()(o32(x), y); l32.backward()
l64 = nn.CrossEntropyLoss()(o64(x.double()), y); l64.backward()
print('loss hip %.9f c32 %.9f c64 %.9f' % (lp.item(), l32.item(), l64.item()))
rows = []
for (n, pp), (_, p32), (_, p64) in zip(net.named_parameters(), o32.named_parameters(), o64.named_parameters()):
    g64 = p64.grad
    ref = g64.norm().item()
    rows.append(((pp.grad.cpu().double() - g64).norm().item() / ref, (p32.grad.double() - g64).norm().item() / ref, n))
for e_h, e_c, n in rows:
    print('%-36s hip %.2e cpu32 %.2e ratio %.2f' % (n, e_h, e_c, e_h / e_c))
r = np.array([[a, b] for a, b, _ in rows])
print('median hip %.2e cpu32 %.2e; geo-mean ratio %.2f' % (np.median(r[:, 0]), np.median(r[:, 1]), np.exp(np.mean(np.log(r[:, 0] / r[:, 1])))))
# is the error systematic?  projection of each implementation's error on the exact gradient, and the two errors on each other
tot = {'hh': 0.0, 'cc': 0.0, 'hc': 0.0, 'hg': 0.0, 'cg': 0.0, 'gg': 0.0}
for (n, pp), (_, p32), (_, p64) in zip(net.named_parameters(), o32.named_parameters(), o64.named_parameters()):
    if not n.startswith('body'):
        continue
    g64 = p64.grad.reshape(-1)
    dh, dc = pp.grad.cpu().double().reshape(-1) - g64, p32.grad.double().reshape(-1) - g64
    tot['hh'] += (dh @ dh).item(); tot['cc'] += (dc @ dc).item(); tot['hc'] += (dh @ dc).item()
    tot['hg'] += (dh @ g64).item(); tot['cg'] += (dc @ g64).item(); tot['gg'] += (g64 @ g64).item()
print('body: |dh|/|g| %.3e |dc|/|g| %.3e cos(dh,dc) %.3f  <dh,g>/|g|^2 %.3e  <dc,g>/|g|^2 %.3e' % (
    (tot['hh'] / tot['gg']) ** .5, (tot['cc'] / tot['gg']) ** .5, tot['hc'] / (tot['hh'] * tot['cc']) ** .5,
    tot['hg'] / tot['gg'], tot['cg'] / tot['gg']))
# activations entering the head and logits
with torch.no_grad():
    fh = net.body(x.cuda()).cpu().double(); f32 = o32.body(x).double(); f64 = o64.body(x.double())
print('body output: hip err %.3e cpu32 err %.3e (rel. max)' % ((fh - f64).abs().max() / f64.abs().max(), (f32 - f64).abs().max() / f64.abs().max()))
