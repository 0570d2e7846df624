"""Diagnostic: per-parameter gradient error of the product ObjectDetectionNet (HIP) and of the fp32 CPU oracle against the fp64
CPU oracle on the G12 inputs (train-mode pass first, then eval mode, as the golden generator runs them)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reference_math as RM, reference_nets as RN, synth  # noqa: E402
from neuralnetworklibrary_amd.Applications import Vision as V  # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g12_objectdetectionnet.npz'), allow_pickle=True)
N, S, K = int(g['N']), int(g['S']), int(g['K'])
x = synth.synth_input((N, 3, S, S), 12)
B, Cc = torch.from_numpy(g['boxes']), torch.from_numpy(g['cats'])
nets = {}
for tag, dt in (('c32', torch.float32), ('c64', torch.float64)):
    n = RN.ObjectDetectionNet(K); synth.fill_detection_net_(n); nets[tag] = n.to(dt)
torch.manual_seed(0)
hip = V.ObjectDetectionNet(K); synth.fill_detection_net_(hip); hip = hip.cuda()
for mode in ['train', 'eval']:
    grads = {}
    for tag, n in nets.items():
        n.train() if mode == 'train' else n.eval()
        for p in n.parameters():
            p.grad = None
        dt = next(n.parameters()).dtype
        a, r, c = n(x.to(dt))
        RM.ssd_loss(a, r, c, B.to(dt), Cc)[0].backward()
        grads[tag] = [p.grad.double() for p in n.parameters()]
        grads[tag + '_out'] = (r.detach().double(), c.detach().double())
    hip.train() if mode == 'train' else hip.eval()
    for p in hip.parameters():
        p.grad = None
    a, r, c = hip(x.cuda())
    V.SSD_loss(0.5, 0.25, 2.0)([a, r, c], [B.cuda(), Cc.cuda()]).backward()
    gh = [p.grad.detach().cpu().double() for p in hip.parameters()]
    r64, c64 = grads['c64_out']
    print(mode, 'reg err hip %.2e cpu32 %.2e | clas err hip %.2e cpu32 %.2e' % (
        (r.detach().cpu().double() - r64).abs().max(), (grads['c32_out'][0] - r64).abs().max(),
        (c.detach().cpu().double() - c64).abs().max(), (grads['c32_out'][1] - c64).abs().max()))
    rows = []
    for (name, _), h, c32, c64 in zip(hip.named_parameters(), gh, grads['c32'], grads['c64']):
        ref = c64.norm().item()
        rows.append(((h - c64).norm().item() / max(ref, 1e-300), (c32 - c64).norm().item() / max(ref, 1e-300), name, ref))
    rows.sort(reverse=True)
    for e_h, e_c, name, ref in rows[:12]:
        print('  %-40s hip %.2e cpu32 %.2e |g64| %.2e' % (name, e_h, e_c, ref))
    print('  median hip %.2e cpu32 %.2e' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
    if mode == 'eval':
        names = [n for n, _ in hip.named_parameters()]
        for sl in [str(s) for s in g['slice_names']]:
            key = 'eval.grad.%s' % sl
            if key + '.f64' not in g:
                continue
            i = names.index(sl)
            r64, r32 = g[key + '.f64'], g[key + '.f32']
            cut = lambda t: t.reshape(-1)[:1024].numpy()
            h, c32, c64 = cut(gh[i]), cut(grads['c32'][i]), cut(grads['c64'][i])
            nr = np.linalg.norm(r64)
            j = int(np.abs(h - r64).argmax())
            print('  slice %-28s |r64| %.2e  hip-r64 %.2e  c32-r64 %.2e  c64-r64 %.2e  r32-r64 %.2e  hip-c32 %.2e | worst elem %d: hip %.6e r64 %.6e c32 %.6e' % (
                sl, nr, np.linalg.norm(h - r64) / nr, np.linalg.norm(c32 - r64) / nr, np.linalg.norm(c64 - r64) / nr,
                np.linalg.norm(r32 - r64) / nr, np.linalg.norm(h - c32) / nr, j, h[j], r64[j], c32[j]))
