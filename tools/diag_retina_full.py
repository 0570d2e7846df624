"""Diagnostic: ObjectDetectionNet(20) at 512x512, 2 images — per-parameter gradient error of HIP and of torch-CPU fp32 vs fp64."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_math as RM, reference_nets as RN, synth  # noqa: E402
from neuralnetworklibrary_amd.Applications import Vision as V  # noqa: E402

K, N, S = 20, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = synth.synth_input((N, 3, S, S), 77)
boxes = -np.ones((N, 4, 4), np.float32); cats = -np.ones((N, 4), np.int64)
boxes[0, :3] = [[30, 40, 200, 260], [250, 100, 420, 300], [100, 300, 180, 380]]; cats[0, :3] = [3, 17, 0]
boxes[1, :2] = [[60, 60, 460, 440], [10, 400, 90, 500]]; cats[1, :2] = [9, 19]
B, Cc = torch.from_numpy(boxes), torch.from_numpy(cats)
o32 = synth.fill_detection_net_(RN.ObjectDetectionNet(K), seed=3).train()
o64 = synth.fill_detection_net_(RN.ObjectDetectionNet(K), seed=3).double().train()
torch.manual_seed(0)
net = synth.fill_detection_net_(V.ObjectDetectionNet(K), seed=3).cuda().train()
if len(sys.argv) > 2 and sys.argv[2] == 'eval':
    o32.eval(); o64.eval(); net.eval()
anchors, reg, clas = net(x.cuda())
reg.retain_grad(); clas.retain_grad()
V.SSD_loss(0.5, 0.25, 2.0)([anchors, reg, clas], [B.cuda(), Cc.cuda()]).backward()
a32, r32, c32 = o32(x); r32.retain_grad(); c32.retain_grad()
RM.ssd_loss(a32, r32, c32, B, Cc, 0.5, 0.25, 2.0)[0].backward()
a64, r64, c64 = o64(x.double()); r64.retain_grad(); c64.retain_grad()
RM.ssd_loss(a64, r64, c64, B.double(), Cc, 0.5, 0.25, 2.0)[0].backward()
rel = lambda a, b: ((a.detach().cpu().double() - b.detach()).norm() / b.detach().norm()).item()
print('reg  hip %.2e cpu32 %.2e | clas hip %.2e cpu32 %.2e' % (rel(reg, r64), rel(r32, r64), rel(clas, c64), rel(c32, c64)))
print('dreg hip %.2e cpu32 %.2e | dclas hip %.2e cpu32 %.2e' % (rel(reg.grad, r64.grad), rel(r32.grad, r64.grad), rel(clas.grad, c64.grad), rel(c32.grad, c64.grad)))
rows = []
for (n, pp), (_, p32), (_, p64) in zip(net.named_parameters(), o32.named_parameters(), o64.named_parameters()):
    ref = p64.grad.norm().item()
    rows.append(((pp.grad.cpu().double() - p64.grad).norm().item() / ref, (p32.grad.double() - p64.grad).norm().item() / ref, n))
for e_h, e_c, n in rows:
    if n.startswith(('fpn', 'classifier', 'regressor')) or e_h > 3 * e_c + 1e-3:
        print('%-36s hip %.2e cpu32 %.2e ratio %.1f' % (n, e_h, e_c, e_h / e_c))
