"""Diagnostic: first steps of the G13 protocol (ResNet-34 224x224 bs 64, reference Learner curve in fp32 / fp64 as golden) on the
HIP path under a switch given on the command line, e.g.  NNL_BN_EPI_STATS=0 python tools/diag_g13.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import synth  # noqa: E402
from neuralnetworklibrary_amd.Applications import Vision as V  # noqa: E402
from neuralnetworklibrary_amd.General.Learner import Learner  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g13_resnet34_curve.npz'), allow_pickle=True)
N, S, steps = 64, 224, int(sys.argv[1]) if len(sys.argv) > 1 else 5


class D:
    sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'


net = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
synth.fill_module_(net, seed=5)
net = net.cuda()
xs = [synth.synth_input((N, 3, S, S), 130 + b).cuda() for b in range(4)]
ys = [((torch.arange(N) * 7 + b) % 2).cuda() for b in range(4)]
d = D(); d.train_dl = [(xs[0], ys[0])]; d.val_dl = d.train_dl
learner = Learner('/tmp/nnl_diag_g13', d, net, optimizer='SGD_Mom')
learner.init_optimizer(wd=1e-4)
net.train()
losses = np.array([learner.train1minibatch(xs[i % 4], ys[i % 4], [1e-3, 3e-3, 1e-2]) for i in range(steps)])
r64, r32 = g['losses.f64'][:steps], g['losses.f32'][:steps]
print({k: os.environ[k] for k in os.environ if k.startswith('NNL_')})
print('rel |hip-f64|  ', np.array2string(np.abs(losses - r64) / r64, precision=2))
print('rel |ref32-f64|', np.array2string(np.abs(r32 - r64) / r64, precision=2))
# where does the step-1 deviation come from?  (a) the forward at step 1, or (b) the parameters after step 0
if os.environ.get('NNL_DIAG_DEEP') == '1':
    import torch.nn as nn
    from oracle import reference_math as RM, reference_nets as RNets
    net2 = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
    synth.fill_module_(net2, seed=5)
    net2 = net2.cuda().train()
    d2 = D(); d2.train_dl = [(xs[0], ys[0])]; d2.val_dl = d2.train_dl
    l2 = Learner('/tmp/nnl_diag_g13', d2, net2, optimizer='SGD_Mom')
    l2.init_optimizer(wd=1e-4)
    l2.train1minibatch(xs[0], ys[0], [1e-3, 3e-3, 1e-2])
    hip_loss1 = nn.CrossEntropyLoss()(net2(xs[1]), ys[1]).item()           # HIP forward at step 1 (training-mode BN)
    o64 = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.)).double().train()
    o64.load_state_dict({k: v.detach().cpu().double() for k, v in net2.state_dict().items()})
    f64_loss_on_hip_params = nn.CrossEntropyLoss()(o64(xs[1].cpu().double()), ys[1].cpu()).item()
    print('step-1 loss: HIP forward %.9f | fp64 forward on HIP post-step-0 params %.9f | golden f64 %.9f' % (
        hip_loss1, f64_loss_on_hip_params, g['losses.f64'][1]))
    # parameters after step 0: HIP vs an fp64 oracle step
    o = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.), probe_sz=(S, S))
    synth.fill_module_(o, seed=5)
    res = {}
    for tag, dt in (('c32', torch.float32), ('c64', torch.float64)):
        m = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.)).to(dt).train()
        m.load_state_dict({k: v.to(dt) for k, v in o.state_dict().items()})
        names = [n for n, _ in m.named_parameters()]
        params = [p for _, p in m.named_parameters()]
        group = lambda n: 2 if n.startswith('head') else (0 if int(n.split('.')[1]) < 6 else 1)
        lrs = [[1e-3, 3e-3, 1e-2][group(n)] for n in names]
        nn.CrossEntropyLoss()(m(xs[0].cpu().to(dt)), ys[0].cpu()).backward()
        RM.optimizer_step(params, [p.grad for p in params], RM.OptimState(params), lrs, [1e-4] * len(params), 'sgd')
        res[tag] = params
    p0 = dict(o.named_parameters())
    rows = []
    for (n, ph), p32, p64 in zip(net2.named_parameters(), res['c32'], res['c64']):
        upd = (p64.detach() - p0[n].detach().double()).norm().item()
        rows.append(((ph.detach().cpu().double() - p64.detach()).norm().item() / upd, (p32.detach().double() - p64.detach()).norm().item() / upd, n))
    rows.sort(reverse=True)
    print('error of the step-0 UPDATE relative to the update size (worst 10 by HIP):')
    for e_h, e_c, n in rows[:10]:
        print('  %-34s hip %.2e cpu32 %.2e' % (n, e_h, e_c))
    print('  median hip %.2e cpu32 %.2e' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
