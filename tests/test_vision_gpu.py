"""GPU parity of the HIP-backed ResNet blocks / ResNet-34 classifier against the reference goldens (G5, G6) and the
CPU oracle.  All convolutions and linears go through the C ABI (libnnl_hip.so)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_close, load_golden
from oracle import synth
from test_vision_oracle import block_cases, check_g6, g6_inputs, run_block

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_g5_blocks_hip():
    from neuralnetworklibrary_amd.Applications.VisionModels import retinanet as PN
    from neuralnetworklibrary_amd.Applications.VisionModels.resnet import ResNetBody
    g = load_golden('g5_blocks')
    for tag, make, x in block_cases(PN):
        run_block(tag, make(), x, g, rtol=1e-4, atol=1e-5, dev=DEV)
    stem = ResNetBody(PN.HipConv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                      nn.MaxPool2d(3, stride=2, padding=1))
    run_block('stem', stem, synth.synth_input((2, 3, 32, 32), 4), g, rtol=1e-4, atol=1e-5, dev=DEV)


def _product_net(S=96, N=4):
    from neuralnetworklibrary_amd.Applications import Vision as V

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
    net = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
    synth.fill_module_(net)
    return net.to(DEV), D


def test_g6_resnet34_hip_forward_backward():
    g = load_golden('g6_resnet34')
    net, _ = _product_net()
    assert len(net.layer_groups) == int(g['n_layer_groups'])
    check_g6(net, g, dev=DEV)


def test_g6_resnet34_hip_learner_step():
    """product Learner.train1minibatch (SGD-momentum, lr per layer group, wd) == the reference's own step."""
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g6_resnet34')
    net, D = _product_net()
    x, y = g6_inputs(g, DEV)
    d = D(); d.train_dl = [(x, y)]; d.val_dl = [(x, y)]
    learner = Learner('/tmp/nnl_test_g6', d, net, optimizer='SGD_Mom')
    learner.init_optimizer(wd=1e-4)
    net.train()
    loss = learner.train1minibatch(x, y, [1e-3, 3e-3, 1e-2])
    assert_close(np.array([loss]), g['step_loss'], 5e-4, 1e-6, 'step loss')   # train-mode BN at N=4: see gen_golden g6
    abs_sums = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()])
    # training-mode BN at N=4 is ill-conditioned (gen_golden g6): the update differs by lr * (gradient noise)
    assert_close(abs_sums, g['after.abs_sums'], 2e-3, 1e-8, 'abs sums after step')


def test_g6_resnet34_hip_learner_step_bn_frozen():
    """The same step with every BatchNorm frozen (bn_freeze('all'), Learner.py:248-264,589-591): well conditioned, so
    the post-step parameters must match the reference's tightly."""
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g6_resnet34')
    net, D = _product_net()
    x, y = g6_inputs(g, DEV)
    d = D(); d.train_dl = [(x, y)]; d.val_dl = [(x, y)]
    learner = Learner('/tmp/nnl_test_g6', d, net, optimizer='SGD_Mom')
    learner.bn_freeze('all')
    learner.init_optimizer(wd=1e-4)
    net.train()
    learner._apply_bn_frozen()
    loss = learner.train1minibatch(x, y, [1e-3, 3e-3, 1e-2])
    assert_close(np.array([loss]), g['frozen.step_loss'], 1e-5, 1e-6, 'step loss')
    abs_sums = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()])
    sums = np.array([p.double().sum().item() for _, p in net.named_parameters()])
    assert_close(abs_sums, g['frozen.after.abs_sums'], 1e-6, 1e-8, 'abs sums after step')
    assert_close(sums, g['frozen.after.sums'], 1e-4, 1e-5 * np.abs(g['frozen.after.abs_sums']).max(), 'sums after step')


def test_resnet34_full_baseline_size_forward_backward_vs_oracle():
    """BASELINE configs[1] at full size — ResNet-34 + default head, 224x224, bs 64, BatchNorm in training mode (well
    conditioned at this batch: 3136+ values per channel): logits, loss (1e-4) and every parameter gradient of the HIP path
    against the fp32 CPU oracle on the same seeded weights and batch.  Exercises the balanced schedule, BK 16 / 32 tiles, split-K wgrad,
    the shortcut-gradient fusion, the BN bit masks and the pooling kernels at the sizes the benchmark runs."""
    from oracle import reference_nets as RNets
    N, S = 64, 224
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(N, 3, S, S, generator=g), torch.randint(0, 2, (N,), generator=g)
    onet = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.), probe_sz=(S, S))   # same constructor probe
    synth.fill_module_(onet, seed=5)
    net, _ = _product_net(S, N)
    synth.fill_module_(net, seed=5)
    onet.train(); net.train()
    lo = onet(x); loss_o = nn.CrossEntropyLoss()(lo, y); loss_o.backward()
    lp = net(x.to(DEV)); loss_p = nn.CrossEntropyLoss()(lp, y.to(DEV)); loss_p.backward()
    assert_close(lp, lo.detach(), 1e-3, 1e-4 * lo.detach().abs().max().item(), 'logits')
    assert_close(loss_p, loss_o.detach(), 1e-4, 0, 'loss')
    worst = 0.0
    for (n, po), (_, pp) in zip(onet.named_parameters(), net.named_parameters()):
        go, gp = po.grad.double(), pp.grad.detach().cpu().double()
        rel = (gp - go).norm().item() / max(go.norm().item(), 1e-12)
        worst = max(worst, rel)
        # ReLU makes the gradient discontinuous: of the ~1e8 activations of this step a few dozen sit within rounding of zero
        # and get the opposite gate in two correct fp32 implementations (checked in isolation: BN+ReLU backward of one
        # [64,512,7,7] tensor differs from torch CPU by 8e-4 in norm because ~2 of 1.6 M gates flip, while the masked values
        # themselves are exact).  The flips accumulate to <1e-2 towards the stem; the head (no ReLU behind it) matches to 1e-4.
        assert rel < (1e-3 if n.startswith('head') else 2e-2), '%s: relative gradient error %.3e' % (n, rel)
    for (n, bo), (_, bp) in zip(onet.named_buffers(), net.named_buffers()):
        assert_close(bp, bo, 1e-4, 1e-5, 'buffer ' + n)
    print('worst relative gradient error %.2e' % worst)
