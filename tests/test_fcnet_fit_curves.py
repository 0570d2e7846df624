"""Golden G4 (FullyConnectedNet with injected dropout masks) and G10 (20-step reference Learner.fit loss curves for the
collaborative-filtering and structured-data heads) — SURVEY.md §8c.  CPU: the oracle restatement; GPU: the product modules and
the product Learner.fit."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import T, assert_close, load_golden
from oracle import reference_math as RM
from oracle import reference_nets as RNets
from oracle import synth

DEV = 'cuda'


class _Mask(nn.Module):
    "stands in for an nn.Dropout: multiplies by a pre-drawn (already 1/(1-p)-scaled) mask"
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, t):
        return t * self.m.to(t.device)


def _check_g4(net, g, dev, rtol):
    synth.fill_module_(net, seed=14)
    net = net.to(dev)
    net.lins[0].drop, net.lins[1].drop, net.final_drop = (_Mask(T(g['g4.mask%d' % i], dev)) for i in range(3))
    net.train()
    assert [n for n, _ in net.named_parameters()] == [str(s) for s in g['g4.param_names']]
    x = synth.synth_input((12, 20), 144).to(dev).requires_grad_(True)
    y = (torch.arange(12) % 3).to(dev)
    logits = net(x)
    loss = nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    assert_close(logits, g['g4.logits'], rtol, 1e-5, 'logits'); assert_close(loss, g['g4.loss'], rtol, 1e-6, 'loss')
    assert_close(x.grad, g['g4.dx'], rtol * 10, 1e-6, 'dx')
    for n, p in net.named_parameters():
        assert_close(p.grad, g['g4.grad.' + n], rtol * 10, 1e-6, 'grad ' + n)
    for n, b in net.named_buffers():
        assert_close(b, g['g4.buf.' + n], rtol, 1e-6, 'buffer ' + n)


def test_g4_fcnet_oracle():
    _check_g4(RNets.FullyConnectedNet([20, 16, 8, 3], [0.3, 0.2, 0.1]), load_golden('g4_g10_fcnet_fit'), 'cpu', 1e-5)


@pytest.mark.gpu
def test_g4_fcnet_hip():
    from neuralnetworklibrary_amd.General.Layers import FullyConnectedNet
    _check_g4(FullyConnectedNet([20, 16, 8, 3], [0.3, 0.2, 0.1]), load_golden('g4_g10_fcnet_fit'), DEV, 1e-4)


# ---- G10 ---------------------------------------------------------------------------------------------------------------------
def _collab_batches(g, dev='cpu'):
    b = [(T(g['g10.collab.x%d' % i], dev), T(g['g10.collab.y%d' % i], dev)) for i in range(13)]
    return b[:10], b[10:]


def _tab_batches(g, dev='cpu'):
    b = [([T(g['g10.tab.xcat%d' % i], dev), T(g['g10.tab.xcont%d' % i], dev)], T(g['g10.tab.y%d' % i], dev)) for i in range(13)]
    return b[:10], b[10:]


def _oracle_fit(net, fwd, tr, va, lrs_of, wd, epochs=2):
    "restated Learner.fit with constant lr: per-step losses + size-weighted validation loss before / after"
    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    state = RM.OptimState(params)
    lrs = [lrs_of(n) for n in names]

    def val():
        net.eval()
        with torch.no_grad():
            tot = sum(len(y) * nn.MSELoss()(fwd(net, x), y).item() for x, y in va)
        net.train()
        return tot / sum(len(y) for _, y in va)
    pre = val()
    losses = []
    net.train()
    for _ in range(epochs):
        for x, y in tr:
            for p in params:
                p.grad = None
            loss = nn.MSELoss()(fwd(net, x), y)
            loss.backward()
            losses.append(loss.item())
            RM.optimizer_step(params, [p.grad for p in params], state, lrs, [wd] * len(params), 'adam')
    return np.array(losses), pre, val()


def test_g10_fit_curves_oracle():
    g = load_golden('g4_g10_fcnet_fit')
    tr, va = _collab_batches(g)
    net = synth.fill_module_(RNets.CollabFilterNet(30, 20, 6, [0.8, 5.2]), seed=15)
    losses, pre, post = _oracle_fit(net, lambda n, x: n(x), tr, va, lambda n: 2e-2, 1e-4)
    assert_close(losses, g['g10.collab.loss_sched'], 1e-5, 1e-7, 'collab loss_sched')
    assert_close(np.array([pre, post]), np.array([g['g10.collab.val_pre'][0], g['g10.collab.val_post'][0]]), 1e-5, 1e-7, 'collab val')
    tr, va = _tab_batches(g)
    dims = [int(d) for d in g['g10.tab.emb_dims']]
    net = synth.fill_module_(RNets.StructuredDataNet('cont', list(zip([7, 5, 4], dims)), 3, [16, 8, 1], output_range=[5, 12]), seed=16)
    losses, pre, post = _oracle_fit(net, lambda n, x: n(x[0], x[1]), tr, va, lambda n: 2e-2 if n.startswith('head') else 1e-2, 1e-3)
    assert_close(losses, g['g10.tab.loss_sched'], 1e-5, 1e-7, 'tabular loss_sched')
    assert_close(np.array([pre, post]), np.array([g['g10.tab.val_pre'][0], g['g10.tab.val_post'][0]]), 1e-5, 1e-7, 'tabular val')


class _Data:
    def __init__(self, tr, va, bs):
        self.train_dl, self.val_dl, self.bs, self.target_type = tr, va, bs, 'cont'


@pytest.mark.gpu
def test_g10_fit_curves_hip_learner():
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    g = load_golden('g4_g10_fcnet_fit')
    tr, va = _collab_batches(g, DEV)
    net = synth.fill_module_(CollabFilterNet(30, 20, 6, [0.8, 5.2]), seed=15)
    learner = Learner('/tmp/nnl_g10', _Data(tr, va, 16), net, optimizer='Adam')
    pre = learner.evaluate('val')[0]
    learner.fit(2e-2, 2, wd=1e-4)
    post = learner.evaluate('val')[0]
    assert_close(np.array(learner.loss_sched), g['g10.collab.loss_sched'], 1e-4, 1e-6, 'collab loss_sched')
    assert_close(np.array([pre, post]), np.array([g['g10.collab.val_pre'][0], g['g10.collab.val_post'][0]]), 1e-4, 1e-6, 'collab val')
    tr, va = _tab_batches(g, DEV)
    net = synth.fill_module_(StructuredDataNet('cont', 3, 3, [{i: i for i in range(c)} for c in [7, 5, 4]], [16, 8, 1], output_range=[5, 12]), seed=16)
    learner = Learner('/tmp/nnl_g10', _Data(tr, va, 16), net, optimizer='Adam')
    pre = learner.evaluate('val')[0]
    learner.fit([1e-2, 2e-2], 2, wd=1e-3)
    post = learner.evaluate('val')[0]
    assert_close(np.array(learner.loss_sched), g['g10.tab.loss_sched'], 1e-3, 1e-6, 'tabular loss_sched')
    assert_close(np.array([pre, post]), np.array([g['g10.tab.val_pre'][0], g['g10.tab.val_post'][0]]), 1e-3, 1e-6, 'tabular val')
