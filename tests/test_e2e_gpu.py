"""End-to-end drop-in flows on the GPU through the public API only: data object -> Learner -> fit / evaluate / predict, with
the MI355X opt-ins (device-resident loaders, whole-step hipGraph replay) switched on.  The loss must go down and the
graph / eager runs must agree on what they learn."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _setup():
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False


def _ratings(n=4000, n_user=60, n_item=40, seed=0):
    rs = np.random.RandomState(seed)
    pu, pi = rs.standard_normal((n_user, 3)), rs.standard_normal((n_item, 3))
    u, i = rs.randint(0, n_user, n), rs.randint(0, n_item, n)
    r = np.clip(3 + (pu[u] * pi[i]).sum(1) + 0.1 * rs.standard_normal(n), 1, 5).astype('float32')
    return pd.DataFrame({'user': u, 'item': i, 'rating': r})


@pytest.mark.parametrize('graphs', [False, True])
def test_collab_fit_device_resident(graphs):
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterDataObj, CollabFilterNet
    from neuralnetworklibrary_amd.General.Learner import Learner
    _setup()
    df = _ratings()
    labels = [{u: k for k, u in enumerate(sorted(df.user.unique()))}, {m: k for k, m in enumerate(sorted(df.item.unique()))}]
    data = CollabFilterDataObj(df[:3200], df[3200:], 'user', 'item', 'rating', labels, bs=256, device_resident=True, seed=1)
    torch.manual_seed(0)
    net = CollabFilterNet(len(labels[0]), len(labels[1]), 8, [0.8, 5.2])
    learner = Learner('/tmp/nnl_e2e', data, net, optimizer='Adam')
    if graphs:
        learner.use_graphs(True)
    before = learner.evaluate('val')[0]
    learner.fit(5e-2, 6, wd=1e-5)
    after = learner.evaluate('val')[0]
    assert after < 0.5 * before, (before, after)
    pred = learner.predict('val')
    assert pred.shape == (800,) and np.isfinite(pred).all() and 0.8 <= pred.min() and pred.max() <= 5.2
    if graphs:
        assert sum(g.graph is not None for g in learner._graphs.values()) >= 1


def test_tabular_fit_device_resident_with_graphs():
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet, StructuredDataObj, StructuredDataset
    from neuralnetworklibrary_amd.General.Learner import Learner
    _setup()
    rs = np.random.RandomState(2)
    N, cards = 3000, [7, 5, 11]
    xcat = np.stack([rs.randint(0, c, N) for c in cards], 1)
    xcont = rs.standard_normal((N, 4)).astype('float32')
    eff = [rs.standard_normal(c) for c in cards]
    y = (sum(e[xcat[:, j]] for j, e in enumerate(eff)) + xcont[:, 0] - 0.5 * xcont[:, 1] + 0.05 * rs.standard_normal(N)).astype('float32')
    tr = StructuredDataset(pd.DataFrame(xcat[:2400]), pd.DataFrame(xcont[:2400]), y[:2400], 'cont')
    va = StructuredDataset(pd.DataFrame(xcat[2400:]), pd.DataFrame(xcont[2400:]), y[2400:], 'cont')
    labels = [{k: k for k in range(c)} for c in cards]
    data = StructuredDataObj(tr, va, labels, None, bs=200, device_resident=True, seed=3)
    torch.manual_seed(0)
    net = StructuredDataNet('cont', 3, 4, labels, [64, 32, 1], dropout_levels=(0.02, 0.02, [0, 0.1, 0.05]))
    learner = Learner('/tmp/nnl_e2e', data, net, optimizer='Adam').use_graphs(True)
    before = learner.evaluate('val')[0]
    learner.fit_one_cycle([3e-3, 3e-3], 8, wd=1e-4)
    after = learner.evaluate('val')[0]
    assert after < 0.25 * before, (before, after)
    assert len(learner.loss_sched) == 8 * len(data.train_dl)
