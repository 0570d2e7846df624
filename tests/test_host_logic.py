"""Host logic of the drop-in API on CPU (no HIP): the product's Optimizer / Learner (schedules, fit, fit_cycles,
fit_one_cycle, find_lr, evaluate, ragged-batch lr scaling, EMA loss) driving a plain torch model, against goldens
produced by the reference's own Learner/Optimizer on the same toy problem (G9/G10)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import T, assert_close, load_golden
from oracle import synth


@pytest.fixture(autouse=True)
def _cpu_device():
    from neuralnetworklibrary_amd.General import Core
    from neuralnetworklibrary_amd.General.Learner import Learner
    old = Core._DEVICE
    Core.set_default_device('cpu')
    Learner.verbose = False
    yield
    Core._DEVICE = old


def toy():
    from neuralnetworklibrary_amd.General.Core import separate_bn_layers
    g1 = nn.Sequential(nn.Linear(5, 7), nn.BatchNorm1d(7), nn.Tanh())
    g2 = nn.Sequential(nn.Linear(7, 1), nn.Flatten(0))
    net = nn.Sequential(g1, g2)
    synth.fill_module_(net, seed=11)
    net.layer_groups = [g1, g2]
    net.param_groups = separate_bn_layers(net.layer_groups)
    return net


class Data:
    target_type = 'cont'

    def __init__(self, g):
        X, Y = T(g['X']), T(g['Y'])
        self.train_dl = [(X[i:i + 8], Y[i:i + 8]) for i in range(0, 36, 8)]
        self.val_dl = self.train_dl[:2]
        self.bs = 8


def flat(net):
    return np.concatenate([p.detach().numpy().reshape(-1) for p in net.parameters()])


def test_param_group_contract():
    net = toy()
    assert len(net.param_groups) == 4                                  # [reg_1, reg_2, bn_1, bn_2]
    assert [type(m).__name__ for m in net.param_groups[2]] == ['BatchNorm1d'] and len(net.param_groups[3]) == 0


@pytest.mark.parametrize('tag,opt,kw', [('sgd', 'SGD_Mom', dict(wd=[1e-2, 3e-2], bn_wd=True, clip=0.5)),
                                        ('adam', 'Adam', dict(wd=1e-2, bn_wd=False, clip=None))])
def test_optimizer_step(tag, opt, kw):
    from neuralnetworklibrary_amd.General.Learner import opt_dict
    from neuralnetworklibrary_amd.General.Optimizer import Optimizer
    g = load_golden('g9_host_logic')
    net = toy()
    o = Optimizer(opt_dict[opt], net)
    o.set_params([1e-1, 3e-1], **kw)
    X, Y = T(g['X']), T(g['Y'])
    for _ in range(3):
        o.opt.zero_grad()
        ((net(X[:8]) - Y[:8]) ** 2).mean().backward()
        o.step()
    for n, p in net.named_parameters():
        assert_close(p, g['opt.%s.%s' % (tag, n)], 1e-5, 1e-7, n)


def test_get_sched():
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    for st in ['linear', 'cos', 'exp', 'poly']:
        np.testing.assert_allclose(np.array(Learner.get_sched(st, 7, 1e-3, 1e-1)), g['sched.%s.scalar' % st], rtol=1e-12)
        np.testing.assert_allclose(np.array(Learner.get_sched(st, 5, [1e-3, 2e-3], [1e-1, 4e-1])), g['sched.%s.vector' % st], rtol=1e-12)


@pytest.mark.parametrize('opt', ['SGD_Mom', 'Adam'])
def test_fit_one_cycle_schedules(opt):
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    learner = Learner('/tmp/nnl_test_g9', Data(g), toy(), optimizer=opt)
    cap = {}
    learner.train_gen_sched = lambda lr, mom, betas, *a, **k: cap.update(lr=lr, mom=mom, betas=betas)
    learner.fit_one_cycle([1e-2, 3e-2], 2, wd=1e-3)
    np.testing.assert_allclose(np.array(cap['lr']), g['onecycle.%s.lr' % opt], rtol=1e-12)
    if opt == 'SGD_Mom':
        np.testing.assert_allclose(np.array(cap['mom']), g['onecycle.%s.mom' % opt], rtol=1e-12)
        assert cap['betas'] is None
    else:
        np.testing.assert_allclose(np.array(cap['betas']), g['onecycle.%s.betas' % opt], rtol=1e-12)
        assert cap['mom'] is None


def test_fit_loss_curve_ragged_batch_and_ema():
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    learner = Learner('/tmp/nnl_test_g9', Data(g), toy(), optimizer='SGD_Mom')
    learner.fit([3e-2, 1e-1], 2, wd=1e-3, clip=1.0, momentum=0.8)
    assert_close(np.array(learner.loss_sched), g['fit.loss_sched'], 1e-5, 1e-7, 'loss curve')
    assert_close(np.array([learner.moving_avg_loss]), g['fit.moving_avg'], 1e-5, 1e-7, 'EMA loss')
    assert_close(flat(learner.model), g['fit.w'], 1e-5, 1e-7, 'weights')
    assert len(learner.lr_sched) == 10 and learner.mom_sched == [0.8] * 10


def test_fit_cycles_loss_curve():
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    learner = Learner('/tmp/nnl_test_g9', Data(g), toy(), optimizer='Adam')
    learner.fit_cycles(3e-2, 1e-3, 2, cycle_type='cos', base_length=1, cycle_mult=2, wd=1e-3, betas=(0.8, 0.99))
    np.testing.assert_allclose(np.array(learner.lr_sched), g['cycles.lr_sched'], rtol=1e-12)
    assert_close(np.array(learner.loss_sched), g['cycles.loss_sched'], 1e-5, 1e-7, 'loss curve')
    assert_close(flat(learner.model), g['cycles.w'], 1e-4, 1e-6, 'weights')


def test_find_lr_restores_state_and_evaluate():
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    learner = Learner('/tmp/nnl_test_g9', Data(g), toy(), optimizer='SGD_Mom')
    before = flat(learner.model)
    learner.find_lr(lr_min=1e-4, lr_max=1.0, length=8, break_fac=None, plot=False)
    np.testing.assert_allclose(np.array(learner.lr_sched), g['findlr.lr_sched'], rtol=1e-12)
    assert_close(np.array(learner.loss_sched), g['findlr.loss_sched'], 1e-5, 1e-7, 'loss curve')
    assert np.abs(flat(learner.model) - before).max() == 0.0 == float(g['findlr.restored'][0])
    assert_close(np.array(learner.evaluate('val')[0:1]), g['evaluate.val'], 1e-5, 1e-7, 'evaluate')


def test_lr_list_length_is_validated_like_the_reference():
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    learner = Learner('/tmp/nnl_test_g9', Data(g), toy())
    with pytest.raises(ValueError):
        learner.fit([1e-2, 1e-2, 1e-2], 1)
    with pytest.raises(ValueError):
        learner.train_gen_sched([1e-2] * 7, None, None)           # not a multiple of len(train_dl)


def test_checkpoint_round_trip_keys():
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g9_host_logic')
    learner = Learner('/tmp/nnl_test_g9', Data(g), toy(), optimizer='Adam')
    learner.fit(1e-2, 1)
    learner.save('ck', save_optimizer=True)
    state = torch.load('/tmp/nnl_test_g9/models/ck.pt')
    assert set(state) == {'model_state', 'optimizer_state'}
    w = flat(learner.model)
    learner.fit(1e-1, 1)
    learner.load('ck', saved_optimizer=True)
    assert np.abs(flat(learner.model) - w).max() == 0.0


def test_predict_and_evaluate_single_sync_paths_match_per_batch_arithmetic():
    """Learner.predict / evaluate keep per-batch results on the device and copy once: same values as the reference's
    per-minibatch ARR(...) / .item() arithmetic (General/Learner.py:286-485)."""
    import torch.nn.functional as F
    from neuralnetworklibrary_amd.General.Core import make_model_basic
    from neuralnetworklibrary_amd.General.Learner import Learner
    Learner.verbose = False
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randn(n, 6, generator=g), torch.randint(0, 3, (n,), generator=g)) for n in (8, 8, 5)]
    class D:
        target_type, bs, categories = 'single_label', 8, ['a', 'b', 'c']
        train_dl = val_dl = batches
    d = D()
    torch.manual_seed(1)
    net = make_model_basic(nn.Sequential(nn.Linear(6, 10), nn.Tanh(), nn.Linear(10, 3)))
    learner = Learner('/tmp/nnl_host_logic', d, net, optimizer='SGD')
    probs, labels = learner.predict('val')
    net.eval()
    with torch.no_grad():
        want = torch.cat([F.log_softmax(net(x), dim=1).exp() for x, _ in batches]).numpy()
        tot = sum(len(y) * F.cross_entropy(net(x), y).item() for x, y in batches)
        correct = sum((net(x).max(dim=1)[1] == y).sum().item() for x, y in batches)
    assert np.array_equal(probs, want) and np.array_equal(labels, want.argmax(axis=1))
    raw, _ = learner.predict('val', correct_probs=False)
    assert raw.shape == (21, 3) and not np.allclose(raw, want)
    loss, acc = learner.evaluate('val')
    assert loss == tot / 21 and acc == correct / 21
    assert learner.evaluate('train') == tot / 21
