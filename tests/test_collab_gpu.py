"""GPU parity of K4 (fused EmbeddingDotBias, through the C ABI) against the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import T, assert_close, load_golden
from oracle import reference_math as RM

pytestmark = pytest.mark.gpu
DEV = 'cuda'
NAMES = ['user_emb', 'item_emb', 'user_bias', 'item_bias']


def _net_from_golden(g, prefix='init.'):
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    net = CollabFilterNet(int(g['n_user']), int(g['n_item']), int(g['D']), [float(g['lo']), float(g['hi'])])
    net.load_state_dict({n + '.weight': T(g[prefix + n + '.weight']) for n in NAMES})
    return net.to(DEV)


def test_golden_forward_backward():
    g = load_golden('g1_collab')
    net = _net_from_golden(g)
    x, y = T(g['x0'], DEV), T(g['y0'], DEV)
    pred = net(x)
    assert_close(pred, g['pred0'], rtol=1e-5, atol=1e-6, msg='pred')
    loss = torch.nn.MSELoss()(pred, y)
    assert_close(loss, g['loss0'], rtol=1e-5, msg='loss')
    loss.backward()
    for n in NAMES:
        assert_close(getattr(net, n).weight.grad, g['grad0.%s.weight' % n], rtol=1e-4, atol=1e-7, msg=n)
    net.output_range = None
    assert_close(net(x), g['pred0_norange'], rtol=1e-5, atol=1e-6, msg='norange')


def test_golden_three_learner_steps():
    """Product Learner.train1minibatch x3 (Adam, wd 1e-4) vs the reference's own 3 steps."""
    from neuralnetworklibrary_amd.General.Learner import Learner

    class Data:
        pass
    g = load_golden('g1_collab')
    net = _net_from_golden(g)
    batches = [(T(g['x%d' % i], DEV), T(g['y%d' % i], DEV)) for i in range(3)]
    d = Data(); d.train_dl = batches; d.val_dl = batches; d.bs = 64; d.target_type = 'cont'
    learner = Learner('/tmp/nnl_test_g1', d, net, optimizer='Adam')
    learner.init_optimizer(wd=1e-4)
    losses = [learner.train1minibatch(x, y, 1e-2) for x, y in batches]
    assert_close(np.array(losses), g['step_losses'], rtol=1e-5, msg='losses')
    for n in NAMES:
        assert_close(getattr(net, n).weight, g['after3.%s.weight' % n], rtol=1e-4, atol=1e-6, msg=n)


@pytest.mark.parametrize('n,n_user,n_item,D,rng', [(64, 943, 1682, 30, True), (1, 5, 7, 1, True), (8192, 1000, 2000, 30, True),
                                                   (333, 17, 19, 50, False), (4097, 64, 64, 128, True),
                                                   (200, 17, 19, 50, True), (256, 3, 300, 7, False)])   # (n <= 256: the one-launch backward, rows with many samples)
def test_vs_oracle_seeded(n, n_user, n_item, D, rng):
    from neuralnetworklibrary_amd import ops
    gen = torch.Generator().manual_seed(n + D)
    x = torch.stack([torch.randint(0, n_user, (n,), generator=gen), torch.randint(0, n_item, (n,), generator=gen)], 1)
    ps = [torch.randn(n_user, D, generator=gen) * 0.3, torch.randn(n_item, D, generator=gen) * 0.3,
          torch.randn(n_user, 1, generator=gen), torch.randn(n_item, 1, generator=gen)]
    dy = torch.randn(n, generator=gen)
    orng = [0.8, 5.2] if rng else None
    cpu = [p.clone().requires_grad_(True) for p in ps]
    ref = RM.embdotbias(x, *cpu, orng)
    ref.backward(dy)
    gpu = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    out = ops.embdotbias(x.to(DEV), *gpu, orng)
    out.backward(dy.to(DEV))
    assert_close(out, ref, rtol=1e-5, atol=1e-5, msg='y')
    # gathers are exact: a sample whose rows are one-hot must reproduce the table entry bit for bit
    for a, b in zip(gpu, cpu):
        assert_close(a.grad, b.grad, rtol=1e-3, atol=1e-5, msg='grad')
    ops.raise_if_index_error()


def test_gather_bit_exact():
    """With M = all-ones rows of width 1 the output is exactly U[u,0] + bu[u] + bi[i]: the gather itself must be
    bit-exact (north_star: 'bit-exact for embedding index gathers')."""
    from neuralnetworklibrary_amd import ops
    gen = torch.Generator().manual_seed(3)
    n_user, n_item, n = 977, 1201, 5000
    U = torch.randn(n_user, 1, generator=gen)
    M = torch.ones(n_item, 1)
    z = torch.zeros(n_user, 1), torch.zeros(n_item, 1)
    x = torch.stack([torch.randint(0, n_user, (n,), generator=gen), torch.randint(0, n_item, (n,), generator=gen)], 1)
    out = ops.embdotbias(x.to(DEV), U.to(DEV), M.to(DEV), z[0].to(DEV), z[1].to(DEV), None).cpu()
    assert torch.equal(out, U[x[:, 0], 0])


def test_empty_and_out_of_range():
    from neuralnetworklibrary_amd import ops
    U, M = torch.randn(5, 4, device=DEV), torch.randn(6, 4, device=DEV)
    bu, bi = torch.randn(5, 1, device=DEV), torch.randn(6, 1, device=DEV)
    out = ops.embdotbias(torch.zeros(0, 2, dtype=torch.long, device=DEV), U, M, bu, bi, [0., 1.])
    assert out.shape == (0,)
    bad = torch.tensor([[0, 1], [5, 0]], device=DEV)      # user index 5 is out of range
    ops.embdotbias(bad, U, M, bu, bi, [0., 1.])
    with pytest.raises(IndexError):
        ops.raise_if_index_error()


@pytest.mark.gpu
@pytest.mark.parametrize('atomic', [0, 1])
def test_backward_skips_a_sample_with_one_bad_index(atomic, monkeypatch):
    """a sample with ONE bad index (the other valid) must be skipped by the backward, not multiplied by 0 —
    the dU pass may not read M[bad item] and the dM pass may not read U[bad user].  The tables sit inside a NaN-filled
    arena so that an out-of-range read that lands next to them would poison the gradient."""
    from neuralnetworklibrary_amd import ops, _lib
    monkeypatch.setenv('NNL_SCATTER_ATOMIC', str(atomic))
    _lib.lib.nnl_reload_env()
    try:
        D = 4
        arena = torch.full((64, D), float('nan'), device=DEV)
        U = arena[8:13].detach()                        # 5 users; row 5 of U is arena[13] = NaN
        M = arena[24:30].detach()                       # 6 items; row 6 of M is arena[30] = NaN
        U.copy_(torch.randn(5, D)); M.copy_(torch.randn(6, D))
        U.requires_grad_(True); M.requires_grad_(True)
        bu = torch.randn(5, 1, device=DEV, requires_grad=True)
        bi = torch.randn(6, 1, device=DEV, requires_grad=True)
        x = torch.tensor([[0, 1], [5, 0], [2, 6], [0, 0], [-1, 3], [4, -1]], device=DEV)     # samples 1, 2, 4, 5 are bad
        dy = torch.randn(6, device=DEV)
        ops.embdotbias(x, U, M, bu, bi, [0., 1.]).backward(dy)
        with pytest.raises(IndexError):
            ops.raise_if_index_error()
        good = torch.tensor([0, 3], device=DEV)
        Uc, Mc, buc, bic = [t.detach().clone().requires_grad_(True) for t in (U, M, bu, bi)]
        ops.embdotbias(x[good], Uc, Mc, buc, bic, [0., 1.]).backward(dy[good])
        for got, want in zip((U, M, bu, bi), (Uc, Mc, buc, bic)):
            assert torch.isfinite(got.grad).all()
            assert torch.equal(got.grad, want.grad)
    finally:
        monkeypatch.delenv('NNL_SCATTER_ATOMIC')
        _lib.lib.nnl_reload_env()


@pytest.mark.gpu
def test_embedding_gradients_are_bitwise_reproducible_and_in_sample_order():
    """VERDICT r1 #9 / SURVEY §7: the three dense embedding-gradient scatters (EmbeddingDotBias, the tabular front end, the
    vocabulary-row-dropout embedding) add the samples of a table row in a FIXED order (rank sort + segment sum) — bit-identical
    run to run on duplicate-heavy indices (atomics were not), equal to the sequential sample-order sum within rounding, and
    bit-identical to it for rows hit only a few times."""
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(5)
    dev = 'cuda'
    # EmbeddingDotBias at the MovieLens-20M notebook batch (8192) with only 7 users / 5 items: ~1000 samples per row
    n, D = 8192, 30
    x = torch.stack([torch.randint(0, 7, (n,), generator=g), torch.randint(0, 5, (n,), generator=g)], 1)
    U, M = torch.randn(7, D, generator=g), torch.randn(5, D, generator=g)
    bu, bi = torch.randn(7, 1, generator=g), torch.randn(5, 1, generator=g)
    dy = torch.randn(n, generator=g)
    runs = []
    for _ in range(3):
        ps = [t.clone().to(dev).requires_grad_(True) for t in (U, M, bu, bi)]
        ops.embdotbias(x.to(dev), *ps, output_range=None).backward(dy.to(dev))
        runs.append([p.grad.clone() for p in ps])
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)
    want = torch.zeros(7, D)
    for i in range(n):                                   # sequential fp32, sample order
        want[x[i, 0]] += dy[i] * M[x[i, 1]]
    assert_close(runs[0][0], want, 1e-5, 1e-4, 'dU vs sequential sample-order sum')
    # tabular: 3 columns of cardinality 2..4, 1024 samples, row masks
    from neuralnetworklibrary_amd.ops import tab_embed_concat
    cards, dims, bs = [2, 3, 4], [3, 5, 2], 1024
    xcat = torch.stack([torch.randint(0, c, (bs,), generator=g) for c in cards], 1).to(dev)
    mask = (torch.rand(3, bs, generator=g) > 0.3).float().to(dev) / 0.7
    dout = torch.randn(bs, sum(dims), generator=g).to(dev)
    outs = []
    for _ in range(3):
        ws = [torch.randn(c, d, generator=torch.Generator().manual_seed(9 + j)).to(dev).requires_grad_(True) for j, (c, d) in enumerate(zip(cards, dims))]
        out, _ = tab_embed_concat(xcat, ws, row_mask=mask)
        out.backward(dout)
        outs.append([w.grad.clone() for w in ws])
    for r in outs[1:]:
        for a, b in zip(outs[0], r):
            assert torch.equal(a, b)
    want = torch.zeros(cards[1], dims[1])
    xc, mc, dc = xcat.cpu(), mask.cpu(), dout.cpu()
    for i in range(bs):
        want[xc[i, 1]] += dc[i, 3:8] * mc[1, i]
    assert_close(outs[0][1], want, 1e-5, 1e-4, 'tabular dW vs sequential sample-order sum')
    # rows hit once: exactly the sample's contribution
    xs = torch.arange(6).view(6, 1).to(dev)
    w1 = torch.zeros(6, 4, device=dev, requires_grad=True)
    d1 = torch.randn(6, 4, generator=g).to(dev)
    o1, _ = tab_embed_concat(xs, [w1])
    o1.backward(d1)
    assert torch.equal(w1.grad, d1)
    # vocabulary embedding: 4480 tokens over 50 rows, row mask, padding row
    V, Dm, ntok = 50, 400, 4480
    tok = torch.randint(0, V, (70, 64), generator=g).to(dev)
    rm = ((torch.rand(V, 1, generator=g) > 0.2).float() / 0.8).to(dev)
    dout = torch.randn(70, 64, Dm, generator=g).to(dev)
    res = []
    for _ in range(3):
        W = torch.randn(V, Dm, generator=torch.Generator().manual_seed(3)).to(dev).requires_grad_(True)
        ops.embedding_rowmask(tok, W, rm, 1).backward(dout)
        res.append(W.grad.clone())
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert float(res[0][1].abs().sum()) == 0.0
    ops.raise_if_index_error()


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 64, 1024, 70001])
def test_hip_mse_loss_vs_torch(n):
    """nn.MSELoss() of the 'cont' target type (reference General/Learner.py:20) on the HIP kernels: value and gradient against torch
    in fp64, an upstream gradient other than 1, and bitwise reproducibility of the fixed-order sum."""
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd.General.Learner import loss_func_dict
    g = torch.Generator().manual_seed(n)
    p, t = torch.randn(n, generator=g) * 3, torch.randn(n, generator=g)
    p64 = p.double().requires_grad_(True)
    ref = ((p64 - t.double()) ** 2).mean()
    (2.5 * ref).backward()
    pd = p.to(DEV).requires_grad_(True)
    out = loss_func_dict['cont'](pd, t.to(DEV))
    (2.5 * out).backward()
    assert_close(out.reshape(1), np.array([float(ref)]), 2e-6, 1e-7, 'value')
    assert_close(pd.grad, p64.grad.float(), 1e-6, 1e-7 * float(p64.grad.abs().max()), 'd pred')
    assert torch.equal(ops.mse_loss(pd.detach(), t.to(DEV)), ops.mse_loss(pd.detach(), t.to(DEV)))
