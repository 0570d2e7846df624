"""Tabular head (K3 + MLP): oracle pinned to reference goldens G2/G3 on CPU; HIP product vs goldens/oracle on GPU."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import T, assert_close, load_golden
from oracle import reference_math as RM
from oracle import reference_nets as RNets
from oracle import synth

DEV = 'cuda'


def make_oracle(g, seed=4):
    cards, dims = [int(c) for c in g['cards']], [int(d) for d in g['emb_dims']]
    net = RNets.StructuredDataNet('cont', list(zip(cards, dims)), int(g['n_cont']), [32, 16, 1], output_range=[5, 12])
    return synth.fill_module_(net, seed=seed)


def make_product(g, seed=4):
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    cards = [int(c) for c in g['cards']]
    labels = [{i: i for i in range(c)} for c in cards]
    net = StructuredDataNet('cont', len(cards), int(g['n_cont']), labels, [32, 16, 1], output_range=[5, 12])
    assert [e.emb.weight.shape[1] for e in net.embeddings] == [int(d) for d in g['emb_dims']]
    return synth.fill_module_(net, seed=seed)


def group_of(name):
    return 1 if name.startswith('head') else 0


def check_forward_backward(net, g, dev, rtol, call):
    net.train()
    xcat, xcont, y = T(g['xcat0'], dev), T(g['xcont0'], dev), T(g['y0'], dev)
    pred = call(net, xcat, xcont)
    assert_close(pred, g['pred0'], rtol, 1e-5, 'pred')
    loss = nn.MSELoss()(pred, y)
    assert_close(loss, g['loss0'], rtol, 1e-6, 'loss')
    loss.backward()
    for n, p in net.named_parameters():
        ref = g['grad0.' + n]
        assert_close(p.grad, ref, 1e-3, 1e-5 * max(np.abs(ref).max(), 1e-3), 'grad ' + n)
        assert_close(p, g['after_fwd0.' + n], 1e-6, 1e-7, 'param after forward (renorm) ' + n)
    for n, b in net.named_buffers():
        assert_close(b, g['buf0.' + n], 1e-5, 1e-6, 'buffer ' + n)


def test_g2_embeddingdrop_oracle():
    g = load_golden('g3_tabular')
    emb = RNets.EmbeddingDrop(20, 5, 0.0, 1.0, 1.5)
    with torch.no_grad():
        emb.emb.weight.copy_(T(g['g2.w_before']))
    y = emb(T(g['g2.x']))
    assert_close(y, g['g2.y'], 1e-6, 1e-7, 'y')
    assert_close(emb.emb.weight, g['g2.w_after'], 1e-6, 1e-7, 'renormed weight')
    changed = np.abs(g['g2.w_after'] - g['g2.w_before']).sum(1) > 0
    assert changed.any() and not changed.all()                     # only touched rows above max_norm moved
    assert set(np.nonzero(changed)[0]) <= set(g['g2.x'].tolist())


def test_g3_oracle_forward_backward_and_steps():
    g = load_golden('g3_tabular')
    check_forward_backward(make_oracle(g), g, 'cpu', 1e-5, lambda n, a, b: n(a, b))
    # restated 3 x train1minibatch: Adam(lr [1e-3,3e-3] per layer group), decoupled wd 1e-3 on reg + bn groups
    net = make_oracle(g).train()
    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    state = RM.OptimState(params)
    lrs = [[1e-3, 3e-3][group_of(n)] for n in names]
    losses = []
    for i in range(3):
        for p in params:
            p.grad = None
        loss = nn.MSELoss()(net(T(g['xcat%d' % i]), T(g['xcont%d' % i])), T(g['y%d' % i]))
        loss.backward()
        losses.append(loss.item())
        RM.optimizer_step(params, [p.grad for p in params], state, lrs, [1e-3] * len(params), 'adam')
    assert_close(np.array(losses), g['step_losses'], 1e-5, 1e-7, 'step losses')
    for n, v in net.state_dict().items():
        assert_close(v, g['after3.' + n], 1e-4, 1e-6, 'after3 ' + n)
    net.eval()
    assert_close(net(T(g['xcat0']), T(g['xcont0'])), g['eval_pred0'], 1e-4, 1e-5, 'eval pred')


@pytest.mark.gpu
def test_g2_embeddingdrop_hip():
    from neuralnetworklibrary_amd.General.Layers import EmbeddingDrop
    g = load_golden('g3_tabular')
    emb = EmbeddingDrop(20, 5, 0.0, 1.0, 1.5)
    with torch.no_grad():
        emb.emb.weight.copy_(T(g['g2.w_before']))
    emb = emb.to(DEV)
    y = emb(T(g['g2.x'], DEV))
    assert_close(y, g['g2.y'], 1e-6, 1e-7, 'y')
    assert_close(emb.emb.weight, g['g2.w_after'], 1e-6, 1e-7, 'renormed weight')
    untouched = np.abs(g['g2.w_after'] - g['g2.w_before']).sum(1) == 0
    assert torch.equal(emb.emb.weight.cpu()[untouched], T(g['g2.w_before'])[untouched])     # bit-exact untouched rows


@pytest.mark.gpu
def test_g3_hip_forward_backward():
    g = load_golden('g3_tabular')
    check_forward_backward(make_product(g).to(DEV), g, DEV, 1e-4, lambda n, a, b: n(a, b))


@pytest.mark.gpu
def test_g3_hip_learner_steps():
    from neuralnetworklibrary_amd.General.Learner import Learner

    class D:
        bs, target_type = 64, 'cont'
    g = load_golden('g3_tabular')
    net = make_product(g).to(DEV)
    batches = [([T(g['xcat%d' % i], DEV), T(g['xcont%d' % i], DEV)], T(g['y%d' % i], DEV)) for i in range(3)]
    d = D(); d.train_dl = batches; d.val_dl = batches
    learner = Learner('/tmp/nnl_test_g3', d, net, optimizer='Adam')
    learner.init_optimizer(wd=1e-3)
    net.train()
    losses = [learner.train1minibatch(xb, yb, [1e-3, 3e-3]) for xb, yb in batches]
    assert_close(np.array(losses), g['step_losses'], 1e-4, 1e-6, 'step losses')
    for n, v in net.state_dict().items():
        assert_close(v, g['after3.' + n], 2e-3, 2e-5, 'after3 ' + n)       # Adam's 1/sqrt(v) amplifies 1e-6 grad noise early on
    net.eval()
    assert_close(net(*batches[0][0]), g['eval_pred0'], 1e-3, 1e-4, 'eval pred')


@pytest.mark.gpu
def test_rossmann_shape_with_dropout_masks_vs_oracle():
    """BASELINE config 3 shapes (bs 1024, 32 columns, 14 cont, fc [1000,500,1]) with injected dropout masks."""
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    from neuralnetworklibrary_amd import ops
    cards = [1116, 5, 4, 13, 53, 13, 4, 8, 32, 23, 27, 24, 28, 9, 5, 5] + [10] * 16
    dims = [RNets.embedding_dim(c) for c in cards]
    labels = [{i: i for i in range(c)} for c in cards]
    bs, n_cont = 1024, 14
    prod = StructuredDataNet('cont', 32, n_cont, labels, [1000, 500, 1], output_range=[5, 12], dropout_levels=(0.04, 0.04, [0, 0., 0.]))
    orac = RNets.StructuredDataNet('cont', list(zip(cards, dims)), n_cont, [1000, 500, 1], output_range=[5, 12])
    synth.fill_module_(prod, 9); synth.fill_module_(orac, 9)
    assert sum(dims) + n_cont == prod.head[0].lins[0].lin.in_features
    rs = np.random.RandomState(1)
    xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=bs) for c in cards], 1).astype(np.int64))
    xcont = torch.from_numpy(rs.standard_normal((bs, n_cont)).astype(np.float32))
    y = torch.from_numpy((5 + 7 * rs.rand(bs)).astype(np.float32))
    row_masks = torch.from_numpy((rs.rand(32, bs) > 0.04).astype(np.float32) / 0.96)
    cont_mask = torch.from_numpy((rs.rand(bs, n_cont) > 0.04).astype(np.float32) / 0.96)
    # oracle: cont dropout mask applied by hand (its nn.Dropout has p=0)
    orac.train()
    cat = torch.cat([e(xcat[:, i], row_masks[i]) for i, e in enumerate(orac.embeddings)], 1)
    ref = orac.head(torch.cat([cat, orac.cont_bn(xcont) * cont_mask], 1))
    lref = nn.MSELoss()(ref, y); lref.backward()
    prod = prod.to(DEV).train()
    prod.inject_masks(row_masks.to(DEV), cont_mask.to(DEV))
    out = prod(xcat.to(DEV), xcont.to(DEV))
    lout = nn.MSELoss()(out, y.to(DEV)); lout.backward()
    ops.raise_if_index_error()
    assert_close(out, ref, 1e-4, 1e-4, 'pred')
    assert_close(lout, lref, 1e-4, 1e-6, 'loss')
    for (n, p), (_, q) in zip(prod.named_parameters(), orac.named_parameters()):
        assert_close(p.grad, q.grad, 1e-3, 1e-4 * max(q.grad.abs().max().item(), 1e-6), 'grad ' + n)
        assert_close(p, q, 1e-6, 1e-7, 'renormed ' + n)


@pytest.mark.gpu
def test_scaled_sigmoid_output_activation():
    "FullyConnectedNet's 'sigmoidal' activation (reference General/Layers.py:150-152) as one kernel each way vs torch in fp64"
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1024, 1, generator=g) * 3
    x64 = x.double().requires_grad_(True)
    ref = 5.0 + (12.0 - 5.0) * x64.sigmoid()
    dy = torch.randn(1024, 1, generator=g)
    ref.backward(dy.double())
    xd = x.to('cuda').requires_grad_(True)
    out = ops.scaled_sigmoid(xd, 5.0, 12.0)
    out.backward(dy.to('cuda'))
    assert_close(out, ref.float(), 1e-6, 1e-6, 'y')
    assert_close(xd.grad, x64.grad.float(), 2e-5, 1e-6, 'dx')


# ---- G16: 20 reference train1minibatch steps at BASELINE config 3's REAL shape (VERDICT r3 next #4b) --------------------------------
def _g16_batches(g):
    bs, n_cont = int(g['bs']), int(g['n_cont'])
    out = []
    for i in range(int(g['steps'])):
        xcat, xcont, y = synth.rossmann_batch(bs, n_cont, i)
        out.append(([torch.from_numpy(xcat), torch.from_numpy(xcont)], torch.from_numpy(y)))
    return out


def test_g16_rossmann_curve_oracle():
    """The oracle (restated StructuredDataNet + restated Optimizer.step) against the REFERENCE's own 20-step fp32 curve at bs 1024 x 32
    columns x fc [1000, 500, 1] (golden G16, oracle/gen_golden_curves.py): every step to 1e-5 — pins the oracle at full size."""
    g = load_golden('g16_rossmann_curve')
    assert [int(c) for c in g['cards']] == synth.ROSSMANN_CARDS
    assert (np.abs(g['losses.f32'] - g['losses.f64']) / np.abs(g['losses.f64'])).max() < 3e-4
    dims = [int(d) for d in g['emb_dims']]
    net = RNets.StructuredDataNet('cont', list(zip(synth.ROSSMANN_CARDS, dims)), int(g['n_cont']), [1000, 500, 1], output_range=[5, 12])
    synth.fill_module_(net, seed=int(g['init_seed']))
    assert [n for n, _ in net.named_parameters()] == [str(s) for s in g['param_names']]
    net.train()
    params = [p for _, p in net.named_parameters()]
    state = RM.OptimState(params)
    lr, wd = [float(v) for v in g['lr']], float(g['wd'])
    lrs = [lr[group_of(n)] for n, _ in net.named_parameters()]
    losses = []
    for (xcat, xcont), y in _g16_batches(g):
        for p in params:
            p.grad = None
        loss = nn.MSELoss()(net(xcat, xcont), y)
        loss.backward()
        losses.append(loss.item())
        RM.optimizer_step(params, [p.grad for p in params], state, lrs, [wd] * len(params), 'adam')
    rel = np.abs(np.array(losses) - g['losses.f32']) / np.abs(g['losses.f32'])
    assert rel.max() < 1e-5, rel
    abs_sums = np.array([p.double().abs().sum().item() for p in params])
    assert_close(abs_sums, g['after.abs_sums.f32'], 1e-5, 1e-7, 'parameter |.|-sums after 20 steps')


@pytest.mark.gpu
def test_g16_rossmann_curve_hip_learner_every_step_within_1e3():
    """The product Learner on the GPU (hipGraph replay on by default for this head) against the same reference curve: |hip - ref32| <=
    1e-3 |ref32| on every one of the 20 steps, parameter |.|-sums afterwards within 3x the reference's fp32-vs-fp64 gap + 1e-3."""
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g16_rossmann_curve')
    labels = [{i: i for i in range(c)} for c in synth.ROSSMANN_CARDS]
    net = StructuredDataNet('cont', 32, int(g['n_cont']), labels, [1000, 500, 1], output_range=[5, 12])
    assert [e.emb.weight.shape[1] for e in net.embeddings] == [int(d) for d in g['emb_dims']]
    synth.fill_module_(net, seed=int(g['init_seed']))
    assert [n for n, _ in net.named_parameters()] == [str(s) for s in g['param_names']]
    batches = _g16_batches(g)

    class D:
        pass
    d = D(); d.train_dl = d.val_dl = batches; d.bs = int(g['bs']); d.target_type = 'cont'
    set_default_device(DEV)
    Learner.verbose = False
    learner = Learner('/tmp/nnl_test_g16', d, net, optimizer='Adam')
    learner.init_optimizer(wd=float(g['wd']))
    net.train()
    lr = [float(v) for v in g['lr']]
    losses = np.array([learner.train1minibatch([b[0][0].to(DEV), b[0][1].to(DEV)], b[1].to(DEV), lr) for b in batches])
    rel = np.abs(losses - g['losses.f32']) / np.abs(g['losses.f32'])
    print('rel |hip - ref32|', np.array2string(rel, precision=1))
    assert (rel <= 1e-3).all(), 'step losses off the reference fp32 curve: worst %.2e at step %d' % (rel.max(), rel.argmax())
    abs_sums = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()])
    a32, a64 = g['after.abs_sums.f32'], g['after.abs_sums.f64']
    tol = 3 * np.abs(a32 - a64) + 1e-3 * np.abs(a64) + 1e-6
    bad = np.nonzero(np.abs(abs_sums - a64) > tol)[0]
    assert len(bad) == 0, [(str(g['param_names'][i]), abs_sums[i], a32[i], a64[i]) for i in bad[:5]]
    net.eval()
    with torch.no_grad():
        pred = net(batches[0][0][0].to(DEV), batches[0][0][1].to(DEV))
    assert_close(pred[:64], g['eval_pred0.f32'], 1e-3, 1e-3, 'eval-mode prediction after the 20 steps')


@pytest.mark.gpu
@pytest.mark.parametrize('bs,cards,n_cont,pad', [(1024, [1116, 5, 4, 13, 53, 13, 4, 8, 32, 23] + [10] * 6, 14, 1), (300, [7, 300, 2], 0, 0),
                                                 (257, [40, 3], 5, 3)])
def test_tabular_backward_without_sort_vs_the_sorted_path(bs, cards, n_cont, pad, monkeypatch):
    """nnl_tab_scan_bwd (one launch: every (table row, component) scans its column and adds in plain sample order — the order
    of torch's CPU embedding backward) against the rank-sort + segment-sum path it replaces by default (several sample slots per
    row + a fixed tree for rows with many samples: another fixed order): equal to rounding, each bitwise repeatable; also with a
    row-strided upstream gradient (the slice of a channel-padded buffer, read in place) and ragged sizes."""
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd.Applications.StructuredData import embedding_dim
    g = torch.Generator().manual_seed(bs)
    dims = [embedding_dim(c) for c in cards]
    xcat = torch.stack([torch.randint(0, c, (bs,), generator=g) for c in cards], 1).to(DEV)
    cont = torch.randn(bs, n_cont, generator=g).to(DEV) if n_cont else None
    row_masks = (torch.rand(len(cards), bs, generator=g) > 0.1).float().div(0.9).to(DEV)
    cont_mask = (torch.rand(bs, n_cont, generator=g) > 0.2).float().div(0.8).to(DEV) if n_cont else None
    ld = sum(dims) + n_cont
    dfull = torch.randn(bs, ld + pad, generator=g).to(DEV)
    outs = []
    for scan in ('0', '1', '1'):
        monkeypatch.setenv('NNL_TAB_SCAN', scan)
        ws = [(torch.randn(c, d, generator=torch.Generator().manual_seed(7 + i)) * 0.1).to(DEV).requires_grad_(True) for i, (c, d) in enumerate(zip(cards, dims))]
        cg = cont.clone().requires_grad_(True) if n_cont else None
        out, _ = ops.tab_embed_concat(xcat, ws, row_masks, cg, cont_mask, None)
        out.backward(dfull[:, :ld])                       # a view with row stride ld + pad
        outs.append([w.grad.clone() for w in ws] + ([cg.grad.clone()] if n_cont else []))
    for a, b, c in zip(*outs):
        assert torch.equal(b, c), 'the scan path is bitwise repeatable'
        assert_close(b, a, 1e-5, 1e-6 * a.abs().max().item(), 'scan vs sorted')
    ops.raise_if_index_error()


@pytest.mark.gpu
@pytest.mark.parametrize('M,K,N,pad,bias', [(1024, 500, 1, 0, True), (7, 37, 3, 0, True), (130, 203, 4, 5, False), (65, 64, 2, 0, True)])
def test_linear_with_few_output_features_vs_torch_fp64(M, K, N, pad, bias):
    """ops.linear's path for 1 - 4 output features (FullyConnectedNet.final_lin of the regression heads; csrc/linear_small.hip):
    forward and all three gradients against torch in fp64, bitwise repeatable, also on a row-strided input."""
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(M + K)
    xfull = torch.randn(M, K + pad, generator=g)
    w, b = torch.randn(N, K, generator=g) / K ** 0.5, (torch.randn(N, generator=g) if bias else None)
    dy = torch.randn(M, N, generator=g)
    xd, wd = xfull[:, :K].double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if bias else None
    ref = torch.nn.functional.linear(xd, wd, bd)
    ref.backward(dy.double())
    runs = []
    for _ in range(2):
        xg = xfull.to(DEV).requires_grad_(True)
        wg = w.to(DEV).requires_grad_(True)
        bg = b.to(DEV).requires_grad_(True) if bias else None
        y = ops.linear(xg[:, :K], wg, bg)
        y.backward(dy.to(DEV))
        runs.append([y.detach().clone(), xg.grad[:, :K].clone(), wg.grad.clone()] + ([bg.grad.clone()] if bias else []))
    for a, c in zip(*runs):
        assert torch.equal(a, c)
    tol = lambda t: 1e-5 * t.abs().max().item()
    assert_close(runs[0][0], ref.detach(), 1e-5, tol(ref.detach()), 'y')
    assert_close(runs[0][1], xd.grad, 1e-5, tol(xd.grad), 'dx')
    assert_close(runs[0][2], wd.grad, 1e-5, tol(wd.grad), 'dw')
    if bias:
        assert_close(runs[0][3], bd.grad, 1e-5, tol(bd.grad), 'db')
