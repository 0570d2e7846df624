"""TEST INFRASTRUCTURE (lives next to __graft_entry__.py, outside the product package).  Small GPU-vs-oracle checks used by
__graft_entry__.smoke(): one tiny forward+backward of every hot-path head on the HIP kernels (through the C ABI), compared with
the CPU oracle (imported here as the CHECKER only)."""
import numpy as np
import torch
import torch.nn as nn


def _close(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    if not bool((err <= atol + rtol * b.abs()).all()):
        raise AssertionError(f'smoke: {what} mismatch, max abs err {err.max().item():.3e}')


def check_collab(device):
    from oracle import reference_math as RM
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(0)
    n, nu, ni, D = 64, 943, 1682, 30
    x = torch.stack([torch.randint(0, nu, (n,), generator=g), torch.randint(0, ni, (n,), generator=g)], 1)
    ps = [torch.randn(nu, D, generator=g) * .3, torch.randn(ni, D, generator=g) * .3,
          torch.randn(nu, 1, generator=g), torch.randn(ni, 1, generator=g)]
    dy = torch.randn(n, generator=g)
    cpu = [p.clone().requires_grad_(True) for p in ps]
    ref = RM.embdotbias(x, *cpu, [0.8, 5.2]); ref.backward(dy)
    gpu = [p.clone().to(device).requires_grad_(True) for p in ps]
    out = ops.embdotbias(x.to(device), *gpu, [0.8, 5.2]); out.backward(dy.to(device))
    _close(out, ref, 1e-5, 1e-5, 'embdotbias fwd')
    for a, b in zip(gpu, cpu):
        _close(a.grad, b.grad, 1e-3, 1e-5, 'embdotbias grad')


def check_resnet_block(device):
    """conv3x3 -> BN -> ReLU -> conv3x3 -> BN -> +downsample(x) -> ReLU, stride 2 (conv fwd/dgrad/wgrad + fused BN)."""
    from oracle import reference_nets as RN
    from oracle import synth
    from neuralnetworklibrary_amd.Applications.VisionModels import retinanet as PN
    ds_p = PN._Downsample(PN.HipConv2d(16, 32, kernel_size=1, stride=2, bias=False), nn.BatchNorm2d(32))
    ds_o = nn.Sequential(nn.Conv2d(16, 32, 1, stride=2, bias=False), nn.BatchNorm2d(32))
    prod, orac = PN.BasicBlock(16, 32, 2, ds_p), RN.BasicBlock(16, 32, 2, ds_o)
    synth.fill_module_(prod, 1); synth.fill_module_(orac, 1)
    x = synth.synth_input((4, 16, 20, 20), 5)
    xo = x.clone().requires_grad_(True)
    yo = orac.train()(xo); dy = synth.synth_input(tuple(yo.shape), 6); yo.backward(dy)
    xp = x.to(device).requires_grad_(True)
    yp = prod.to(device).train()(xp); yp.backward(dy.to(device))
    _close(yp, yo, 1e-4, 1e-5, 'block y'); _close(xp.grad, xo.grad, 1e-3, 1e-5, 'block dx')
    for (n, p), (_, q) in zip(prod.named_parameters(), orac.named_parameters()):
        _close(p.grad, q.grad, 1e-3, 1e-4 * q.grad.abs().max().item(), 'block grad ' + n)


def check_tabular(device):
    from oracle import reference_nets as RN
    from oracle import synth
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    cards, n_cont, bs = [20, 5, 4, 13], 3, 32
    dims = [RN.embedding_dim(c) for c in cards]
    prod = StructuredDataNet('cont', 4, n_cont, [{i: i for i in range(c)} for c in cards], [32, 16, 1], output_range=[5, 12])
    orac = RN.StructuredDataNet('cont', list(zip(cards, dims)), n_cont, [32, 16, 1], output_range=[5, 12])
    synth.fill_module_(prod, 2); synth.fill_module_(orac, 2)
    rs = np.random.RandomState(0)
    xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=bs) for c in cards], 1).astype(np.int64))
    xcont = torch.from_numpy(rs.standard_normal((bs, n_cont)).astype(np.float32))
    y = torch.from_numpy((5 + 7 * rs.rand(bs)).astype(np.float32))
    lo = nn.MSELoss()(orac.train()(xcat, xcont), y); lo.backward()
    lp = nn.MSELoss()(prod.to(device).train()(xcat.to(device), xcont.to(device)), y.to(device)); lp.backward()
    _close(lp, lo, 1e-4, 1e-6, 'tabular loss')
    for (n, p), (_, q) in zip(prod.named_parameters(), orac.named_parameters()):
        _close(p.grad, q.grad, 1e-3, 1e-5 * max(q.grad.abs().max().item(), 1e-3), 'tabular grad ' + n)


def check_retina_loss(device):
    from oracle import reference_math as RM
    from neuralnetworklibrary_amd import ops
    rs = np.random.RandomState(2)
    anchors = RM.anchors_for(64, 64)
    A, K, bs = len(anchors), 5, 2
    boxes = -np.ones((bs, 3, 4), np.float32); cats = -np.ones((bs, 3), np.int64)
    boxes[1, :2] = [[4, 6, 40, 44], [20, 10, 60, 34]]; cats[1, :2] = [2, 0]
    reg = torch.from_numpy(rs.standard_normal((bs, A, 4)).astype(np.float32) * .3)
    clas = torch.from_numpy(rs.uniform(0.01, 0.5, (bs, A, K)).astype(np.float32))
    rc, cc = reg.clone().requires_grad_(True), clas.clone().requires_grad_(True)
    tot, r, c = RM.ssd_loss(anchors, rc, cc, torch.from_numpy(boxes), torch.from_numpy(cats)); tot.backward()
    rg, cg = reg.to(device).requires_grad_(True), clas.to(device).requires_grad_(True)
    out = ops.retina_loss(anchors.to(device), rg, cg, torch.from_numpy(boxes).to(device), torch.from_numpy(cats).to(device))
    out[0].backward()
    _close(out, torch.stack([tot, r, c]), 1e-4, 1e-6, 'retina loss')
    _close(rg.grad, rc.grad, 1e-4, 1e-9, 'retina dreg'); _close(cg.grad, cc.grad, 1e-4, 1e-8, 'retina dclas')


def check_lstm_lm(device):
    from oracle import reference_text as RT
    from oracle import synth
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, RegSeqCrossEntropyLoss, _Vocab
    V, bs, seq = 40, 4, 6
    stoi = {('t%d' % i): i for i in range(V)}
    stoi['_pad_'] = 1
    del stoi['t1']
    d = _Vocab(stoi, bs)
    prod = LanguageModelNet(d, enc_drops=[0., 0., 0., 0.], dec_drop=0., emb_dim=16, hidden_size=24, num_layers=3)
    orac = RT.LanguageModelNet(V, 1, bs, E=16, Hh=24, L=3)
    synth.fill_module_(prod, 3); synth.fill_module_(orac, 3)
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.randint(0, V, (bs, seq)).astype(np.int64)); y = torch.from_numpy(rs.randint(0, V, (bs, seq)).astype(np.int64))
    lo, _ = RT.reg_seq_cross_entropy(orac.train()(x), y); lo.backward()
    lp = RegSeqCrossEntropyLoss()(prod.to(device).train()(x.to(device)), y.to(device)); lp.backward()
    _close(lp, lo, 1e-4, 1e-6, 'LM loss')
    for (n, p), (_, q) in zip(prod.named_parameters(), orac.named_parameters()):
        _close(p.grad, q.grad, 1e-3, 1e-5 * max(q.grad.abs().max().item(), 1e-3), 'LM grad ' + n)


CHECKS = [check_collab, check_resnet_block, check_tabular, check_retina_loss, check_lstm_lm]


def run_all(device='cuda:0'):
    from neuralnetworklibrary_amd import ops
    for c in CHECKS:
        c(device)
        print('  smoke check passed:', c.__name__)
    ops.raise_if_index_error()
    torch.cuda.synchronize()
