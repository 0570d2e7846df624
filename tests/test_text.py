"""Language-model path (K5 / K5b): oracle pinned to the reference golden G7 on CPU; HIP product vs golden / oracle on GPU."""
import numpy as np
import pytest
import torch

from conftest import T, assert_close, load_golden
from oracle import reference_math as RM
from oracle import reference_text as RT
from oracle import synth

DEV = 'cuda'


def lm_group_of(name):
    "layer groups of LanguageModelNet: [enc.lstms, head=dec(tied embedding)] (Text.py:642)"
    return 0 if '.lstms.' in name else 1


def b_masks(g, dev='cpu'):
    return {'emb_rows': T(g['b.mask.emb_rows'], dev), 'emb_locked': T(g['b.mask.emb_locked'], dev),
            'weights': [T(g['b.mask.weights%d' % i], dev) for i in range(3)],
            'hidden': [T(g['b.mask.hidden%d' % i], dev) for i in range(3)]}


def check_layer(mod, g, dev, call, rtol):
    synth.fill_module_(mod, seed=7)
    mod = mod.to(dev)
    x = synth.synth_input((6, 4, 8), 41).to(dev).requires_grad_(True)
    h0, c0 = synth.synth_input((1, 4, 12), 42, 0.5).to(dev), synth.synth_input((1, 4, 12), 43, 0.5).to(dev)
    y, (hT, cT) = call(mod, x, (h0, c0), T(g['a.wmask'], dev))
    assert_close(y, g['a.y'], rtol, 1e-6, 'y'); assert_close(hT, g['a.hT'], rtol, 1e-6, 'hT'); assert_close(cT, g['a.cT'], rtol, 1e-6, 'cT')
    (y * synth.synth_input((6, 4, 12), 44).to(dev)).sum().backward()
    assert_close(x.grad, g['a.dx'], rtol * 10, 1e-6, 'dx')
    assert [n for n, _ in mod.named_parameters()] == [str(s) for s in g['a.param_names']]
    for n, p in mod.named_parameters():
        assert_close(p.grad, g['a.grad.' + n], rtol * 10, 1e-5, 'grad ' + n)


def check_encoder(enc, g, dev, call, rtol):
    synth.fill_module_(enc, seed=8)
    enc = enc.to(dev)
    enc.train()
    masks = b_masks(g, dev)
    for b in range(2):
        out = call(enc, T(g['b.x%d' % b], dev), masks)
        assert_close(out, g['b.out%d' % b], rtol, 1e-6, 'out%d' % b)
    (out * synth.synth_input(tuple(out.shape), 62).to(dev)).sum().backward()
    for n, p in enc.named_parameters():
        assert_close(p.grad, g['b.grad.' + n], rtol * 10, 1e-5, 'grad ' + n)
    assert_close(enc.h[0], g['b.h_final0'], rtol, 1e-6, 'carried h'); assert_close(enc.c[2], g['b.c_final2'], rtol, 1e-6, 'carried c')


def test_g7_layer_oracle():
    g = load_golden('g7_text')
    check_layer(RT.WeightDropLSTM1(8, 12), g, 'cpu', lambda m, x, hc, wm: m(x, hc, wm), 1e-5)


def test_g7_encoder_oracle():
    g = load_golden('g7_text')
    check_encoder(RT.LSTM_Encoder(50, 8, 12, 3, 1, 4), g, 'cpu', lambda e, x, m: e(x, m), 1e-5)


def _lm_setup(net, scale_emb=True):
    synth.fill_module_(net, seed=9)
    with torch.no_grad():
        net.enc.word_embed.embed.weight.mul_(0.3)
    return net


def check_lm_forward_backward(net, lossf, g, dev, rtol):
    net.train()
    x, y = T(g['c.x0'], dev), T(g['c.y0'], dev)
    out = net(x)
    loss, ce = lossf(out, y)
    assert_close(loss, g['c.loss'], rtol, 1e-6, 'loss'); assert_close(ce, g['c.ce'], rtol, 1e-6, 'ce')
    assert_close(out[0][:, :, :2], g['c.preds_slice'], rtol * 10, 1e-5, 'preds')
    loss.backward()
    assert [n for n, _ in net.named_parameters()] == [str(s) for s in g['c.param_names']]
    norms = np.array([p.grad.norm().item() for _, p in net.named_parameters()])
    assert_close(norms, g['c.grad_norms'], 1e-3, 1e-8, 'grad norms')
    sd = dict(net.named_parameters())
    assert_close(sd['enc.word_embed.embed.weight'].grad, g['c.grad.emb'], 1e-3, 1e-6, 'tied embedding grad')
    assert_close(sd['enc.lstms.0.lstm.weight_hh_l0_raw'].grad[:64, :64], g['c.grad.whh0_slice'], 1e-3, 1e-7, 'w_hh grad')
    assert_close(sd['enc.lstms.2.lstm.bias_ih_l0'].grad, g['c.grad.bias2'], 1e-3, 1e-7, 'bias grad')


def test_g7_language_model_oracle():
    g = load_golden('g7_text')
    net = _lm_setup(RT.LanguageModelNet(60, 1, 4))
    check_lm_forward_backward(net, lambda o, y: RT.reg_seq_cross_entropy(o, y, 2.0, 1.0), g, 'cpu', 1e-5)
    # two restated Learner steps: Adam betas (.8,.99), lr [1e-3, 2e-3] per layer group, wd 1e-6, clip 0.4
    net = _lm_setup(RT.LanguageModelNet(60, 1, 4)).train()
    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    # the reference Optimizer orders torch param groups [reg_0, reg_1, bn_0, bn_1]; the tied embedding sits in BOTH layer
    # groups' module lists only once as a Parameter of group 1's decoder AND group... it is a parameter of enc.word_embed,
    # which is in neither layer group: it is reached through dec.lin.weight (group 1)
    lrs = [[1e-3, 2e-3][lm_group_of(n)] for n in names]
    state = RM.OptimState(params)
    losses = []
    for i in range(2):
        for p in params:
            p.grad = None
        loss, _ = RT.reg_seq_cross_entropy(net(T(g['c.x%d' % i])), T(g['c.y%d' % i]), 2.0, 1.0)
        loss.backward()
        losses.append(loss.item())
        RM.optimizer_step(params, [p.grad for p in params], state, lrs, [1e-6] * len(params), 'adam', betas=(0.8, 0.99), clip=0.4)
    assert_close(np.array(losses), g['c.step_losses'], 1e-5, 1e-7, 'step losses')
    assert_close(np.array([p.double().abs().sum().item() for p in params]), g['c.after.abs_sums'], 1e-5, 1e-8, 'abs sums')
    assert_close(net.enc.word_embed.embed.weight, g['c.after.emb'], 1e-4, 2e-5, 'embedding after 2 steps')   # Adam: lr*m/(sqrt(v)+eps) amplifies 1e-8 grad noise on near-zero-grad rows


# ---- GPU ---------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_g7_layer_hip():
    from neuralnetworklibrary_amd.Applications.Text import WeightDropLSTM1
    g = load_golden('g7_text')
    check_layer(WeightDropLSTM1(8, 12, 0.5), g, DEV, lambda m, x, hc, wm: m(x, hc, wm), 1e-4)


@pytest.mark.gpu
def test_g7_encoder_hip():
    from neuralnetworklibrary_amd.Applications.Text import LSTM_Encoder
    g = load_golden('g7_text')

    def call(enc, x, masks):
        enc.fixed_masks = masks
        return enc(x)
    check_encoder(LSTM_Encoder(50, 8, 12, 3, 1, [0.3, 0.3, 0.4, 0.3], 4), g, DEV, call, 1e-4)


def _product_lm():
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, _Vocab
    stoi = {('tok%d' % i): i for i in range(60)}
    stoi['_pad_'] = 1
    del stoi['tok1']
    d = _Vocab(stoi, 4)
    d.target_type = 'lang_model'
    return _lm_setup(LanguageModelNet(d, enc_drops=[0., 0., 0., 0.], dec_drop=0.)).to(DEV), d


@pytest.mark.gpu
def test_g7_language_model_hip():
    from neuralnetworklibrary_amd.Applications.Text import RegSeqCrossEntropyLoss
    g = load_golden('g7_text')
    net, _ = _product_lm()
    lf = RegSeqCrossEntropyLoss(2.0, 1.0)
    check_lm_forward_backward(net, lambda o, y: (lf(o, y), lf.cross_entropy), g, DEV, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('ask_for_graphs', [False, True])
def test_g7_language_model_hip_learner_steps(ask_for_graphs):
    """ask_for_graphs: Learner.use_graphs() on a language model must NOT capture the step (the carried hidden state and the
    weight-drop seed live on the host: LSTM_Encoder / WeightDropLSTM1 are marked nnl_stateful_forward) — same golden losses, eagerly."""
    from neuralnetworklibrary_amd.Applications.Text import RegSeqCrossEntropyLoss
    from neuralnetworklibrary_amd.General.Learner import Learner
    g = load_golden('g7_text')
    net, d = _product_lm()
    batches = [(T(g['c.x%d' % i], DEV), T(g['c.y%d' % i], DEV)) for i in range(2)]
    d.train_dl, d.val_dl = batches, batches
    learner = Learner('/tmp/nnl_test_g7', d, net, optimizer='Adam', loss_func=RegSeqCrossEntropyLoss(2.0, 1.0))
    learner.init_optimizer(wd=1e-6, clip=0.4)
    if ask_for_graphs:
        learner.use_graphs(True, warmup=0)
    net.train()
    losses = [learner.train1minibatch(x, y, [1e-3, 2e-3], betas_batch=(0.8, 0.99)) for x, y in batches]
    assert_close(np.array(losses), g['c.step_losses'], 1e-4, 1e-6, 'step losses')
    assert_close(np.array([p.double().abs().sum().item() for p in net.parameters()]), g['c.after.abs_sums'], 1e-4, 1e-7, 'abs sums')
    assert_close(net.enc.word_embed.embed.weight, g['c.after.emb'], 1e-3, 1e-4, 'embedding after 2 steps')
    assert learner._graphs == {}, 'a stateful forward was captured'


@pytest.mark.gpu
@pytest.mark.parametrize('rows,V', [(7, 10), (64, 47343), (33, 1001)])
def test_softmax_ce_vs_torch(rows, V):
    from neuralnetworklibrary_amd import ops
    gen = torch.Generator().manual_seed(rows + V)
    logits = torch.randn(rows, V, generator=gen) * 3
    logits[0, :min(V, 5)] += 30                      # a dominant class: exercises the online-max rescale
    target = torch.randint(0, V, (rows,), generator=gen)
    lc = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lc, target)
    ref.backward()
    lg = logits.to(DEV).requires_grad_(True)
    out = ops.softmax_cross_entropy(lg, target.to(DEV))
    out.backward()
    assert_close(out, ref, 1e-5, 1e-6, 'loss')
    assert_close(lg.grad, lc.grad, 1e-4, 1e-9, 'dlogits')
    ops.raise_if_index_error()


@pytest.mark.gpu
def test_embedding_rowmask_bit_exact_and_grad():
    from neuralnetworklibrary_amd import ops
    gen = torch.Generator().manual_seed(4)
    V, D = 977, 400
    W = torch.randn(V, D, generator=gen)
    x = torch.randint(0, V, (70, 64), generator=gen)
    out = ops.embedding_rowmask(x.to(DEV), W.to(DEV), None, 1)
    assert torch.equal(out.cpu(), W[x])                                   # index gather is bit exact
    mask = (torch.rand(V, 1, generator=gen) > 0.3).float() / 0.7
    Wc = W.clone().requires_grad_(True)
    ref = torch.nn.functional.embedding(x, Wc * mask, 1)
    dy = torch.randn(ref.shape, generator=gen)
    ref.backward(dy)
    Wg = W.to(DEV).requires_grad_(True)
    o = ops.embedding_rowmask(x.to(DEV), Wg, mask.to(DEV), 1)
    o.backward(dy.to(DEV))
    assert_close(o, ref, 1e-6, 1e-7, 'out')
    assert_close(Wg.grad, Wc.grad, 1e-4, 1e-5, 'dW')
    assert float(Wg.grad[1].abs().sum()) == 0.0                            # padding row gets no gradient


@pytest.mark.gpu
@pytest.mark.parametrize('fused_bwd,persist', [('0', '1'), ('1', '0'), ('0', '3'), ('0', '0'), ('0', '5')])
def test_lstm_full_size_recurrence_vs_torch_and_bitwise_repeatable(fused_bwd, persist, monkeypatch):
    """BASELINE-size layer (bs 64, bptt 70, 1150 -> 1150) on every recurrence path: the persistent cooperative kernels
    (lstm_persist.hip: 230 workgroups exchange h_t / dgates_t through per-timestep slots and meet at a grid barrier per step —
    a stale read or a missed arrival would show as a run-to-run difference, a large error or the time-out flag) and the
    per-timestep path, whose fused step kernel hands split-K partial tiles between
    workgroups (ticket + device-scope stores / loads, 72 tiles x 70 steps per pass); a stale hand-over would show up as a
    run-to-run difference or a large error.  Checked bitwise over repeated runs and against torch's LSTM in fp64."""
    from neuralnetworklibrary_amd import ops_text
    from neuralnetworklibrary_amd._lib import lib
    monkeypatch.setenv('NNL_LSTM_FUSED_BWD', fused_bwd)       # '1': the (slower) fused backward step is kept tested too
    monkeypatch.setenv('NNL_LSTM_PERSIST', persist)           # 1: persistent forward; 3: + first persistent BPTT; 5: + 2-D partitioned BPTT (lstm_bptt2.hip); 0: per-timestep
    lib.nnl_reload_env()
    T, B, I, H = 70, 64, 1150, 1150
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(T, B, I, generator=g) * 0.5)
    w_ih, w_hh = torch.randn(4 * H, I, generator=g) / I ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5
    b_ih, b_hh = torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1
    h0, c0 = torch.randn(1, B, H, generator=g) * 0.1, torch.randn(1, B, H, generator=g) * 0.1
    dev = lambda t: t.to('cuda')
    runs = []
    for _ in range(3):
        xg, wg = dev(x).requires_grad_(True), dev(w_hh).requires_grad_(True)
        y, (hT, cT) = ops_text.lstm_layer(xg, dev(h0), dev(c0), dev(w_ih), wg, dev(b_ih), dev(b_hh))
        (y.sum() + 0.5 * cT.sum()).backward()
        runs.append((y.detach().clone(), cT.detach().clone(), xg.grad.clone(), wg.grad.clone()))
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b), 'LSTM forward / backward must be bitwise repeatable'
    ref = torch.nn.LSTM(I, H).double()
    with torch.no_grad():
        ref.weight_ih_l0.copy_(w_ih); ref.weight_hh_l0.copy_(w_hh); ref.bias_ih_l0.copy_(b_ih); ref.bias_hh_l0.copy_(b_hh)
    xd = x.double().requires_grad_(True)
    yd, (hd, cd) = ref(xd, (h0.double(), c0.double()))
    (yd.sum() + 0.5 * cd.sum()).backward()
    assert_close(runs[0][0], yd.detach(), 1e-4, 1e-5, 'y')
    assert_close(runs[0][1], cd.detach(), 1e-4, 1e-5, 'cT')
    assert_close(runs[0][2], xd.grad, 1e-3, 1e-4 * xd.grad.abs().max().item(), 'dx')
    assert_close(runs[0][3], ref.weight_hh_l0.grad, 1e-3, 1e-4 * ref.weight_hh_l0.grad.abs().max().item(), 'dW_hh')
    from neuralnetworklibrary_amd import ops
    ops.raise_if_index_error()                                 # (also carries the persistent kernel's barrier time-out flag)


@pytest.mark.gpu
@pytest.mark.parametrize('persist', ['5'])
@pytest.mark.parametrize('T,B,I,H', [(12, 20, 48, 100), (70, 64, 1150, 400), (9, 64, 32, 32), (5, 33, 64, 256), (6, 7, 40, 37)])
def test_lstm_bptt2_partition_shapes_vs_torch_fp64(T, B, I, H, persist, monkeypatch):
    """The 2-D partitioned persistent BPTT (lstm_bptt2.hip, NNL_LSTM_PERSIST bit 2) on other shapes than the headline's: the
    last AWD-LSTM layer (1150 -> 400), ragged batches (B < 64: masked rows), H not a multiple of the column tile, a k range
    that leaves the padded gate columns to the last slice.  Against torch's LSTM in fp64, bitwise repeatable, and the planner's
    partition must cover the problem."""
    import ctypes
    from neuralnetworklibrary_amd import ops_text, ops
    from neuralnetworklibrary_amd._lib import lib
    out5 = (ctypes.c_int32 * 5)()
    assert lib.nnl_debug_lstm_bptt2_plan(B, H, out5) == 1
    KG, NG, Ks, Ns, NT = list(out5)
    Gp = int(lib.nnl_lstm_padded_gates(H))
    assert KG * NG <= 256 and KG * Ks == Gp and NG * Ns >= H and (NG - 1) * Ns < H and 16 * NT >= Ns
    monkeypatch.setenv('NNL_LSTM_PERSIST', persist)
    lib.nnl_reload_env()
    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, B, I, generator=g) * 0.5
    w_ih, w_hh = torch.randn(4 * H, I, generator=g) / I ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5
    b_ih, b_hh = torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1
    h0, c0 = torch.randn(1, B, H, generator=g) * 0.1, torch.randn(1, B, H, generator=g) * 0.1
    wy, wc, wh = torch.randn(T, B, H, generator=g), torch.randn(1, B, H, generator=g), torch.randn(1, B, H, generator=g)
    dev = lambda t: t.to('cuda')
    runs = []
    for _ in range(2):
        xg, wg, hg, cg = dev(x).requires_grad_(True), dev(w_hh).requires_grad_(True), dev(h0).requires_grad_(True), dev(c0).requires_grad_(True)
        y, (hT, cT) = ops_text.lstm_layer(xg, hg, cg, dev(w_ih), wg, dev(b_ih), dev(b_hh))
        ((y * dev(wy)).sum() + (cT * dev(wc)).sum() + (hT * dev(wh)).sum()).backward()
        runs.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone(), hg.grad.clone(), cg.grad.clone()))
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b), 'bitwise repeatable'
    ref = torch.nn.LSTM(I, H).double()
    with torch.no_grad():
        ref.weight_ih_l0.copy_(w_ih); ref.weight_hh_l0.copy_(w_hh); ref.bias_ih_l0.copy_(b_ih); ref.bias_hh_l0.copy_(b_hh)
    xd, hd0, cd0 = x.double().requires_grad_(True), h0.double().requires_grad_(True), c0.double().requires_grad_(True)
    yd, (hd, cd) = ref(xd, (hd0, cd0))
    ((yd * wy.double()).sum() + (cd * wc.double()).sum() + (hd * wh.double()).sum()).backward()
    tol = lambda t: 1e-4 * t.abs().max().item()
    assert_close(runs[0][0], yd.detach(), 1e-4, 1e-5, 'y')
    assert_close(runs[0][1], xd.grad, 1e-3, tol(xd.grad), 'dx')
    assert_close(runs[0][2], ref.weight_hh_l0.grad, 1e-3, tol(ref.weight_hh_l0.grad), 'dW_hh')
    assert_close(runs[0][3], hd0.grad, 1e-3, tol(hd0.grad), 'dh0')
    assert_close(runs[0][4], cd0.grad, 1e-3, tol(cd0.grad), 'dc0')
    ops.raise_if_index_error()


@pytest.mark.gpu
def test_language_model_full_baseline_size_vs_oracle_fp64():
    """BASELINE configs[3] assembled at its own size — LanguageModelNet 400 / 1150 / 3 layers, V = 47 343, bs 64, bptt 70, every
    dropout mask injected (embedding rows, locked embedding / hidden / decoder masks, the three weight-drop masks) — one
    forward + RegSeqCrossEntropyLoss(2, 1) + backward of the HIP path against the CPU oracle run in fp32 AND fp64 on the same
    weights, tokens and masks.  Per parameter gradient: ||hip - f64|| <= 3 ||cpu32 - f64|| + 1e-3 ||f64|| (as far from the exact
    answer as torch's own fp32 run, x3, plus north_star's 1e-3); loss and cross-entropy likewise."""
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, RegSeqCrossEntropyLoss, _Vocab
    V, bs, bptt, E, H = 47343, 64, 70, 400, 1150
    stoi = {i: i for i in range(V)}
    stoi['_pad_'] = 1
    del stoi[1]
    torch.manual_seed(11)
    net = LanguageModelNet(_Vocab(stoi, bs))                     # the product's (= the reference's) initialisation
    o32 = RT.LanguageModelNet(V, 1, bs)
    sd = {k: v.clone() for k, v in net.state_dict().items() if k in o32.state_dict()}     # (the product's `head` aliases `dec`)
    o64 = RT.LanguageModelNet(V, 1, bs).double()
    o32.load_state_dict(sd)
    o64.load_state_dict({k: v.double() for k, v in sd.items()})
    o64.enc.h, o64.enc.c = [t.double() for t in o64.enc.h], [t.double() for t in o64.enc.c]      # (carried state: plain tensors)
    g = torch.Generator().manual_seed(12)
    x, y = torch.randint(4, V, (bs, bptt), generator=g), torch.randint(4, V, (bs, bptt), generator=g)
    keep = lambda shape, p: torch.bernoulli(torch.full(shape, 1 - p), generator=g) / (1 - p)
    sizes = [E, H, H, E]
    masks = {'emb_rows': keep((V, 1), 0.035), 'emb_locked': keep((1, bs, E), 0.175),
             'weights': [keep((4 * sizes[i + 1], sizes[i + 1]), 0.14) for i in range(3)],
             'hidden': [keep((1, bs, sizes[i + 1]), 0.105) for i in range(3)]}
    dec_mask = keep((1, bs, E), 0.07)
    to = lambda m, f: {k: ([f(t) for t in v] if isinstance(v, list) else f(v)) for k, v in m.items()}
    net = net.to(DEV).train()
    net.enc.fixed_masks, net.dec.fixed_mask = to(masks, lambda t: t.to(DEV)), dec_mask.to(DEV)
    lf = RegSeqCrossEntropyLoss(2.0, 1.0)
    loss_p = lf(net(x.to(DEV)), y.to(DEV))
    loss_p.backward()
    l32, ce32 = RT.reg_seq_cross_entropy(o32.train()(x, masks, dec_mask), y, 2.0, 1.0)
    l32.backward()
    l64, ce64 = RT.reg_seq_cross_entropy(o64.train()(x, to(masks, lambda t: t.double()), dec_mask.double()), y, 2.0, 1.0)
    l64.backward()
    for name, hip, c32, c64 in (('loss', loss_p.item(), l32.item(), l64.item()), ('ce', float(lf.cross_entropy), ce32.item(), ce64.item())):
        assert abs(hip - c64) <= 3 * abs(c32 - c64) + 1e-3 * abs(c64), '%s: hip %.8f cpu32 %.8f f64 %.8f' % (name, hip, c32, c64)
    worst = 0.0
    for (n, pp), (_, p32), (_, p64) in zip(net.named_parameters(), o32.named_parameters(), o64.named_parameters()):
        g64, g32, gp = p64.grad, p32.grad.double(), pp.grad.detach().cpu().double()
        e_hip, e_cpu, ref = (gp - g64).norm().item(), (g32 - g64).norm().item(), g64.norm().item()
        worst = max(worst, e_hip / max(ref, 1e-300))
        assert e_hip <= 3 * e_cpu + 1e-3 * ref, '%s: |hip-f64| %.3e vs |cpu32-f64| %.3e (|f64| %.3e)' % (n, e_hip, e_cpu, ref)
    for i in range(3):                                            # the carried state the next minibatch starts from (Text.py:547-550)
        assert_close(net.enc.h[i], o64.enc.h[i].float(), 1e-3, 1e-5, 'carried h%d' % i)
        assert_close(net.enc.c[i], o64.enc.c[i].float(), 1e-3, 1e-5, 'carried c%d' % i)
    print('worst relative gradient error vs fp64: %.2e' % worst)


def _g14_batches(g, dev):
    V, bs, bptt, steps = int(g['V']), int(g['bs']), int(g['bptt']), int(g['steps'])
    stream = synth.lm_stream(V, bs, steps * bptt + 1, 14)
    return [(torch.from_numpy(stream[:, i * bptt:(i + 1) * bptt].copy()).to(dev), torch.from_numpy(stream[:, i * bptt + 1:(i + 1) * bptt + 1].copy()).to(dev))
            for i in range(steps)]


@pytest.mark.gpu
def test_g14_language_model_20_step_loss_curve_every_step_within_1e3():
    """20 consecutive `train1minibatch` steps of the product Learner at BASELINE configs[3]'s own size
    (400 / 1150 / 3, V = 47 343, bs 64, bptt 70, Adam betas (0.8, 0.99), lr [5e-4, 1e-3], wd 1e-6, RegSeqCrossEntropyLoss(2, 1),
    dropout 0, hidden state carried over 20 consecutive windows of one token stream) against the REFERENCE's own Learner on the
    same closed-form init and tokens (golden G14, oracle/gen_golden_curves.py; the reference's fp32 and fp64 runs stay within
    3e-4 of each other on every step).  |hip - ref32| <= 1e-3 |ref32| on EVERY step."""
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, RegSeqCrossEntropyLoss, _Vocab
    from neuralnetworklibrary_amd.General.Learner import Learner, opt_dict
    from neuralnetworklibrary_amd.General.Optimizer import Optimizer
    from functools import partial
    g = load_golden('g14_lm_curve')
    V, bs = int(g['V']), int(g['bs'])
    r32, r64 = g['losses.f32'], g['losses.f64']
    assert (np.abs(r32 - r64) / np.abs(r64)).max() < 3e-4
    stoi = {i: i for i in range(V)}
    stoi['_pad_'] = 1
    del stoi[1]
    d = _Vocab(stoi, bs)
    d.target_type = 'lang_model'
    net = LanguageModelNet(d, enc_drops=[0., 0., 0., 0.], dec_drop=0.)
    net.clear_non_raw()
    synth.fill_lm_reference_init_(net, seed=int(g['init_seed']))
    assert sorted(n for n, _ in net.named_parameters()) == sorted(str(s) for s in g['param_names'])
    net = net.to(DEV)
    batches = _g14_batches(g, DEV)
    d.train_dl, d.val_dl = batches[:1], batches[:1]
    opt = Optimizer(partial(torch.optim.Adam, betas=(0.8, 0.99)), net)
    learner = Learner('/tmp/nnl_test_g14', d, net, opt, RegSeqCrossEntropyLoss(2.0, 1.0))
    learner.init_optimizer(wd=float(g['wd']))
    net.train()
    lr = [float(v) for v in g['lr']]
    losses = np.array([learner.train1minibatch(x, y, lr, betas_batch=(0.8, 0.99)) for x, y in batches])
    rel32 = np.abs(losses - r32) / np.abs(r32)
    print('losses         ', np.array2string(losses, precision=4))
    print('rel |hip-ref32|', np.array2string(rel32, precision=1))
    print('rel |hip-f64|  ', np.array2string(np.abs(losses - r64) / np.abs(r64), precision=1))
    print('rel |ref32-f64|', np.array2string(np.abs(r32 - r64) / np.abs(r64), precision=1))
    assert (rel32 <= 1e-3).all(), 'step losses off the reference fp32 curve: worst %.2e at step %d' % (rel32.max(), rel32.argmax())
    by_name = {str(n): (a, b) for n, a, b in zip(g['param_names'], g['after.abs_sums.f32'], g['after.abs_sums.f64'])}
    for n, p in net.named_parameters():
        a32, a64 = by_name[n]
        got = p.double().abs().sum().item()
        assert abs(got - a64) <= 3 * abs(a32 - a64) + 1e-3 * abs(a64), '|.|-sum of %s after the 20 steps: hip %.8g ref32 %.8g f64 %.8g' % (n, got, a32, a64)


def test_g14_oracle_first_steps_at_baseline_size():
    """the CPU oracle (restated LM + restated Optimizer.step, Adam) reproduces the first 2 steps of the reference's G14 curve at
    BASELINE configs[3]'s own size, state carried from the first window to the second"""
    g = load_golden('g14_lm_curve')
    V, bs = int(g['V']), int(g['bs'])
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))          # (restored below: the thread count changes the rounding of later CPU oracle runs)
    try:
        net = RT.LanguageModelNet(V, 1, bs)
        synth.fill_lm_reference_init_(net, seed=int(g['init_seed']))
        net.train()
        names = [n for n, _ in net.named_parameters()]
        params = [p for _, p in net.named_parameters()]
        lrs = [float(g['lr'][lm_group_of(n)]) for n in names]
        state = RM.OptimState(params)
        E, H = 400, 1150
        sizes = [E, H, H, E]
        ones = {'emb_rows': torch.ones(V, 1), 'emb_locked': torch.ones(1, bs, E), 'weights': [torch.ones(4 * sizes[i + 1], sizes[i + 1]) for i in range(3)],
                'hidden': [torch.ones(1, bs, sizes[i + 1]) for i in range(3)]}
        for i, (x, y) in enumerate(_g14_batches(g, 'cpu')[:2]):
            for p in params:
                p.grad = None
            loss = RT.reg_seq_cross_entropy(net(x, ones, torch.ones(1, bs, E)), y, 2.0, 1.0)[0]
            loss.backward()
            RM.optimizer_step(params, [p.grad for p in params], state, lrs, [float(g['wd'])] * len(params), 'adam', betas=(0.8, 0.99))
            assert abs(loss.item() - g['losses.f32'][i]) <= 5e-5 * abs(g['losses.f32'][i]), (i, loss.item(), g['losses.f32'][i])
    finally:
        torch.set_num_threads(threads)


@pytest.mark.gpu
@pytest.mark.parametrize('T,B,E,alpha,beta', [(70, 64, 400, 2.0, 1.0), (5, 3, 7, 0.5, 3.0), (1, 4, 8, 2.0, 1.0), (9, 2, 16, 0.0, 1.0)])
def test_seq_activation_reg_vs_torch(T, B, E, alpha, beta):
    """the AR / TAR terms of RegSeqCrossEntropyLoss (reference Text.py:775-776) on one HIP reduction kernel: value and gradient
    against torch's `alpha * h.pow(2).mean() + beta * (h[1:] - h[:-1]).pow(2).mean()` in fp64"""
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(T * 100 + B)
    h = torch.randn(T, B, E, generator=g)
    h64 = h.double().requires_grad_(True)
    ref = alpha * h64.pow(2).mean() + (beta * (h64[1:] - h64[:-1]).pow(2).mean() if T > 1 else 0.0)
    (3.0 * ref).backward()
    hd = h.to(DEV).requires_grad_(True)
    out = ops.seq_activation_reg(hd, alpha, beta)
    (3.0 * out).backward()
    assert_close(out.reshape(1), np.array([float(ref)]), 1e-5, 1e-7, 'value')
    assert_close(hd.grad, h64.grad.float(), 1e-5, 1e-6 * h64.grad.abs().max().item(), 'dh')
    a = ops.seq_activation_reg(hd.detach(), alpha, beta)
    assert torch.equal(a, ops.seq_activation_reg(hd.detach(), alpha, beta))            # fixed-order sums: reproducible


@pytest.mark.gpu
def test_weight_drop_kernel_masks_pads_and_is_its_own_backward():
    """nnl_weight_drop (WeightDropLSTM1's `Dropout_p(W_raw)`, Text.py:511, fused with the padding of W_hh): (1) an explicit mask
    reproduces raw * mask with zero pad columns, bit for bit; (2) a generated mask is Bernoulli(1 - p) / (1 - p), the SAME for the
    same seed (so the backward call re-derives the forward's mask) and different for another; (3) through the LSTM layer the
    gradient of the raw matrix is dW * mask."""
    from neuralnetworklibrary_amd import ops_text
    from neuralnetworklibrary_amd._lib import check, lib, ptr, stream
    G, H, Hp, p = 4 * 50, 50, 64, 0.3
    g = torch.Generator().manual_seed(4)
    raw = torch.randn(G, H, generator=g).to(DEV)
    mask = ((torch.rand(G, H, generator=g) >= p).float() / (1 - p)).to(DEV)
    out = torch.full((G, Hp), float('nan'), device=DEV)
    check(lib.nnl_weight_drop(ptr(raw), H, ptr(mask), ptr(out), Hp, G, H, 0, 0.0, stream()))
    assert torch.equal(out[:, :H], raw * mask) and float(out[:, H:].abs().sum()) == 0.0
    outs = []
    for seed in (123, 123, 124):
        o = torch.empty(G, Hp, device=DEV)
        check(lib.nnl_weight_drop(ptr(raw), H, None, ptr(o), Hp, G, H, seed, p, stream()))
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    m = outs[0][:, :H] / raw
    keep = (m != 0)
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    assert_close(m[keep], torch.full_like(m[keep], 1 / (1 - p)), 1e-5, 0, 'kept elements are scaled by 1 / (1 - p)')
    back = torch.empty(G, H, device=DEV)                      # the backward call: padded source, dense destination, same seed
    check(lib.nnl_weight_drop(ptr(outs[0]), Hp, None, ptr(back), H, G, H, 123, p, stream()))
    assert_close(back, raw * m * m, 1e-5, 1e-6, 'the backward call applies the same mask')
    # through the layer, explicit mask: d raw = (d of the dropped matrix) * mask
    T, B, I = 4, 3, 6
    x = torch.randn(T, B, I, generator=g).to(DEV)
    w_ih, b1, b2 = torch.randn(G, I, generator=g).to(DEV) * 0.3, torch.randn(G, generator=g).to(DEV) * 0.1, torch.zeros(G, device=DEV)
    h0 = c0 = torch.zeros(1, B, H, device=DEV)
    r1 = (raw * 0.2).clone().requires_grad_(True)
    y1, _ = ops_text.lstm_layer(x, h0, c0, w_ih, r1, b1, b2, weight_mask=mask)
    y1.sum().backward()
    r2 = (raw * 0.2 * mask).clone().requires_grad_(True)       # the dropped matrix as a leaf, no mask
    y2, _ = ops_text.lstm_layer(x, h0, c0, w_ih, r2, b1, b2)
    y2.sum().backward()
    assert torch.equal(y1, y2)
    assert_close(r1.grad, r2.grad * mask, 1e-6, 1e-7, 'd raw = dW * mask')
