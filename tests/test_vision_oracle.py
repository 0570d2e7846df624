"""Pins the oracle's vision restatement (oracle/reference_nets.py) to goldens produced by the reference itself
(G5: ResNet blocks + stem, G6: ResNet-34 classifier incl. one Learner step).  CPU only."""
import numpy as np
import torch
import torch.nn as nn

from conftest import assert_close, load_golden
from oracle import reference_math as RM
from oracle import reference_nets as RNets
from oracle import synth


def run_block(tag, mod, x, g, rtol=1e-5, atol=1e-6, dev='cpu'):
    synth.fill_module_(mod)
    mod = mod.to(dev).train()
    x = x.to(dev).requires_grad_(True)
    y = mod(x)
    assert_close(y, g[tag + '.y'], rtol, atol, tag + '.y')
    y.backward(torch.from_numpy(g[tag + '.dy']).to(dev))
    assert_close(x.grad, g[tag + '.dx'], rtol, atol * 10, tag + '.dx')
    for n, p in mod.named_parameters():
        ref = g[tag + '.grad.' + n]
        assert_close(p.grad, ref, rtol * 10, atol + 1e-5 * np.abs(ref).max(), tag + '.grad.' + n)
    for n, b in mod.named_buffers():
        assert_close(b, g[tag + '.buf.' + n], rtol, atol, tag + '.buf.' + n)


def block_cases(NS):
    """(tag, module factory, input) for G5, built from namespace NS (oracle nets or product nets)."""
    conv = getattr(NS, 'HipConv2d', nn.Conv2d)
    ds_cls = getattr(NS, '_Downsample', nn.Sequential)
    ds = lambda: ds_cls(conv(8, 16, kernel_size=1, stride=2, bias=False), nn.BatchNorm2d(16))
    return [('bb', lambda: NS.BasicBlock(8, 8), synth.synth_input((2, 8, 14, 14), 1)),
            ('bbs', lambda: NS.BasicBlock(8, 16, 2, ds()), synth.synth_input((2, 8, 14, 14), 2)),
            ('bn', lambda: NS.Bottleneck(8, 4, 2, ds()), synth.synth_input((2, 8, 14, 14), 3))]


def test_g5_blocks_oracle():
    g = load_golden('g5_blocks')
    for tag, make, x in block_cases(RNets):
        run_block(tag, make(), x, g)
    stem = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                         nn.MaxPool2d(3, stride=2, padding=1))
    run_block('stem', stem, synth.synth_input((2, 3, 32, 32), 4), g)


def layer_group_of(name):
    """layer-group index of a parameter of the default-split ResNet classifier: body[:6] | body[6:] | head."""
    if name.startswith('head'):
        return 2
    return 0 if int(name.split('.')[1]) < 6 else 1


def assert_within_reference_gap(got, g, key, msg, slack=3.0, floor=1e-3):
    """|got - ref64| <= slack*|ref32 - ref64| + floor*scale: an fp32 implementation must sit as close to the exact
    (fp64) value of the reference network as the reference's own fp32 run does (training-mode BN at small batch is
    ill-conditioned, see oracle/gen_golden.py g6)."""
    r32, r64 = g[key + '.f32'], g[key + '.f64']
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    scale = np.maximum(np.maximum(np.abs(r64), np.abs(r64).max() * 1e-3), 1e-30)
    gap = np.abs(r32 - r64) / scale                        # the reference's own relative fp32 error, per element
    gap = np.maximum(gap, np.quantile(gap, 0.95))          # one fp32 run is one sample of that error: use its bulk level
    tol = (slack * gap + floor) * scale
    err = np.abs(got.reshape(r64.shape) - r64)
    bad = err > tol
    assert not bad.any(), '%s: %d elements outside the reference fp32/fp64 gap, worst err %.3e vs tol %.3e' % (
        msg, bad.sum(), (err - tol).max() + tol.flat[np.argmax(err - tol)], tol.flat[np.argmax(err - tol)])


def g6_inputs(g, dev='cpu'):
    N, S = int(g['N']), int(g['S'])
    return synth.synth_input((N, 3, S, S), 6).to(dev), (torch.arange(N) % 2).to(dev)


def check_g6(net, g, dev='cpu'):
    x, y = g6_inputs(g, dev)
    net.train()
    logits = net(x)
    loss = nn.CrossEntropyLoss()(logits, y)
    assert_within_reference_gap(logits, g, 'logits', 'logits')
    assert_within_reference_gap(loss, g, 'loss', 'loss')
    loss.backward()
    names = [n for n, _ in net.named_parameters()]
    assert names == [str(s) for s in g['param_names']]
    norms = np.array([p.grad.norm().item() for _, p in net.named_parameters()])
    assert_within_reference_gap(norms, g, 'grad_norms', 'grad norms')
    # eval-mode BatchNorm (running stats as left by the training forward above): well-conditioned, compared elementwise
    net.eval()
    for p in net.parameters():
        p.grad = None
    logits = net(x)
    loss = nn.CrossEntropyLoss()(logits, y)
    assert_within_reference_gap(logits, g, 'eval.logits', 'eval logits')
    assert_within_reference_gap(loss, g, 'eval.loss', 'eval loss')
    loss.backward()
    norms = np.array([p.grad.norm().item() for _, p in net.named_parameters()])
    assert_within_reference_gap(norms, g, 'eval.grad_norms', 'eval grad norms')
    sd = dict(net.named_parameters())
    for k in sorted(set(k[10:-4] for k in g if k.startswith('eval.grad.'))):
        assert_within_reference_gap(sd[k].grad.reshape(-1)[:2048], g, 'eval.grad.' + k, 'eval.grad.' + k)
    net.train()
    return names


def test_g6_resnet34_oracle_forward_backward_and_step():
    g = load_golden('g6_resnet34')
    S = int(g['S'])
    net = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, (512,), (0., 0.), probe_sz=(S, S))
    synth.fill_module_(net)
    names = check_g6(net, g)
    assert_close(net.body[1].running_mean, g['buf.body.1.running_mean'], 1e-5, 1e-7, 'running_mean')
    assert_close(net.body[1].running_var, g['buf.body.1.running_var'], 1e-5, 1e-7, 'running_var')
    assert_close(nn.CrossEntropyLoss()(net(g6_inputs(g)[0]), g6_inputs(g)[1]), g['loss.f32'], 1e-6, 1e-7, 'loss == reference fp32')
    # restated Optimizer.step: SGD momentum .9, per-layer-group lr, decoupled wd 1e-4 on reg AND bn groups (bn_wd=True)
    net = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, (512,), (0., 0.), probe_sz=(S, S))
    synth.fill_module_(net)
    net.train()
    params = [p for _, p in net.named_parameters()]
    x, y = g6_inputs(g)
    loss = nn.CrossEntropyLoss()(net(x), y)
    loss.backward()
    assert_close(np.array([loss.item()]), g['step_loss'], 1e-5, 1e-7, 'step loss')
    lr_g = [1e-3, 3e-3, 1e-2]
    lrs = [lr_g[layer_group_of(n)] for n in names]
    RM.optimizer_step(params, [p.grad for p in params], RM.OptimState(params), lrs, [1e-4] * len(params), 'sgd', momentum=0.9)
    sums = np.array([p.double().sum().item() for p in params])
    abs_sums = np.array([p.double().abs().sum().item() for p in params])
    assert_close(abs_sums, g['after.abs_sums'], 1e-6, 1e-9, 'abs sums after step')
    assert_close(sums, g['after.sums'], 1e-5, 1e-5, 'sums after step')


def test_g13b_oracle_first_steps_at_baseline_size():
    """the CPU oracle (restated nets + restated Optimizer.step) reproduces the first 3 steps of the REFERENCE's own 20-step
    curve at BASELINE configs[1]'s size (G13b: ResNet-34, 224 x 224, bs 64, SGD momentum, lr per layer group, wd)."""
    import torch.nn as nn
    from oracle import reference_math as RM, reference_nets as RNets
    g = load_golden('g13b_resnet34_curve')
    N, S = int(g['N']), int(g['S'])
    onet = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.), probe_sz=(S, S))
    synth.fill_reference_init_(onet, seed=int(g['init_seed']))
    onet.train()
    names = [n for n, _ in onet.named_parameters()]
    assert names == [str(s) for s in g['param_names']]
    params = [p for _, p in onet.named_parameters()]
    group = lambda n: 2 if n.startswith('head') else (0 if int(n.split('.')[1]) < 6 else 1)
    lrs = [float(g['lr'][group(n)]) for n in names]
    state = RM.OptimState(params)
    for i in range(3):
        x, y = synth.curve_batch_images(N, S, 1300 + i)
        for p in params:
            p.grad = None
        loss = nn.CrossEntropyLoss()(onet(x), y)
        loss.backward()
        RM.optimizer_step(params, [p.grad for p in params], state, lrs, [float(g['wd'])] * len(params), 'sgd')
        assert abs(loss.item() - g['losses.f32'][i]) <= 2e-5 * abs(g['losses.f32'][i]), (i, loss.item(), g['losses.f32'][i])


def test_g13c_oracle_first_steps_frozen_bn_at_baseline_size():
    """the CPU oracle reproduces the first 3 steps of the REFERENCE's frozen-BatchNorm curve at BASELINE configs[1]'s size (G13c: ResNet-34,
    224 x 224, bs 64, bn_freeze('all') + running statistics, SGD momentum, body lr large enough that the head-only curve leaves the full
    one by >= 20 %) — BatchNorm parameters are out of the optimizer (Learner.py:248-264), the layers run on their running statistics."""
    import torch.nn as nn
    from oracle import reference_math as RM, reference_nets as RNets
    g = load_golden('g13c_resnet34_frozen_bn_curve')
    assert (np.abs(g['losses.f32.headonly'] - g['losses.f32']) / g['losses.f32']).max() >= 0.2
    assert (np.abs(g['losses.f32'] - g['losses.f64']) / g['losses.f64']).max() < 3e-4
    N, S = int(g['N']), int(g['S'])
    onet = RNets.ImageClassificationNet(RNets.resnet34(), 2, 512, drops=(0., 0.), probe_sz=(S, S))
    synth.fill_reference_init_(onet, seed=int(g['init_seed']))
    synth.tame_residual_branches_(onet)
    onet.train()
    bn_params = set()
    for m in onet.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.training = False
            bn_params.update(id(p) for p in m.parameters())
    names = [n for n, _ in onet.named_parameters()]
    assert names == [str(s) for s in g['param_names']]
    group = lambda n: 2 if n.startswith('head') else (0 if int(n.split('.')[1]) < 6 else 1)
    train = [(n, p) for n, p in onet.named_parameters() if id(p) not in bn_params]
    params = [p for _, p in train]
    lrs = [float(g['lr'][group(n)]) for n, _ in train]
    state = RM.OptimState(params)
    for i in range(3):
        x, y = synth.curve_batch_images(N, S, 1400 + i)
        for p in onet.parameters():
            p.grad = None
        loss = nn.CrossEntropyLoss()(onet(x), y)
        loss.backward()
        RM.optimizer_step(params, [p.grad for p in params], state, lrs, [float(g['wd'])] * len(params), 'sgd')
        assert abs(loss.item() - g['losses.f32'][i]) <= 2e-5 * abs(g['losses.f32'][i]), (i, loss.item(), g['losses.f32'][i])
