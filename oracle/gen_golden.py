"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden-vector generator (build container only).

Runs the REAL reference (imported read-only from /root/reference under oracle/_ref_import.py's shims, torch
2.10 CPU — the reference pins torch 1.2, README.md:21; see SURVEY.md §8c for the caveat) on small seeded
inputs and writes inputs + expected outputs to tests/golden/*.npz.  Only data is written: no reference
source text is copied.  Usage:  python oracle/gen_golden.py [g1 g2 ...]   (default: all)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
R = _ref_import.load()
Learner = R['General.Learner'].Learner
Optimizer = R['General.Optimizer'].Optimizer


def A(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, {k: getattr(v, 'shape', None) for k, v in arrays.items()}, '%.1f KB' % (os.path.getsize(path) / 1024))


class FakeData:
    """Minimal data object honouring the Learner's data protocol (General/Learner.py:103-110,504,557,596)."""

    def __init__(self, train_batches, val_batches, bs, target_type):
        self.train_dl, self.val_dl, self.bs, self.target_type = train_batches, val_batches, bs, target_type


# ---------------------------------------------------------------------------------------------------------
def g1_collab():
    """G1: CollabFilterNet fwd / MSE / 4 grads / params after 3 Adam train1minibatch steps (wd=1e-4)."""
    CF = R['Applications.CollabFiltering']
    torch.manual_seed(101)
    n_user, n_item, D, bs = 50, 40, 8, 64
    net = CF.CollabFilterNet(n_user, n_item, D, [0.8, 5.2])
    with torch.no_grad():                      # larger than the 0.01-std init so the dot term matters
        for p in net.parameters():
            p.mul_(30.0)
    g = torch.Generator().manual_seed(7)
    xs = [torch.stack([torch.randint(0, n_user, (bs,), generator=g), torch.randint(0, n_item, (bs,), generator=g)], 1)
          for _ in range(3)]
    ys = [torch.randint(1, 6, (bs,), generator=g).float() for _ in range(3)]
    out = {'n_user': n_user, 'n_item': n_item, 'D': D, 'lo': 0.8, 'hi': 5.2}
    for k, p in net.state_dict().items():
        out['init.' + k] = A(p)
    for i in range(3):
        out['x%d' % i], out['y%d' % i] = A(xs[i]), A(ys[i])
    pred = net(xs[0])
    loss = torch.nn.MSELoss()(pred, ys[0])
    loss.backward()
    out['pred0'], out['loss0'] = A(pred), A(loss)
    for k, p in net.named_parameters():
        out['grad0.' + k] = A(p.grad)
    # no-range variant of the forward (output_range=None branch, CollabFiltering.py:201)
    net.output_range = None
    out['pred0_norange'] = A(net(xs[0]))
    net.output_range = [0.8, 5.2]
    # 3 steps of the reference Learner.train1minibatch with Adam, lr 1e-2, wd 1e-4
    data = FakeData(list(zip(xs, ys)), list(zip(xs, ys)), bs, 'cont')
    learner = Learner('/tmp/nnl_golden_g1', data, net, optimizer='Adam')
    learner.init_optimizer(wd=1e-4)
    losses = [learner.train1minibatch(xs[i], ys[i], 1e-2) for i in range(3)]
    out['step_losses'] = np.array(losses, dtype=np.float64)
    for k, p in net.state_dict().items():
        out['after3.' + k] = A(p)
    save('g1_collab', **out)


def g5_blocks():
    """G5: BasicBlock(8,8), strided BasicBlock(8,16,2)+downsample, stem conv7x7+BN+ReLU+maxpool: outputs, input and
    parameter gradients, BN running stats after one training-mode forward.  Weights/inputs from synth.py."""
    RN = R['Applications.VisionModels.retinanet']
    nn = torch.nn
    out = {}

    def run(tag, mod, x):
        synth.fill_module_(mod)
        mod.train()
        x = x.clone().requires_grad_(True)
        y = mod(x)
        dy = synth.synth_input(tuple(y.shape), 77 + len(out), 1.0)
        y.backward(dy)
        out[tag + '.y'], out[tag + '.dy'], out[tag + '.dx'] = A(y), A(dy), A(x.grad)
        for n, p in mod.named_parameters():
            out[tag + '.grad.' + n] = A(p.grad)
        for n, b in mod.named_buffers():
            out[tag + '.buf.' + n] = A(b)

    run('bb', RN.BasicBlock(8, 8), synth.synth_input((2, 8, 14, 14), 1))
    ds = nn.Sequential(nn.Conv2d(8, 16, kernel_size=1, stride=2, bias=False), nn.BatchNorm2d(16))
    run('bbs', RN.BasicBlock(8, 16, 2, ds), synth.synth_input((2, 8, 14, 14), 2))
    ds = nn.Sequential(nn.Conv2d(8, 16, kernel_size=1, stride=2, bias=False), nn.BatchNorm2d(16))
    run('bn', RN.Bottleneck(8, 4, 2, ds), synth.synth_input((2, 8, 14, 14), 3))
    stem = nn.Sequential(nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
                         nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
    run('stem', stem, synth.synth_input((2, 3, 32, 32), 4))
    save('g5_blocks', **out)


def g6_resnet34():
    """G6: the reference's ImageClassificationNet over a ResNet-34 body (RetinaNet(.,BasicBlock,[3,4,6,3]) cut at
    child 8, split at 6 = default_cut/default_split of a torchvision ResNet, Vision.py:1211-1212,1225-1228) + default
    head (dropout 0) on x [4,3,96,96]: logits, CE loss, per-parameter gradient norms and gradient slices — each from
    the reference in fp32 AND from the same reference modules cast to fp64 (training-mode BN at small batch is
    ill-conditioned: fp32 and fp64 runs of the reference itself differ by ~1e-3 on early-layer gradients, so tests
    bound the error of any fp32 implementation by the reference's own fp32-vs-fp64 gap).  Then one real
    Learner.train1minibatch (SGD momentum 0.9, lr [1e-3,3e-3,1e-2], wd 1e-4) and post-step parameter checksums."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    N, S = 4, 96

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'

    def make(dtype):
        arch = RN.RetinaNet(2, RN.BasicBlock, [3, 4, 6, 3])
        net = V.ImageClassificationNet(D, arch, head=[[512], [0., 0.]], cutpoint=8, splits=[6])
        synth.fill_module_(net)
        return net.to(dtype).train()

    x, y = synth.synth_input((N, 3, S, S), 6), torch.arange(N) % 2
    out = {'N': N, 'S': S}
    slices = ['body.0.weight', 'body.4.0.conv1.weight', 'body.5.0.downsample.0.weight', 'body.7.2.conv2.weight',
              'head.2.lins.0.lin.weight', 'head.2.final_lin.bias']
    for tag, dtype in [('f32', torch.float32), ('f64', torch.float64)]:
        net = make(dtype)
        logits = net(x.to(dtype))
        loss = torch.nn.CrossEntropyLoss()(logits, y)
        loss.backward()
        out['logits.' + tag], out['loss.' + tag] = A(logits).astype(np.float64), A(loss).astype(np.float64)
        out['grad_norms.' + tag] = np.array([p.grad.norm().item() for _, p in net.named_parameters()], dtype=np.float64)
        sd = dict(net.named_parameters())
        for n in slices:
            out['grad.%s.%s' % (n, tag)] = A(sd[n].grad).reshape(-1)[:2048].astype(np.float64)
        # eval-mode BatchNorm (running statistics: no batch-statistics chaos) — a WELL-conditioned full-depth
        # forward/backward used for the strict elementwise gradient comparison
        net.eval()
        for p in net.parameters():
            p.grad = None
        logits = net(x.to(dtype))
        loss = torch.nn.CrossEntropyLoss()(logits, y)
        loss.backward()
        out['eval.logits.' + tag], out['eval.loss.' + tag] = A(logits).astype(np.float64), A(loss).astype(np.float64)
        out['eval.grad_norms.' + tag] = np.array([p.grad.norm().item() for _, p in net.named_parameters()], dtype=np.float64)
        for n in slices:
            out['eval.grad.%s.%s' % (n, tag)] = A(sd[n].grad).reshape(-1)[:2048].astype(np.float64)
        net.train()
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
            out['buf.body.1.running_mean'] = A(net.body[1].running_mean)
            out['buf.body.1.running_var'] = A(net.body[1].running_var)
            out['n_layer_groups'] = len(net.layer_groups)
    # one real optimizer step through the reference Learner (fresh fp32 net)
    net = make(torch.float32)
    d = D(); d.train_dl = [(x, y)]; d.val_dl = [(x, y)]
    learner = Learner('/tmp/nnl_golden_g6', d, net, optimizer='SGD_Mom')
    learner.init_optimizer(wd=1e-4)
    out['step_loss'] = np.array([learner.train1minibatch(x, y, [1e-3, 3e-3, 1e-2])], dtype=np.float64)
    out['after.sums'] = np.array([p.double().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
    out['after.abs_sums'] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
    # the same step with every BatchNorm frozen (Learner.bn_freeze('all') + the m.training=False loop of
    # train_gen_sched, Learner.py:248-264,589-591): running statistics instead of batch statistics => well conditioned
    net = make(torch.float32)
    learner = Learner('/tmp/nnl_golden_g6', d, net, optimizer='SGD_Mom')
    learner.bn_freeze('all')
    learner.init_optimizer(wd=1e-4)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.training = False
    out['frozen.step_loss'] = np.array([learner.train1minibatch(x, y, [1e-3, 3e-3, 1e-2])], dtype=np.float64)
    out['frozen.after.sums'] = np.array([p.double().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
    out['frozen.after.abs_sums'] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
    save('g6_resnet34', **out)


def g13_resnet34_curve():
    """G13: 20 consecutive `Learner.train1minibatch` steps of the REFERENCE at BASELINE configs[1]'s own size — ResNet-34 body +
    default head (dropout 0), 224x224, bs 64, SGD momentum 0.9, lr [1e-3, 3e-3, 1e-2], wd 1e-4, BatchNorm in training mode — in
    fp32 and, with the same modules cast to fp64, in fp64: per-step losses, post-run parameter checksums.  Four synthetic batches
    (synth_input tags 130..133, labels (7 i + b) mod 2) are cycled.  ~10 min on 8 cores (fp64 convolutions)."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    N, S, STEPS = 64, 224, 20

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'

    xs = [synth.synth_input((N, 3, S, S), 130 + b) for b in range(4)]
    ys = [(torch.arange(N) * 7 + b) % 2 for b in range(4)]
    out = {'N': N, 'S': S, 'steps': STEPS, 'lr': np.array([1e-3, 3e-3, 1e-2]), 'wd': 1e-4}
    for tag, dtype in [('f32', torch.float32), ('f64', torch.float64)]:
        arch = RN.RetinaNet(2, RN.BasicBlock, [3, 4, 6, 3])
        net = V.ImageClassificationNet(D, arch, head=[[512], [0., 0.]], cutpoint=8, splits=[6])
        synth.fill_module_(net, seed=5)
        net = net.to(dtype).train()
        d = D(); d.train_dl = [(xs[0], ys[0])]; d.val_dl = d.train_dl
        learner = Learner('/tmp/nnl_golden_g13', d, net, optimizer='SGD_Mom')
        learner.init_optimizer(wd=1e-4)
        losses = []
        for i in range(STEPS):
            losses.append(learner.train1minibatch(xs[i % 4].to(dtype), ys[i % 4], [1e-3, 3e-3, 1e-2]))
            print(tag, i, losses[-1], flush=True)
        out['losses.' + tag] = np.array(losses, dtype=np.float64)
        out['after.abs_sums.' + tag] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        out['after.sums.' + tag] = np.array([p.double().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
    save('g13_resnet34_curve', **out)


def g3_tabular():
    """G2+G3: EmbeddingDrop renorm (weight before/after, output) and StructuredDataNet (cards [20,5,4,13], n_cont 3, fc
    [32,16,1], range [5,12], dropout 0): forward, MSE loss, all gradients, then 3 Learner.train1minibatch steps (Adam,
    lr 1e-3, wd 1e-3), BN running stats and every parameter afterwards."""
    L = R['General.Layers']
    SD = R['Applications.StructuredData']
    out = {}
    # --- G2: single EmbeddingDrop column with rows above max_norm
    emb = L.EmbeddingDrop(20, 5, 0.0, 1.0, 1.5)
    synth.fill_module_(emb, seed=3)
    with torch.no_grad():
        emb.emb.weight.mul_(1.8)
    x = torch.from_numpy(np.random.RandomState(11).randint(0, 20, size=12).astype(np.int64))
    out['g2.w_before'] = A(emb.emb.weight)
    out['g2.x'] = A(x)
    out['g2.y'] = A(emb(x))
    out['g2.w_after'] = A(emb.emb.weight)
    # --- G3
    cards, n_cont, bs = [20, 5, 4, 13], 3, 64
    labels = [{i: i for i in range(c)} for c in cards]
    net = SD.StructuredDataNet('cont', len(cards), n_cont, labels, [32, 16, 1], output_range=[5, 12])
    synth.fill_module_(net, seed=4)
    rs = np.random.RandomState(12)
    batches = []
    for i in range(3):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=bs) for c in cards], 1).astype(np.int64))
        xcont = torch.from_numpy(rs.standard_normal((bs, n_cont)).astype(np.float32) * 2 + 1)
        y = torch.from_numpy((5 + 7 * rs.rand(bs)).astype(np.float32))
        batches.append(([xcat, xcont], y))
        out['xcat%d' % i], out['xcont%d' % i], out['y%d' % i] = A(xcat), A(xcont), A(y)
    out['cards'], out['n_cont'] = np.array(cards), n_cont
    out['emb_dims'] = np.array([e.emb.weight.shape[1] for e in net.embeddings])
    net.train()
    (xcat, xcont), y = batches[0]
    pred = net(xcat, xcont)
    loss = torch.nn.MSELoss()(pred, y)
    loss.backward()
    out['pred0'], out['loss0'] = A(pred), A(loss)
    for n, p in net.named_parameters():
        out['grad0.' + n] = A(p.grad)
    for n, b in net.named_buffers():
        out['buf0.' + n] = A(b)
    for n, p in net.named_parameters():
        out['after_fwd0.' + n] = A(p)                     # embeddings were renormed in place by the forward
    # fresh net, 3 real Learner steps
    net = SD.StructuredDataNet('cont', len(cards), n_cont, labels, [32, 16, 1], output_range=[5, 12])
    synth.fill_module_(net, seed=4)
    data = FakeData(batches, batches, bs, 'cont')
    learner = Learner('/tmp/nnl_golden_g3', data, net, optimizer='Adam')
    learner.init_optimizer(wd=1e-3)
    net.train()
    out['step_losses'] = np.array([learner.train1minibatch(xb, yb, [1e-3, 3e-3]) for xb, yb in batches], dtype=np.float64)
    for n, p in net.state_dict().items():
        out['after3.' + n] = A(p)
    net.eval()
    out['eval_pred0'] = A(net(*batches[0][0]))
    save('g3_tabular', **out)


def g8_detection():
    """G8: anchors for 64x64 and 512x512 (shape, first/last rows, checksums); SSD_loss (beta .5, alpha .25, gamma 2) and
    its gradients wrt reg / clas on 3 images with {0, 3, 2} boxes, K=5, on the 64x64 anchor set; per-image pieces from
    match_anchors_objects / focal_loss_retina / smoothL1_loss_retina."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    out = {}
    gen = RN.AnchorGenerator()
    for sz in [64, 512]:
        a = A(gen(torch.zeros(1, 3, sz, sz)))
        out['anchors%d.shape' % sz] = np.array(a.shape)
        out['anchors%d.head' % sz], out['anchors%d.tail' % sz] = a[:40].copy(), a[-40:].copy()
        out['anchors%d.sum' % sz] = a.astype(np.float64).sum(0)
        out['anchors%d.abs_sum' % sz] = np.abs(a.astype(np.float64)).sum(0)
    anchors = gen(torch.zeros(1, 3, 64, 64))
    Na, K, bs, M = len(anchors), 5, 3, 4
    rs = np.random.RandomState(21)
    boxes = -np.ones((bs, M, 4), np.float32); cats = -np.ones((bs, M), np.int64)
    # image 0: no objects; image 1: 3 objects; image 2: 2 objects (one tiny: exercises the w,h >= 1 clamp)
    boxes[1, :3] = [[4, 6, 40, 44], [20, 10, 60, 34], [30, 30, 46, 62]]; cats[1, :3] = [2, 0, 4]
    boxes[2, :2] = [[10.2, 12.7, 10.9, 13.1], [0, 0, 33, 31]]; cats[2, :2] = [1, 3]
    reg = torch.from_numpy(rs.standard_normal((bs, Na, 4)).astype(np.float32) * 0.5).requires_grad_(True)
    clas = torch.from_numpy(rs.uniform(0, 1, (bs, Na, K)).astype(np.float32))
    clas.view(-1)[::97] = 1e-6; clas.view(-1)[::89] = 1 - 1e-6          # values outside the clamp range
    clas.requires_grad_(True)
    B, Cc = torch.from_numpy(boxes), torch.from_numpy(cats)
    lf = V.SSD_loss(0.5, 0.25, 2.0)
    loss = lf([anchors, reg, clas], [B, Cc])
    loss.backward()
    out.update({'boxes': boxes, 'cats': cats, 'reg': A(reg), 'clas': A(clas), 'loss': A(loss), 'reg_loss': A(lf.reg_loss),
                'clas_loss': A(lf.clas_loss), 'dreg': A(reg.grad), 'dclas': A(clas.grad)})
    for i in range(bs):
        bb, cc = B[i][Cc[i] >= 0], Cc[i][Cc[i] >= 0]
        pos, neg, matches = V.match_anchors_objects(bb, anchors)
        out['img%d.pos' % i], out['img%d.neg' % i], out['img%d.matches' % i] = A(pos), A(neg), A(matches)
        r, c = V.ssd1(anchors, bb, cc, reg[i].detach(), clas[i].detach())
        out['img%d.reg_loss' % i], out['img%d.clas_loss' % i] = A(r), A(c)
    save('g8_detection', **out)


class _FixedDrop(torch.nn.Module):
    "stands in for an nn.Dropout inside a reference module: returns pre-drawn masks in call order"
    def __init__(self, masks):
        super().__init__()
        self.masks, self.i = masks, 0
    def forward(self, t):
        m = self.masks[self.i % len(self.masks)]
        self.i += 1
        assert tuple(m.shape) == tuple(t.shape), (m.shape, t.shape)
        return t * m


def _mask(shape, p, tag):
    keep = (np.random.RandomState(tag).rand(*shape) >= p).astype(np.float32)
    return torch.from_numpy(keep / (1 - p))


def g7_text():
    """G7: WeightDropLSTM1(8,12) with an injected weight mask (y, hT, cT, all grads); LSTM_Encoder(V=50, emb 8, hid 12,
    3 layers) in training mode with every dropout mask injected, two consecutive batches (hidden-state carry);
    full-size LanguageModelNet (400/1150/3, V=60, dropout 0): RegSeqCrossEntropyLoss(2,1) forward/backward and two
    Learner.train1minibatch steps (Adam betas (0.8,0.99), lr [1e-3,2e-3], wd 1e-6, clip 0.4)."""
    TX = R['Applications.Text']
    out = {}
    # (a) one weight-dropped layer
    T, B, I, H = 6, 4, 8, 12
    m = TX.WeightDropLSTM1(I, H, 0.5)
    m.clear_non_raw()
    synth.fill_module_(m, seed=7)
    wmask = _mask((4 * H, H), 0.5, 31)
    m.weight_drop = _FixedDrop([wmask])
    x = synth.synth_input((T, B, I), 41).requires_grad_(True)
    h0, c0 = synth.synth_input((1, B, H), 42, 0.5), synth.synth_input((1, B, H), 43, 0.5)
    y, (hT, cT) = m(x, (h0, c0))
    dy = synth.synth_input((T, B, H), 44)
    (y * dy).sum().backward()
    out.update({'a.wmask': A(wmask), 'a.y': A(y), 'a.hT': A(hT), 'a.cT': A(cT), 'a.dx': A(x.grad)})
    for n, p in m.named_parameters():
        out['a.grad.' + n] = A(p.grad)
    out['a.param_names'] = np.array([n for n, _ in m.named_parameters()])
    # (b) encoder, two consecutive batches, all masks injected
    V, E, Hh, bs, seq = 50, 8, 12, 4, 5
    enc = TX.LSTM_Encoder(V, E, Hh, 3, 1, [0.3, 0.3, 0.4, 0.3], bs)
    for l in enc.lstms:
        l.clear_non_raw()
    synth.fill_module_(enc, seed=8)
    sizes = [E, Hh, Hh, E]
    masks = {'emb_rows': _mask((V, 1), 0.3, 51), 'emb_locked': _mask((1, bs, E), 0.3, 52),
             'weights': [_mask((4 * sizes[i + 1], sizes[i + 1]), 0.4, 53 + i) for i in range(3)],
             'hidden': [_mask((1, bs, sizes[i + 1]), 0.3, 56 + i) for i in range(3)]}
    enc.word_embed.drop1 = _FixedDrop([masks['emb_rows']])
    enc.word_embed.drop2.drop = _FixedDrop([masks['emb_locked']])
    for i, l in enumerate(enc.lstms):
        l.weight_drop = _FixedDrop([masks['weights'][i]])
    enc.hidden_drop.drop = _FixedDrop(masks['hidden'])
    enc.train()
    rs = np.random.RandomState(61)
    for b in range(2):
        xb = torch.from_numpy(rs.randint(0, V, (bs, seq)).astype(np.int64))
        ob = enc(xb)
        out['b.x%d' % b], out['b.out%d' % b] = A(xb), A(ob)
    dyb = synth.synth_input(tuple(ob.shape), 62)
    (ob * dyb).sum().backward()
    for n, p in enc.named_parameters():
        out['b.grad.' + n] = A(p.grad)
    for k in ['emb_rows', 'emb_locked']:
        out['b.mask.' + k] = A(masks[k])
    for i in range(3):
        out['b.mask.weights%d' % i], out['b.mask.hidden%d' % i] = A(masks['weights'][i]), A(masks['hidden'][i])
    out['b.h_final0'], out['b.c_final2'] = A(enc.h[0]), A(enc.c[2])
    # (c) full-size language model, dropout 0
    class D:
        pass
    Vc, bs, seq = 60, 4, 7
    d = D(); d.stoi = {('tok%d' % i): i for i in range(Vc)}; d.stoi['_pad_'] = 1; d.bs = bs; d.target_type = 'lang_model'
    del d.stoi['tok1']
    net = TX.LanguageModelNet(d, enc_drops=[0., 0., 0., 0.], dec_drop=0.)
    net.clear_non_raw()
    synth.fill_module_(net, seed=9)
    with torch.no_grad():
        net.enc.word_embed.embed.weight.mul_(0.3)
    rs = np.random.RandomState(71)
    stream_ = rs.randint(0, Vc, (bs, 2 * seq + 1)).astype(np.int64)
    batches = [(torch.from_numpy(stream_[:, i * seq:(i + 1) * seq].copy()), torch.from_numpy(stream_[:, i * seq + 1:(i + 1) * seq + 1].copy()))
               for i in range(2)]
    for i, (xb, yb) in enumerate(batches):
        out['c.x%d' % i], out['c.y%d' % i] = A(xb), A(yb)
    lf = TX.RegSeqCrossEntropyLoss(2.0, 1.0)
    net.train()
    outp = net(batches[0][0])
    loss = lf(outp, batches[0][1])
    loss.backward()
    out['c.loss'], out['c.ce'] = A(loss), A(lf.cross_entropy)
    out['c.preds_slice'] = A(outp[0])[:, :, :2].copy()
    out['c.param_names'] = np.array([n for n, _ in net.named_parameters()])
    out['c.grad_norms'] = np.array([p.grad.norm().item() for _, p in net.named_parameters()], dtype=np.float64)
    sd = dict(net.named_parameters())
    out['c.grad.emb'] = A(sd['enc.word_embed.embed.weight'].grad)
    out['c.grad.whh0_slice'] = A(sd['enc.lstms.0.lstm.weight_hh_l0_raw'].grad)[:64, :64].copy()
    out['c.grad.bias2'] = A(sd['enc.lstms.2.lstm.bias_ih_l0'].grad)
    # two real Learner steps on a fresh net (state carried from batch 0 to batch 1)
    net = TX.LanguageModelNet(d, enc_drops=[0., 0., 0., 0.], dec_drop=0.)
    net.clear_non_raw()
    synth.fill_module_(net, seed=9)
    with torch.no_grad():
        net.enc.word_embed.embed.weight.mul_(0.3)
    d.train_dl, d.val_dl = batches, batches
    learner = Learner('/tmp/nnl_golden_g7', d, net, optimizer='Adam', loss_func=TX.RegSeqCrossEntropyLoss(2.0, 1.0))
    learner.init_optimizer(wd=1e-6, clip=0.4)
    net.train()
    out['c.step_losses'] = np.array([learner.train1minibatch(xb, yb, [1e-3, 2e-3], betas_batch=(0.8, 0.99)) for xb, yb in batches],
                                    dtype=np.float64)
    out['c.after.abs_sums'] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
    out['c.after.emb'] = A(net.enc.word_embed.embed.weight)
    save('g7_text', **out)


def g9_host_logic():
    """G9+G10: Optimizer.step (wd / bn_wd / clip, SGD-momentum and Adam) on a 2-layer-group toy model; Learner.get_sched
    for its four types (scalar and vector); fit_one_cycle lr / mom / betas schedules; 2-epoch Learner.fit,
    fit_cycles and find_lr loss curves on a toy regression problem (reference Learner driving a plain torch model)."""
    Core = R['General.Core']
    out = {}

    def toy():
        torch.manual_seed(0)
        g1 = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Tanh())
        g2 = torch.nn.Sequential(torch.nn.Linear(7, 1), torch.nn.Flatten(0))
        net = torch.nn.Sequential(g1, g2)
        synth.fill_module_(net, seed=11)
        net.layer_groups = [g1, g2]
        net.param_groups = Core.separate_bn_layers(net.layer_groups)
        return net

    rs = np.random.RandomState(3)
    X = torch.from_numpy(rs.standard_normal((40, 5)).astype(np.float32))
    Y = torch.from_numpy(rs.standard_normal(40).astype(np.float32))
    out['X'], out['Y'] = A(X), A(Y)
    batches = [(X[i:i + 8], Y[i:i + 8]) for i in range(0, 36, 8)]          # 4 full batches + one of 4 (ragged)
    # --- Optimizer.step variants
    for tag, opt, kw in [('sgd', 'SGD_Mom', dict(wd=[1e-2, 3e-2], bn_wd=True, clip=0.5)),
                         ('adam', 'Adam', dict(wd=1e-2, bn_wd=False, clip=None))]:
        net = toy()
        LN = R['General.Learner']
        o = Optimizer(LN.opt_dict[opt], net)
        o.set_params([1e-1, 3e-1], **kw)
        for step in range(3):
            o.opt.zero_grad()
            ((net(X[:8]) - Y[:8]) ** 2).mean().backward()
            o.step()
        for n, p in net.named_parameters():
            out['opt.%s.%s' % (tag, n)] = A(p)
    # --- schedules
    Lr = Learner.get_sched
    for st in ['linear', 'cos', 'exp', 'poly']:
        out['sched.%s.scalar' % st] = np.array(Lr(st, 7, 1e-3, 1e-1), dtype=np.float64)
        out['sched.%s.vector' % st] = np.array(Lr(st, 5, [1e-3, 2e-3], [1e-1, 4e-1]), dtype=np.float64)
    data = FakeData(batches, batches[:2], 8, 'cont')
    for opt in ['SGD_Mom', 'Adam']:
        learner = Learner('/tmp/nnl_golden_g9', data, toy(), optimizer=opt)
        captured = {}
        learner.train_gen_sched = lambda lr, mom, betas, *a, **k: captured.update(lr=lr, mom=mom, betas=betas)
        learner.fit_one_cycle([1e-2, 3e-2], 2, wd=1e-3)
        out['onecycle.%s.lr' % opt] = np.array(captured['lr'], dtype=np.float64)
        if captured['mom'] is not None:
            out['onecycle.%s.mom' % opt] = np.array(captured['mom'], dtype=np.float64)
        if captured['betas'] is not None:
            out['onecycle.%s.betas' % opt] = np.array(captured['betas'], dtype=np.float64)
    # --- full fit loops (loss per minibatch, lr bookkeeping, final weights)
    learner = Learner('/tmp/nnl_golden_g9', data, toy(), optimizer='SGD_Mom')
    learner.fit([3e-2, 1e-1], 2, wd=1e-3, clip=1.0, momentum=0.8)
    out['fit.loss_sched'] = np.array(learner.loss_sched, dtype=np.float64)
    out['fit.moving_avg'] = np.array([learner.moving_avg_loss])
    out['fit.w'] = np.concatenate([A(p).reshape(-1) for p in learner.model.parameters()])
    learner = Learner('/tmp/nnl_golden_g9', data, toy(), optimizer='Adam')
    learner.fit_cycles(3e-2, 1e-3, 2, cycle_type='cos', base_length=1, cycle_mult=2, wd=1e-3, betas=(0.8, 0.99))
    out['cycles.loss_sched'] = np.array(learner.loss_sched, dtype=np.float64)
    out['cycles.lr_sched'] = np.array(learner.lr_sched, dtype=np.float64)
    out['cycles.w'] = np.concatenate([A(p).reshape(-1) for p in learner.model.parameters()])
    import matplotlib
    matplotlib.use('Agg')
    learner = Learner('/tmp/nnl_golden_g9', data, toy(), optimizer='SGD_Mom')
    w_before = np.concatenate([A(p).reshape(-1) for p in learner.model.parameters()])
    learner.find_lr(lr_min=1e-4, lr_max=1.0, length=8, break_fac=None)
    out['findlr.loss_sched'] = np.array(learner.loss_sched, dtype=np.float64)
    out['findlr.lr_sched'] = np.array(learner.lr_sched, dtype=np.float64)
    out['findlr.restored'] = np.array([np.abs(np.concatenate([A(p).reshape(-1) for p in learner.model.parameters()]) - w_before).max()])
    ev = learner.evaluate('val')
    out['evaluate.val'] = np.array(ev[0:1], dtype=np.float64)
    save('g9_host_logic', **out)


def _ragged(prefix, out, boxes, classes, scores):
    "store one image's prediction lists as arrays"
    out[prefix + '.boxes'] = np.array(boxes, dtype=np.float32).reshape(-1, 4)
    out[prefix + '.classes'] = np.array(classes, dtype=np.int64).reshape(-1)
    out[prefix + '.scores'] = np.array(scores, dtype=np.float32).reshape(-1)


def g11_bbox_inference():
    """G11: detection inference post-processing.  BBoxPredictor.__call__ (retinanet.py:733-812: threshold, decode, clip,
    drop empty, nms) on synthetic activations over the 128x128 anchor set (3 images, K=4; image 2 has no candidate) for
    four parameter sets (defaults / rel_thresh / inc+dup / small top_k); nms() alone on a hand-made cluster list; mAP1 and
    mAP (Vision.py:1696-1800) on a small prediction / target set; ComputeMaxOverlaps (Vision.py:1666-1694)."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    out = {}
    gen = RN.AnchorGenerator()
    img = torch.zeros(3, 3, 128, 128)
    anchors = gen(img)
    Na, K = len(anchors), 4
    rs = np.random.RandomState(31)
    reg = torch.from_numpy((rs.standard_normal((3, Na, 4)) * 0.6).astype(np.float32))
    logits = rs.standard_normal((3, Na, K)) * 1.6 - 2.6
    clas = torch.from_numpy((1 / (1 + np.exp(-logits))).astype(np.float32))
    clas[2] = 0.01                                                    # image 2: nothing above the threshold
    out.update({'anchors': A(anchors), 'reg': A(reg), 'clas': A(clas), 'img_hw': np.array([128, 128])})
    cases = {
        'default': dict(thresh=0.05, max_overlap=0.5, rel_thresh=None, top_k=1000, max_boxes=20, dup=None, inc=None),
        'rel': dict(thresh=0.3, max_overlap=0.4, rel_thresh=[0.4, 0.6], top_k=300, max_boxes=1000, dup=None, inc=None),
        'incdup': dict(thresh=0.2, max_overlap=0.6, rel_thresh=None, top_k=1000, max_boxes=1000,
                       dup=[0.5, [(0, 1), (1, 0), (2, 3), (3, 2)]], inc=[0.8, [1]]),
        'topk': dict(thresh=0.05, max_overlap=0.5, rel_thresh=[0.2, 0.2], top_k=40, max_boxes=10, dup=None, inc=None),
    }
    pred = RN.BBoxPredictor()
    for name, kw in cases.items():
        PB, PC, CS = pred(img, reg, clas, anchors, **kw)
        for i in range(3):
            _ragged('%s.img%d' % (name, i), out, PB[i], PC[i], CS[i])
    # nms alone: clusters of near-duplicates of 2 classes + one contained box
    base = np.array([[10, 10, 50, 50], [12, 11, 52, 49], [11, 12, 49, 51], [60, 60, 100, 110], [61, 62, 99, 108],
                     [10, 10, 50, 50], [20, 20, 40, 40], [15, 15, 45, 45], [70, 5, 120, 40], [72, 6, 118, 41]], np.float32)
    cls = np.array([0, 0, 0, 1, 1, 1, 0, 0, 2, 2], np.int64)
    sc = np.array([0.9, 0.8, 0.85, 0.7, 0.75, 0.6, 0.5, 0.95, 0.3, 0.31], np.float32)
    out.update({'nms.in_boxes': base, 'nms.in_classes': cls, 'nms.in_scores': sc})
    for name, kw in {'a': dict(max_overlap=0.5), 'b': dict(max_overlap=0.3, rel_thresh=[0.5, 0.9]),
                     'c': dict(max_overlap=0.7, inc=[0.9, []], dup=[0.4, [(0, 1), (1, 0)]], max_boxes=4)}.items():
        b, c, s_ = RN.nms(torch.from_numpy(base), torch.from_numpy(cls), torch.from_numpy(sc), **kw)
        _ragged('nms.' + name, out, b, c, s_)
    # mAP
    targets = [[(np.array([10, 10, 50, 50], np.float32), 0), (np.array([60, 60, 100, 110], np.float32), 1)],
               [(np.array([5, 5, 30, 40], np.float32), 0)],
               []]
    predictions = [[[np.array([11, 10, 49, 52], np.float32), np.array([58, 61, 101, 108], np.float32), np.array([0, 0, 20, 20], np.float32)],
                    [0, 1, 0], [0.9, 0.8, 0.3]],
                   [[np.array([6, 4, 31, 41], np.float32), np.array([5, 5, 30, 40], np.float32)], [0, 1], [0.7, 0.6]],
                   [[np.array([1, 1, 9, 9], np.float32)], [1], [0.2]]]
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m_all = V.mAP(predictions, targets, {0: 'a', 1: 'b'})
        m_two = V.mAP(predictions, targets, {0: 'a', 1: 'b'}, thresholds=[0.5, 0.75])
    out['map.coco'], out['map.two'] = np.array([m_all]), np.array([m_two])
    targs0 = [[t[0] for t in T if t[1] == 0] for T in targets]
    preds0 = [[b for b, c in zip(P[0], P[1]) if c == 0] for P in predictions]
    scores0 = [[s_ for s_, c in zip(P[2], P[1]) if c == 0] for P in predictions]
    out['map1.cat0_t50'] = np.array([V.mAP1(targs0, preds0, scores0, 0.5)])
    # ComputeMaxOverlaps
    objs = -torch.ones(2, 3, 4)
    objs[0, :2] = torch.tensor([[10., 12., 60., 70.], [30., 30., 46., 62.]])
    objs[1, :1] = torch.tensor([[0., 0., 127., 100.]])
    cmo = V.ComputeMaxOverlaps()
    out['cmo.objects'] = A(objs)
    out['cmo.value'] = A(cmo([anchors, None, None], [objs, None])).reshape(1)
    out['cmo.list'] = np.array(cmo.max_overlaps, dtype=np.float32)
    save('g11_bbox_inference', **out)


def g12_objectdetectionnet():
    """G12: the reference's ObjectDetectionNet (ResNet-50 Bottleneck body + FPN + re-initialised heads, Vision.py:1382-1471) on
    x [2,3,64,64], K=3: reg / clas activations, SSD_loss (0.5, 0.25, 2.0) and per-parameter gradient norms, in fp32 AND fp64,
    with BatchNorm in training mode (fp32-vs-fp64 gap criterion, as G6) and in eval mode (well conditioned: also gradient
    slices).  The constructor's `vmods.retinanet.retinanet()` loads a COCO checkpoint that is an LFS pointer in the snapshot
    (retinanet.py:430-435): it is replaced HERE by the same architecture with seeded weights (synth.fill_detection_net_:
    every activation O(1), so the comparison is not dominated by cancellation of ~1e3-sized head activations)."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    RN.retinanet = lambda *a, **k: RN.RetinaNet(80, RN.Bottleneck, [3, 4, 6, 3])
    N, S, K = 2, 64, 3
    x = synth.synth_input((N, 3, S, S), 12)
    boxes = -np.ones((N, 2, 4), np.float32); cats = -np.ones((N, 2), np.int64)
    boxes[0] = [[4, 6, 40, 44], [20, 10, 60, 34]]; cats[0] = [2, 0]
    boxes[1, 0] = [8, 30, 50, 62]; cats[1, 0] = 1
    B, Cc = torch.from_numpy(boxes), torch.from_numpy(cats)
    out = {'N': N, 'S': S, 'K': K, 'boxes': boxes, 'cats': cats}
    slices = ['layer0.0.weight', 'layer2.0.conv2.weight', 'fpn.P5_1.weight', 'fpn.P3_2.weight', 'classifier.conv1.weight',
              'classifier.output.bias', 'regressor.output.weight']
    for tag, dtype in [('f32', torch.float32), ('f64', torch.float64)]:
        torch.manual_seed(0)
        net = V.ObjectDetectionNet(K)
        synth.fill_detection_net_(net)          # well-conditioned seeded weights: activations stay O(1) in eval mode too
        net = net.to(dtype)
        sd = dict(net.named_parameters())
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
            out['slice_names'] = np.array([n for n in slices if n in sd])
        for mode in ['train', 'eval']:
            net.train() if mode == 'train' else net.eval()
            for p in net.parameters():
                p.grad = None
            anchors, reg, clas = net(x.to(dtype))
            lf = V.SSD_loss(0.5, 0.25, 2.0)
            loss = lf([anchors, reg, clas], [B.to(dtype), Cc])
            loss.backward()
            pre = '%s.' % mode
            out[pre + 'reg.' + tag], out[pre + 'clas.' + tag] = A(reg).astype(np.float64), A(clas).astype(np.float64)
            out[pre + 'loss.' + tag] = A(loss).astype(np.float64).reshape(1)
            out[pre + 'grad_norms.' + tag] = np.array([0.0 if p.grad is None else p.grad.norm().item() for _, p in net.named_parameters()], dtype=np.float64)
            if mode == 'eval':
                for n in slices:
                    if n in sd and sd[n].grad is not None:
                        out['eval.grad.%s.%s' % (n, tag)] = A(sd[n].grad).reshape(-1)[:1024].astype(np.float64)
        if tag == 'f32':
            out['anchors.shape'] = np.array(A(anchors).shape)
    save('g12_objectdetectionnet', **out)


def g4_fcnet_and_g10_fit_curves():
    """G4: FullyConnectedNet([20,16,8,3], drops [.3,.2,.1], pre_bn) with every nn.Dropout replaced by pre-drawn masks: forward,
    loss, all gradients, BN running stats (batch 12).  G10: 20-step loss curves of the reference Learner.fit (2 epochs x 10
    minibatches) on seeded synthetic data objects — CollabFilterNet (n_user 30, n_item 20, D 6; Adam, wd 1e-4) and
    StructuredDataNet (cards [7,5,4], 3 continuous, fc [16,8,1], dropout 0; Adam, wd 1e-3) — plus the pre-training and final
    validation losses."""
    L = R['General.Layers']
    CF = R['Applications.CollabFiltering']
    SD = R['Applications.StructuredData']
    out = {}
    # ---- G4
    net = L.FullyConnectedNet([20, 16, 8, 3], [0.3, 0.2, 0.1])
    synth.fill_module_(net, seed=14)
    masks = [_mask((12, 20), 0.3, 141), _mask((12, 16), 0.2, 142), _mask((12, 8), 0.1, 143)]
    net.lins[0].drop, net.lins[1].drop, net.final_drop = _FixedDrop([masks[0]]), _FixedDrop([masks[1]]), _FixedDrop([masks[2]])
    net.train()
    x = synth.synth_input((12, 20), 144).requires_grad_(True)
    y = torch.arange(12) % 3
    logits = net(x)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    out.update({'g4.mask0': A(masks[0]), 'g4.mask1': A(masks[1]), 'g4.mask2': A(masks[2]), 'g4.logits': A(logits), 'g4.loss': A(loss),
                'g4.dx': A(x.grad), 'g4.param_names': np.array([n for n, _ in net.named_parameters()])})
    for n, p in net.named_parameters():
        out['g4.grad.' + n] = A(p.grad)
    for n, b in net.named_buffers():
        out['g4.buf.' + n] = A(b)
    # ---- G10 collab
    rs = np.random.RandomState(100)
    def collab_batches(n):
        return [(torch.from_numpy(np.stack([rs.randint(0, 30, 16), rs.randint(0, 20, 16)], 1)),
                 torch.from_numpy(rs.randint(1, 6, 16).astype(np.float32))) for _ in range(n)]
    tr, va = collab_batches(10), collab_batches(3)
    net = CF.CollabFilterNet(30, 20, 6, [0.8, 5.2])
    synth.fill_module_(net, seed=15)
    learner = Learner('/tmp/nnl_golden_g10', FakeData(tr, va, 16, 'cont'), net, optimizer='Adam')
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        pre = learner.evaluate('val')[0]
        learner.fit(2e-2, 2, wd=1e-4)
        post = learner.evaluate('val')[0]
    out.update({'g10.collab.loss_sched': np.array(learner.loss_sched, dtype=np.float64), 'g10.collab.val_pre': np.array([pre]),
                'g10.collab.val_post': np.array([post])})
    for i, (xb, yb) in enumerate(tr + va):
        out['g10.collab.x%d' % i], out['g10.collab.y%d' % i] = A(xb), A(yb)
    # ---- G10 tabular
    cards = [7, 5, 4]
    def tab_batches(n):
        return [([torch.from_numpy(np.stack([rs.randint(0, c, 16) for c in cards], 1)),
                  torch.from_numpy(rs.standard_normal((16, 3)).astype(np.float32))],
                 torch.from_numpy((5 + 7 * rs.rand(16)).astype(np.float32))) for _ in range(n)]
    tr, va = tab_batches(10), tab_batches(3)
    labels = [{i: i for i in range(c)} for c in cards]
    net = SD.StructuredDataNet('cont', 3, 3, labels, [16, 8, 1], output_range=[5, 12])
    synth.fill_module_(net, seed=16)
    learner = Learner('/tmp/nnl_golden_g10', FakeData(tr, va, 16, 'cont'), net, optimizer='Adam')
    with contextlib.redirect_stdout(io.StringIO()):
        pre = learner.evaluate('val')[0]
        learner.fit([1e-2, 2e-2], 2, wd=1e-3)
        post = learner.evaluate('val')[0]
    out.update({'g10.tab.loss_sched': np.array(learner.loss_sched, dtype=np.float64), 'g10.tab.val_pre': np.array([pre]),
                'g10.tab.val_post': np.array([post]), 'g10.tab.emb_dims': np.array([e.emb.weight.shape[1] for e in net.embeddings])})
    for i, ((xc, xf), yb) in enumerate(tr + va):
        out['g10.tab.xcat%d' % i], out['g10.tab.xcont%d' % i], out['g10.tab.y%d' % i] = A(xc), A(xf), A(yb)
    save('g4_g10_fcnet_fit', **out)


GROUPS = {'g4': g4_fcnet_and_g10_fit_curves, 'g12': g12_objectdetectionnet, 'g11': g11_bbox_inference, 'g1': g1_collab, 'g9': g9_host_logic, 'g7': g7_text, 'g8': g8_detection, 'g3': g3_tabular, 'g5': g5_blocks, 'g6': g6_resnet34, 'g13': g13_resnet34_curve}

if __name__ == '__main__':
    names = sys.argv[1:] or sorted(GROUPS)
    for n in names:
        GROUPS[n]()
