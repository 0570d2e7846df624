"""ORACLE — TEST INFRASTRUCTURE ONLY.  Loss-CURVE golden vectors at BASELINE sizes (build container only).

Runs the REAL reference's `Learner.train1minibatch` (imported read-only from /root/reference through oracle/_ref_import.py) for
many consecutive steps on WELL-CONDITIONED fixtures — the reference constructors' own init distributions in closed form
(synth.fill_reference_init_), learning rates <= 1e-3 per layer group, a DISTINCT synthetic minibatch every step, dropout 0 —
in fp32 and, with the same modules cast to fp64, in fp64, and writes per-step losses (data only) to tests/golden/.  The fixture
is accepted only if the reference's own fp32-vs-fp64 separation stays below 3e-4 relative on every step (asserted here, so that
the GPU test can demand |hip - ref32| <= 1e-3 |ref32| on EVERY step, north_star's loss-curve tolerance).
Usage:  python oracle/gen_golden_curves.py [g13b] [g14] [g15] [g16]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import R, Learner, save, A  # noqa: E402  (imports the reference)
import synth  # noqa: E402

SEP_LIMIT = 3e-4


def _check_sep(l32, l64, what):
    sep = np.abs(l32 - l64) / np.abs(l64)
    print(what, 'fp32-vs-fp64 relative separation per step:', np.array2string(sep, precision=1), flush=True)
    assert sep.max() < SEP_LIMIT, '%s: the reference itself separates by %.1e (>= %.0e): fixture not well conditioned' % (what, sep.max(), SEP_LIMIT)


def g13b_resnet34_curve(steps=20):
    """G13b: 20 consecutive `Learner.train1minibatch` steps of the REFERENCE at BASELINE configs[1]'s own size — ResNet-34 body
    (retinanet.py:30-59,299-356 BasicBlock [3,4,6,3]) + default head (Vision.py:1311-1317, dropout 0), 224x224, bs 64, SGD
    momentum 0.9, wd 1e-4, BatchNorm in training mode, 20 DISTINCT learnable batches (synth.curve_batch_images tags 1300+i).

    Conditioning (measured here, /tmp probes of this same script's protocol): every fp32 gradient of this network carries ~6e-3 of
    gate-flip noise, so the per-step loss noise is ~6e-3 x (the loss change that step's update causes).  At the freshly
    initialised network the layer groups' gradient norms are 279 (stem) / 119 / 96 | 64 / 32 | 7 (head), i.e. lr G^2 — the
    first-order loss change per step — is 8 for the stem at lr 1e-4: the round-2 fixture (and lr [1e-4, 3e-4, 1e-3]) separates
    x20 per step in the reference's own fp32 / fp64 runs.  lr [1e-7, 1e-6, 1e-4] balances the groups (lr G^2 ~ 1e-2 each): the
    curve descends 1.0 -> ~0.5 in 20 steps while fp32 and fp64 stay within 3e-4 (asserted below).  `losses.f32.headonly` is the
    same run with the two body groups' lr set to 0: its distance from `losses.f32` is how much of the curve is due to the
    convolution weight gradients (the test prints it: the curve is informative about them well above the 1e-3 tolerance)."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    N, S = 64, 224
    lr = [1e-7, 1e-6, 1e-4]

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'

    out = {'N': N, 'S': S, 'steps': steps, 'lr': np.array(lr), 'wd': 1e-4, 'init_seed': 13}
    for tag, dtype, lrs in [('f32', torch.float32, lr), ('f32.headonly', torch.float32, [0., 0., lr[2]]), ('f64', torch.float64, lr)]:
        arch = RN.RetinaNet(2, RN.BasicBlock, [3, 4, 6, 3])
        net = V.ImageClassificationNet(D, arch, head=[[512], [0., 0.]], cutpoint=8, splits=[6])
        synth.fill_reference_init_(net, seed=13)
        net = net.to(dtype).train()
        d = D(); d.train_dl = [(None, torch.zeros(N))]; d.val_dl = d.train_dl
        learner = Learner('/tmp/nnl_golden_g13b', d, net, optimizer='SGD_Mom')
        learner.init_optimizer(wd=1e-4)
        losses = []
        for i in range(steps):
            x, y = synth.curve_batch_images(N, S, 1300 + i)
            losses.append(learner.train1minibatch(x.to(dtype), y, lrs))
            print(tag, i, losses[-1], flush=True)
        out['losses.' + tag] = np.array(losses, dtype=np.float64)
        out['after.abs_sums.' + tag] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
    print('body share of the curve: max rel |headonly - f32| = %.2e' % (np.abs(out['losses.f32.headonly'] - out['losses.f32']) / out['losses.f32']).max())
    _check_sep(out['losses.f32'], out['losses.f64'], 'g13b')
    save('g13b_resnet34_curve', **out)


def g14_lm_curve(steps=20):
    """G14: 20 consecutive `Learner.train1minibatch` steps of the REFERENCE at BASELINE configs[3]'s own size — LanguageModelNet
    400 / 1150 / 3 layers (Text.py:611-702), V = 47 343, bs 64, bptt 70, every dropout 0, Adam betas (0.8, 0.99)
    (IMDB.ipynb cell 14), lr [5e-4, 1e-3] per layer group, wd 1e-6, RegSeqCrossEntropyLoss(2, 1); 20 CONSECUTIVE windows of one
    token stream (synth.lm_stream tag 14), hidden state carried from batch to batch (Text.py:547-550)."""
    TX = R['Applications.Text']
    V, bs, bptt = 47343, 64, 70
    lr = [5e-4, 1e-3]

    class D:
        pass
    stream = synth.lm_stream(V, bs, steps * bptt + 1, 14)
    out = {'V': V, 'bs': bs, 'bptt': bptt, 'steps': steps, 'lr': np.array(lr), 'wd': 1e-6, 'init_seed': 14}
    for tag, dtype in [('f32', torch.float32), ('f64', torch.float64)]:
        d = D(); d.stoi = {i: i for i in range(V)}; d.stoi['_pad_'] = 1; del d.stoi[1]; d.bs = bs; d.target_type = 'lang_model'
        net = TX.LanguageModelNet(d, enc_drops=[0., 0., 0., 0.], dec_drop=0.)
        net.clear_non_raw()
        synth.fill_lm_reference_init_(net, seed=14)
        net = net.to(dtype)
        net.enc.h, net.enc.c = [t.to(dtype) for t in net.enc.h], [t.to(dtype) for t in net.enc.c]
        d.train_dl = [(None, torch.zeros(bs))]; d.val_dl = d.train_dl
        from functools import partial
        opt = R['General.Optimizer'].Optimizer(partial(torch.optim.Adam, betas=(0.8, 0.99)), net)
        learner = Learner('/tmp/nnl_golden_g14', d, net, opt, TX.RegSeqCrossEntropyLoss(2.0, 1.0))
        learner.init_optimizer(wd=1e-6)
        net.train()
        losses = []
        for i in range(steps):
            xb = torch.from_numpy(stream[:, i * bptt:(i + 1) * bptt].copy())
            yb = torch.from_numpy(stream[:, i * bptt + 1:(i + 1) * bptt + 1].copy())
            losses.append(learner.train1minibatch(xb, yb, lr, betas_batch=(0.8, 0.99)))
            print(tag, i, losses[-1], flush=True)
        out['losses.' + tag] = np.array(losses, dtype=np.float64)
        out['after.abs_sums.' + tag] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
    _check_sep(out['losses.f32'], out['losses.f64'], 'g14')
    save('g14_lm_curve', **out)


def g15_retinanet_bs16(steps=10):
    """G15: the REFERENCE's assembled ObjectDetectionNet(20) + SSD_loss(0.5, 0.25, 2) at BASELINE configs[4]'s own size — 512 x 512,
    bs 16, 49 104 anchors (Vision.py:1446-1471, 1620-1644), BatchNorm in training mode, in fp32 and fp64:
      (a) ONE forward + backward on the well-conditioned seeded weights of G12 (synth.fill_detection_net_, seed 15: head output
          convolutions non-zero so every tower gradient is exercised): loss, reg / clas activation checksums per pyramid level and
          strided samples, every parameter's gradient norm, 1024-element gradient slices of 9 tensors;
      (b) a 10-step `Learner.train1minibatch` curve (SGD momentum 0.9, lr [1e-4, 3e-4, 1e-3], wd 1e-4, 10 distinct batches) from the
          same weights.
    The COCO checkpoint the constructor loads is an LFS pointer (retinanet.py:430-435): replaced by the same architecture."""
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    RN.retinanet = lambda *a, **k: RN.RetinaNet(80, RN.Bottleneck, [3, 4, 6, 3])
    N, S, K, M = 16, 512, 20, 8
    lr = [1e-4, 3e-4, 1e-3]
    out = {'N': N, 'S': S, 'K': K, 'M': M, 'steps': steps, 'lr': np.array(lr), 'wd': 1e-4, 'init_seed': 15}
    slices = ['layer0.0.weight', 'layer1.0.conv1.weight', 'layer2.0.conv2.weight', 'layer3.5.conv3.weight', 'layer4.2.conv2.weight',
              'fpn.P5_1.weight', 'fpn.P3_2.weight', 'classifier.conv1.weight', 'classifier.output.bias', 'regressor.conv4.weight',
              'regressor.output.weight']
    level_sizes = [(S // 2 ** l) ** 2 * 9 for l in range(3, 8)]

    class D:
        bs, target_type = N, 'bbox'

    for tag, dtype in [('f32', torch.float32), ('f64', torch.float64)]:
        torch.manual_seed(0)
        net = V.ObjectDetectionNet(K)
        synth.fill_detection_net_(net, seed=15)
        net = net.to(dtype).train()
        sd = dict(net.named_parameters())
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
            out['slice_names'] = np.array([n for n in slices if n in sd])
        # (a) one forward + backward
        x = synth.synth_input((N, 3, S, S), 1500).to(dtype)
        boxes, cats = synth.detection_targets(N, M, S, K, 1500)
        anchors, reg, clas = net(x)
        lf = V.SSD_loss(0.5, 0.25, 2.0)
        loss = lf([anchors, reg, clas], [torch.from_numpy(boxes).to(dtype), torch.from_numpy(cats)])
        loss.backward()
        out['a.loss.' + tag] = np.array([loss.item(), float(lf.reg_loss), float(lf.clas_loss)], dtype=np.float64)
        r, c = reg.detach().double(), clas.detach().double()
        o = 0
        lv = []
        for n_l in level_sizes:
            lv.append([r[:, o:o + n_l].sum().item(), r[:, o:o + n_l].abs().sum().item(), c[:, o:o + n_l].sum().item(), (c[:, o:o + n_l] ** 2).sum().item()])
            o += n_l
        assert o == reg.shape[1] == 49104
        out['a.level_sums.' + tag] = np.array(lv, dtype=np.float64)
        out['a.reg_sample.' + tag] = A(r.reshape(-1)[::397]).astype(np.float64)
        out['a.clas_sample.' + tag] = A(c.reshape(-1)[::1987]).astype(np.float64)
        out['a.grad_norms.' + tag] = np.array([0.0 if p.grad is None else p.grad.double().norm().item() for _, p in net.named_parameters()], dtype=np.float64)
        for n in slices:
            if n in sd and sd[n].grad is not None:
                out['a.grad.%s.%s' % (n, tag)] = A(sd[n].grad).reshape(-1)[:1024].astype(np.float64)
        print(tag, 'a', out['a.loss.' + tag], flush=True)
        del anchors, reg, clas, loss, r, c
        # (b) the curve, from the same weights
        d = D(); d.train_dl = [(None, [torch.zeros(N)])]; d.val_dl = d.train_dl
        learner = Learner('/tmp/nnl_golden_g15', d, net, optimizer='SGD_Mom', loss_func=V.SSD_loss(0.5, 0.25, 2.0))
        learner.init_optimizer(wd=1e-4)
        losses = []
        for i in range(steps):
            x = synth.synth_input((N, 3, S, S), 1501 + i).to(dtype)
            boxes, cats = synth.detection_targets(N, M, S, K, 1501 + i)
            losses.append(learner.train1minibatch(x, [torch.from_numpy(boxes).to(dtype), torch.from_numpy(cats)], lr))
            print(tag, i, losses[-1], flush=True)
        out['losses.' + tag] = np.array(losses, dtype=np.float64)
        out['after.abs_sums.' + tag] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        del net, learner
    _check_sep(out['losses.f32'], out['losses.f64'], 'g15')
    save('g15_retinanet_bs16', **out)


GROUPS = {'g13b': g13b_resnet34_curve, 'g14': g14_lm_curve, 'g15': g15_retinanet_bs16}

if __name__ == '__main__':
    for name in ([a for a in sys.argv[1:] if a in GROUPS] if sys.argv[1:] else list(GROUPS)):
        GROUPS[name]()


def _g13c_net(dtype):
    RN = R['Applications.VisionModels.retinanet']
    V = R['Applications.Vision']
    N, S = 64, 224

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
    arch = RN.RetinaNet(2, RN.BasicBlock, [3, 4, 6, 3])
    net = V.ImageClassificationNet(D, arch, head=[[512], [0., 0.]], cutpoint=8, splits=[6])
    synth.fill_reference_init_(net, seed=13)
    synth.tame_residual_branches_(net)                          # frozen BatchNorm = no renormalisation: keep the 16 residual sums O(1)
    net = net.to(dtype)
    d = D(); d.train_dl = [(None, torch.zeros(N))]; d.val_dl = d.train_dl
    learner = Learner('/tmp/nnl_golden_g13c', d, net, optimizer='SGD_Mom')
    learner.bn_freeze('all')                                   # Learner.py:248-264: BatchNorm parameters out of the optimizer ...
    learner.init_optimizer(wd=1e-4)
    net.train()
    for m in net.modules():                                    # ... and, as train_gen_sched does (Learner.py:589-591), running statistics in the forward
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.training = False
    return net, learner, N, S


def g13c_probe():
    "gradient norms per layer group of the frozen-BatchNorm network at its initial weights (chooses G13c's learning rates)"
    net, learner, N, S = _g13c_net(torch.float32)
    x, y = synth.curve_batch_images(N, S, 1400)
    loss = torch.nn.functional.cross_entropy(net(x), y)
    loss.backward()
    print('loss', loss.item())
    for gi, grp in enumerate(net.layer_groups):
        g2 = sum(float(p.grad.double().pow(2).sum()) for p in grp.parameters() if p.grad is not None)
        print('group', gi, 'grad norm', g2 ** 0.5, flush=True)


def g13c_resnet34_frozen_bn_curve(steps=20):
    """G13c (VERDICT r3 next #4a): a 20-step curve at BASELINE configs[1]'s size that is SENSITIVE to the convolution weight gradients —
    ResNet-34 + default head, 224x224, bs 64, SGD momentum 0.9, wd 1e-4, `Learner.bn_freeze('all')` with the BatchNorm layers on their
    running statistics as `train_gen_sched` runs them (Learner.py:248-264, 589-591: no batch-statistics coupling, well conditioned —
    DESIGN.md section 4, conditioning note), 20 distinct learnable batches (synth.curve_batch_images tags 1400+i).  The body learning
    rates are chosen so that the reference's own HEAD-ONLY run (body lr 0) leaves the full curve by >= 20 % while its fp32 and fp64 runs
    stay within 3e-4 of each other on every step — both asserted here; a ~5 % error in every body weight gradient moves this curve by
    ~1e-2, ten times the GPU test's tolerance (G13b moves by 6e-4 for the same error)."""
    lr = [float(v) for v in os.environ.get('G13C_LR', '1e-4,2e-5,1.2e-5').split(',')]
    out = {'N': 64, 'S': 224, 'steps': steps, 'lr': np.array(lr), 'wd': 1e-4, 'init_seed': 13}
    for tag, dtype, lrs in [('f32', torch.float32, lr), ('f32.headonly', torch.float32, [0., 0., lr[2]]), ('f64', torch.float64, lr)]:
        net, learner, N, S = _g13c_net(dtype)
        losses = []
        for i in range(steps):
            x, y = synth.curve_batch_images(N, S, 1400 + i)
            losses.append(learner.train1minibatch(x.to(dtype), y, lrs))
            print(tag, i, losses[-1], flush=True)
        out['losses.' + tag] = np.array(losses, dtype=np.float64)
        out['after.abs_sums.' + tag] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
    share = (np.abs(out['losses.f32.headonly'] - out['losses.f32']) / out['losses.f32']).max()
    print('body share of the curve: max rel |headonly - f32| = %.2e' % share)
    assert share >= 0.2, 'the curve is not sensitive enough to the body gradients'
    _check_sep(out['losses.f32'], out['losses.f64'], 'g13c')
    save('g13c_resnet34_frozen_bn_curve', **out)


GROUPS.update({'g13c': g13c_resnet34_frozen_bn_curve, 'g13c_probe': g13c_probe})

if __name__ == '__main__' and 'GROUPS_LATE' not in globals():
    GROUPS_LATE = True
    for name in [a for a in sys.argv[1:] if a in ('g13c', 'g13c_probe')]:
        GROUPS[name]()


ROSSMANN_CARDS, rossmann_batch = synth.ROSSMANN_CARDS, synth.rossmann_batch


def g16_rossmann_curve(steps=20):
    """G16 (VERDICT r3 next #4b): 20 consecutive `Learner.train1minibatch` steps of the REFERENCE at BASELINE configs[2]'s real shape —
    StructuredDataNet (StructuredData.py:1038-1084) with the 32 Rossmann cardinalities of SURVEY 8d (default embedding sizes -> 189 + 14
    = 203 inputs), fc [1000, 500, 1], output range [5, 12], bs 1024, Adam, lr [5e-4, 5e-4], wd 1e-3, every dropout 0, BatchNorm1d in
    training mode, nn.Embedding(max_norm = 1.5) renormalising the looked-up rows in place on every forward; 20 distinct learnable batches
    (rossmann_batch).  fp32 and fp64 runs of the reference must stay within 3e-4 (asserted)."""
    SD = R['Applications.StructuredData']
    bs, n_cont = 1024, 14
    lr = [5e-4, 5e-4]                # (at 1e-3 the reference's own fp32 / fp64 runs separate by 4.1e-4 at step 19)
    labels = [{i: i for i in range(c)} for c in ROSSMANN_CARDS]
    out = {'bs': bs, 'n_cont': n_cont, 'steps': steps, 'lr': np.array(lr), 'wd': 1e-3, 'init_seed': 16, 'cards': np.array(ROSSMANN_CARDS)}
    for tag, dtype in [('f32', torch.float32), ('f64', torch.float64)]:
        net = SD.StructuredDataNet('cont', len(ROSSMANN_CARDS), n_cont, labels, [1000, 500, 1], output_range=[5, 12])
        synth.fill_module_(net, seed=16)
        net = net.to(dtype)
        data = FakeData([(None, torch.zeros(bs))], [(None, torch.zeros(bs))], bs, 'cont')
        learner = Learner('/tmp/nnl_golden_g16', data, net, optimizer='Adam')
        learner.init_optimizer(wd=1e-3)
        net.train()
        losses = []
        for i in range(steps):
            xcat, xcont, y = rossmann_batch(bs, n_cont, i)
            losses.append(learner.train1minibatch([torch.from_numpy(xcat), torch.from_numpy(xcont).to(dtype)], torch.from_numpy(y).to(dtype), lr))
        print(tag, np.array2string(np.array(losses), precision=5), flush=True)
        out['losses.' + tag] = np.array(losses, dtype=np.float64)
        out['after.abs_sums.' + tag] = np.array([p.double().abs().sum().item() for _, p in net.named_parameters()], dtype=np.float64)
        if tag == 'f32':
            out['param_names'] = np.array([n for n, _ in net.named_parameters()])
            out['emb_dims'] = np.array([e.emb.weight.shape[1] for e in net.embeddings])
            net.eval()
            xcat, xcont, y = rossmann_batch(bs, n_cont, 0)
            out['eval_pred0.f32'] = A(net(torch.from_numpy(xcat), torch.from_numpy(xcont)))[:64]
    _check_sep(out['losses.f32'], out['losses.f64'], 'g16')
    save('g16_rossmann_curve', **out)


GROUPS['g16'] = g16_rossmann_curve
from gen_golden import FakeData  # noqa: E402

if __name__ == '__main__' and 'g16' in sys.argv[1:]:
    g16_rossmann_curve()
