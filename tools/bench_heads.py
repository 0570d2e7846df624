"""Throughput of Learner.train1minibatch for the other BASELINE configs, 1 GPU, synthetic data:
  1 collab   ML-100K CollabFilterNet bs=64 (and bs=8192, the MovieLens-20M notebook batch)      samples/s
  3 tabular  Rossmann-shape StructuredDataNet bs=1024, fc [1000,500,1], Adam                      samples/s
  4 lm       AWD-LSTM LanguageModelNet 400/1150/3, V=47343, bs=64 bptt=70, Adam, RegSeqCE(2,1)    tokens/s
  5 retina   ObjectDetectionNet(20) R50-FPN 512x512 bs=16, SSD_loss, SGD momentum                 images/s
Usage: python tools/bench_heads.py [collab tabular lm retina] [--steps 10] [--graphs]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import _lib  # noqa: E402
from neuralnetworklibrary_amd.General.Learner import Learner  # noqa: E402

DEV = 'cuda'
Learner.verbose = False


class Data:
    def __init__(self, batches, bs, target_type, **kw):
        self.train_dl, self.val_dl, self.bs, self.target_type = batches, batches[:1], bs, target_type
        self.__dict__.update(kw)


GRAPHS = False


def run(name, learner, batches, lr, unit, units_per_step, steps, warmup=3, **kw):
    learner.model.train()
    if GRAPHS and 'lm' not in name and 'retina' not in name:
        learner.use_graphs(True)
        name += ' [hipGraph step]'
        warmup += 3
    for i in range(warmup):
        learner.train1minibatch(*batches[i % len(batches)], lr, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = learner.train1minibatch(*batches[i % len(batches)], lr, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _lib.prof_enable(True)
    for i in range(3):
        learner.train1minibatch(*batches[i % len(batches)], lr, **kw)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    prof = {k: round(v['ms'] / 3, 3) for k, v in _lib.prof_collect().items() if v['launches']}
    print(json.dumps({'config': name, 'ms_per_step': round(dt * 1e3, 3), 'value': round(units_per_step / dt, 1), 'unit': unit,
                      'last_loss': loss, 'hip_ms_per_step_by_kind': prof}))


def collab(steps, bs):
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    g = torch.Generator(device=DEV).manual_seed(1234)
    batches = [(torch.stack([torch.randint(0, 943, (bs,), device=DEV, generator=g), torch.randint(0, 1682, (bs,), device=DEV, generator=g)], 1),
                torch.randint(1, 6, (bs,), device=DEV, generator=g).float()) for _ in range(4)]
    net = CollabFilterNet(943, 1682, 30, [0.8, 5.2])
    learner = Learner('/tmp/nnl_bh', Data(batches, bs, 'cont'), net, optimizer='Adam')
    learner.init_optimizer(wd=1e-4)
    run('collab ML-100K bs=%d' % bs, learner, batches, 1e-2, 'samples/s', bs, steps)


def tabular(steps):
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    cards = [1116, 5, 4, 13, 53, 13, 4, 8, 32, 23, 27, 24, 28, 9, 5, 5] + [10] * 16
    bs, n_cont = 1024, 14
    rs = np.random.RandomState(1236)
    batches = []
    for _ in range(4):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=bs) for c in cards], 1).astype(np.int64)).to(DEV)
        xcont = torch.from_numpy(rs.standard_normal((bs, n_cont)).astype(np.float32)).to(DEV)
        y = torch.from_numpy((5 + 7 * rs.rand(bs)).astype(np.float32)).to(DEV)
        batches.append(([xcat, xcont], y))
    net = StructuredDataNet('cont', 32, n_cont, [{i: i for i in range(c)} for c in cards], [1000, 500, 1], output_range=[5, 12],
                            dropout_levels=(0.04, 0.04, [0, 0.5, 0.25]))
    learner = Learner('/tmp/nnl_bh', Data(batches, bs, 'cont'), net, optimizer='Adam')
    learner.init_optimizer(wd=1e-3)
    run('tabular Rossmann-shape bs=1024', learner, batches, [1e-3, 1e-3], 'samples/s', bs, steps)


def lm(steps):
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, RegSeqCrossEntropyLoss, _Vocab
    V, bs, bptt = 47343, 64, 70
    stoi = {i: i for i in range(V)}
    stoi['_pad_'] = 1
    del stoi[1]
    d = _Vocab(stoi, bs)
    g = torch.Generator(device=DEV).manual_seed(1237)
    stream_ = torch.randint(4, V, (bs, bptt * 4 + 1), device=DEV, generator=g)
    batches = [(stream_[:, i * bptt:(i + 1) * bptt].contiguous(), stream_[:, i * bptt + 1:(i + 1) * bptt + 1].contiguous()) for i in range(4)]
    net = LanguageModelNet(d)
    learner = Learner('/tmp/nnl_bh', Data(batches, bs, 'lang_model'), net, optimizer='Adam', loss_func=RegSeqCrossEntropyLoss(2.0, 1.0))
    learner.init_optimizer(wd=1e-6, clip=0.4)
    run('AWD-LSTM LM bs=64 bptt=70 V=47343', learner, batches, [1e-3, 1e-3], 'tokens/s', bs * bptt, steps, betas_batch=(0.8, 0.99))


def retina(steps):
    from neuralnetworklibrary_amd.Applications.Vision import ObjectDetectionNet, SSD_loss
    bs, M = 16, 8
    rs = np.random.RandomState(1238)
    batches = []
    for _ in range(2):
        x = torch.randn(bs, 3, 512, 512, device=DEV)
        boxes = -np.ones((bs, M, 4), np.float32); cats = -np.ones((bs, M), np.int64)
        for i in range(bs):
            m = rs.randint(1, M + 1)
            xy = rs.uniform(0, 300, (m, 2)); wh = rs.uniform(30, 210, (m, 2))
            boxes[i, :m] = np.concatenate([xy, xy + wh], 1); cats[i, :m] = rs.randint(0, 20, m)
        batches.append((x, [torch.from_numpy(boxes).to(DEV), torch.from_numpy(cats).to(DEV)]))
    torch.manual_seed(1238)
    net = ObjectDetectionNet(20)
    learner = Learner('/tmp/nnl_bh', Data(batches, bs, 'bbox'), net, optimizer='SGD_Mom', loss_func=SSD_loss(0.5, 0.25, 2.0))
    learner.init_optimizer(wd=1e-4)
    run('RetinaNet R50-FPN 512x512 bs=16 K=20', learner, batches, [1e-4, 1e-3, 1e-3], 'images/s', bs, steps)


def bbox_infer(steps):
    """BBoxPredictor post-processing alone (decode + top_k sort + NMS + host pruning) on RetinaNet-512 sized activations."""
    from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import AnchorGenerator, BBoxPredictor
    bs, K = 16, 20
    img = torch.zeros(bs, 3, 512, 512, device=DEV)
    anchors = AnchorGenerator()(img)
    A = len(anchors)
    g = torch.Generator(device=DEV).manual_seed(3)
    reg = torch.randn(bs, A, 4, device=DEV, generator=g) * 0.5
    clas = torch.sigmoid(torch.randn(bs, A, K, device=DEV, generator=g) * 1.2 - 4.0)     # ~2 % of the anchors above 0.05
    pred = BBoxPredictor()
    for _ in range(2):
        out = pred(img, reg, clas, anchors)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = pred(img, reg, clas, anchors)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({'config': 'BBoxPredictor bs=16, 49104 anchors x 20 classes, top_k 1000', 'ms_per_batch': round(dt * 1e3, 3),
                      'value': round(bs / dt, 1), 'unit': 'images/s', 'boxes_img0': len(out[0][0])}))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='*', default=['collab', 'tabular', 'lm', 'retina'])
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--graphs', action='store_true', help='Learner.use_graphs(): replay the captured step (collab / tabular)')
    a = ap.parse_args()
    GRAPHS = a.graphs
    from neuralnetworklibrary_amd.General.Core import set_default_device
    set_default_device(DEV)
    for w in a.which:
        if w == 'collab':
            collab(max(a.steps, 50), 64); collab(max(a.steps, 50), 8192)
        elif w == 'tabular':
            tabular(max(a.steps, 30))
        elif w == 'lm':
            lm(a.steps)
        elif w == 'retina':
            retina(a.steps)
        elif w == 'bbox':
            bbox_infer(max(a.steps, 10))
