"""Fused RetinaNet loss kernels alone at BASELINE size (16 images x 49 104 anchors x 20 classes, <= 8 objects): forward
(one launch) and backward (one launch) time against the 224 B/anchor algorithmic traffic of SURVEY.md §8d."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuralnetworklibrary_amd import ops  # noqa: E402
from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import AnchorGenerator  # noqa: E402

dev = torch.device('cuda', 0)
bs, K = 16, 20
anchors = AnchorGenerator()(torch.zeros(bs, 3, 512, 512, device=dev))
A = len(anchors)
g = torch.Generator(device=dev).manual_seed(0)
reg = (torch.randn(bs, A, 4, device=dev, generator=g) * 0.3).requires_grad_(True)
clas = torch.sigmoid(torch.randn(bs, A, K, device=dev, generator=g) - 4.0).requires_grad_(True)
boxes, cats = bench._retina_targets(np.random.RandomState(0), bs)
boxes, cats = torch.from_numpy(boxes).to(dev), torch.from_numpy(cats).to(dev)


def run(n):
    ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        reg.grad = clas.grad = None
        ef[0].record()
        out = ops.retina_loss(anchors, reg, clas, boxes, cats)
        ef[1].record()
        out[0].backward()
        ef[2].record()
        torch.cuda.synchronize()
        tf += ef[0].elapsed_time(ef[1]); tb += ef[1].elapsed_time(ef[2])
    return tf / n, tb / n, out


run(3)
tf, tb, out = run(20)
alg = 224.0 * bs * A
print(json.dumps({'A': A, 'fwd_us': round(tf * 1e3, 1), 'bwd_us_incl_autograd': round(tb * 1e3, 1), 'loss': out.tolist(),
                  'algorithmic_MB': round(alg / 1e6, 1), 'TBps_fwd_plus_bwd': round(alg / ((tf + tb) * 1e-3) / 1e12, 2)}))
