"""Strong-scaling compute ceiling on one GPU: the ResNet-34 headline step at 64/N images (N = 8, 4, 2, 1), eager and as a
replayed whole-step hipGraph (Learner.use_graphs), with the HIP-event kernel time of the eager step beside the wall time.
Usage: python tools/bench_small_batch.py [--bs 8,16,32,64] [--steps 30] [--no-graph]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--bs', default='8,16,32,64')
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--no-graph', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda', 0)
from neuralnetworklibrary_amd.General.Core import set_default_device  # noqa: E402
set_default_device(dev)
clock = bench.Clock(None, dev)
for bs in [int(v) for v in a.bs.split(',')]:
    wl = bench.resnet34_workload(dev, bs, 1235, 1)
    eager = clock.timed(wl.step, 5, a.steps) / a.steps * 1e3
    prof = bench.profile_kinds(wl, 5)
    kern = sum(v['ms'] for v in prof.values()) / 5
    launches = sum(v['launches'] for v in prof.values()) / 5
    out = {'bs': bs, 'eager_ms': round(eager, 3), 'nnl_kernel_ms': round(kern, 3), 'nnl_launches': launches,
           'by_kind': bench.by_kind(prof, 5)}
    if not a.no_graph:
        wl.learner.use_graphs(True)
        out['graph_ms'] = round(clock.timed(wl.step, 8, a.steps) / a.steps * 1e3, 3)
    print(json.dumps(out), flush=True)
    del wl
    torch.cuda.empty_cache()
