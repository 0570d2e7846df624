"""ResNet-34 headline step with and without Learner.use_graphs() (whole-step hipGraph replay), same process.
(bench.py's strong_scaling_proxy reports the same pair at 8 / 16 / 32 images.)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuralnetworklibrary_amd.General.Core import set_default_device  # noqa: E402

dev = torch.device('cuda', 0)
set_default_device(dev)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = bench.resnet34_workload(dev, bs, 1234, 1)
clock = bench.Clock(None, dev)
eager = clock.timed(wl.step, 5, 20) / 20 * 1e3
l0 = wl.loss
wl.learner.use_graphs(True)
graph = clock.timed(wl.step, 8, 20) / 20 * 1e3
print(json.dumps({'bs': bs, 'eager_ms': round(eager, 3), 'graph_ms': round(graph, 3), 'loss_eager': l0, 'loss_graph': wl.loss,
                  'graphs': sum(g.graph is not None for g in wl.learner._graphs.values())}))
