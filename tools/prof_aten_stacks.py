"""Which Python lines launch the ATen glue kernels (fill / copy / add / cat) of one training step?  torch.profiler with stacks, grouped by
(op, innermost frame inside this repo).  Usage: python tools/prof_aten_stacks.py lm|retinanet|resnet [bs]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda', 0)
from neuralnetworklibrary_amd.General.Core import set_default_device  # noqa: E402
set_default_device(dev)
which = sys.argv[1] if len(sys.argv) > 1 else 'lm'
if which == 'lm':
    wl = bench.lm_workload(dev, 64, 1237, 1)
elif which == 'retinanet':
    wl = bench.retina_workload(dev, 16, 1238, 1)
else:
    wl = bench.resnet34_workload(dev, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 1235, 1)
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    wl.step(0)
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if not e.name.startswith('aten::') or e.name in ('aten::empty', 'aten::empty_like', 'aten::view', 'aten::as_strided', 'aten::permute', 'aten::slice', 'aten::select',
                                                        'aten::reshape', 'aten::detach', 'aten::alias', 'aten::_unsafe_view', 'aten::empty_strided', 'aten::transpose',
                                                        'aten::expand', 'aten::unsqueeze', 'aten::squeeze', 'aten::narrow', 't', 'aten::t', 'aten::result_type',
                                                        'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::flatten', 'aten::unbind', 'aten::item', 'aten::is_nonzero',
                                                        'aten::_local_scalar_dense', 'aten::lift_fresh', 'aten::resolve_conj', 'aten::resolve_neg', 'aten::zeros', 'aten::ones',
                                                        'aten::zeros_like', 'aten::new_empty', 'aten::new_zeros', 'aten::clone', 'aten::float', 'aten::view_as'):
        continue
    frame = '?'
    for f in (e.stack or []):
        if ROOT in f and 'prof_aten_stacks' not in f:
            frame = f.replace(ROOT + '/', '')
            break
    if e.stack is None or frame == '?':
        frame = 'autograd engine / no repo frame'
    agg[(e.name, frame[:110])] += 1
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1])[:45]:
    print('x%-4d %-26s %s' % (n, name, frame))
