import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
from neuralnetworklibrary_amd.General.Core import set_default_device
set_default_device(dev)
wl = bench.retina_workload(dev, 16, 1238, 1)
for i in range(3): wl.step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step(0)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    ct = getattr(e, 'device_time_total', None) or getattr(e, 'cuda_time_total', 0)
    if ct > 15 and e.key.startswith('aten::'):
        rows.append((ct, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for r in rows[:40]:
    print('%9.1f us  x%-3d %-28s %s' % r)
