"""Does an EXTERNAL event recorded inside a captured hipGraph let another stream start work in the MIDDLE of a replay?  (round 4: the
mechanism behind all-reduce / backward overlap under Learner.use_graphs() + GradSync)."""
import time
import torch

dev = torch.device('cuda:0')
a = torch.randn(8192, 8192, device=dev)
out1 = torch.zeros(8192, 8192, device=dev); out2 = torch.zeros(8192, 8192, device=dev)
flag = torch.zeros(1024, 1024, device=dev)
side = torch.cuda.Stream()
ev = torch.cuda.Event(external=True)
t_side = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
t_main = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    b = a
    for _ in range(10):
        b = torch.tanh(b * 1.01 + a)
    out1.copy_(b)
    ev.record()                          # <- the "bucket complete" point
    for _ in range(10):
        b = torch.tanh(b * 1.01 + a)
    out2.copy_(b)
for rep in range(3):
    torch.cuda.synchronize()
    t_main[0].record()
    g.replay()
    t_main[1].record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        t_side[0].record()
        flag.add_(1.0)
        t_side[1].record()
    torch.cuda.synchronize()
    print('replay %.2f ms; side work started %.2f ms after the replay began, i.e. %s its end' % (
        t_main[0].elapsed_time(t_main[1]), t_main[0].elapsed_time(t_side[0]),
        'BEFORE' if t_main[0].elapsed_time(t_side[0]) < 0.9 * t_main[0].elapsed_time(t_main[1]) else 'after'))
ref = a
for _ in range(10):
    ref = torch.tanh(ref * 1.01 + a)
print('out1 ok', torch.allclose(out1, ref), 'flag', flag[0, 0].item())
