"""How much of a replayed whole-step hipGraph is the per-step loss read-back?  Replays the captured tabular / collab step of
tools/bench_heads.py with and without `loss.item()` after every replay (the reference's train1minibatch returns the float:
General/Learner.py:516).  Usage: python tools/replay_nosync.py [tabular|collab]"""
import sys
import time

import torch

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['tabular'])
import bench_heads as bh  # noqa: E402

which = sys.argv[1]
captured = {}
orig_run = bh.run


def run(name, learner, batches, lr, unit, units, steps, warmup=3, **kw):
    learner.model.train()
    learner.use_graphs(True)
    for i in range(8):
        learner.train1minibatch(*batches[i % len(batches)], lr, **kw)
    gs = next(iter(learner._graphs.values()))
    opt = learner.optimizer
    from neuralnetworklibrary_amd.General.Learner import _tensor_leaves

    def step(i, sync):
        xb, yb = batches[i % len(batches)]
        for dst, src in zip(_tensor_leaves(gs.x) + _tensor_leaves(gs.y), _tensor_leaves(xb) + _tensor_leaves(yb)):
            dst.copy_(src, non_blocking=True)
        opt.replay_step(gs.opt_capture)
        gs.graph.replay()
        return gs.loss.item() if sync else None
    for sync in (True, False, True, False):
        for i in range(20):
            step(i, sync)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 500
        for i in range(n):
            step(i, sync)
        torch.cuda.synchronize()
        print('%s: loss.item() every step = %s: %.1f us per step' % (name, sync, (time.perf_counter() - t0) / n * 1e6))


bh.run = run
bh.GRAPHS = True
getattr(bh, which)(10) if which == 'tabular' else bh.collab(10, 64)
