"""cProfile of the HOST side of the pipelined replayed step (collab / tabular): where do the microseconds of Python go?
Usage (from tools/): python prof_fit_host.py [collab|tabular]"""
import cProfile
import pstats
import sys

import torch

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['collab'])
import bench_heads as bh  # noqa: E402

which = sys.argv[1]


def run(name, learner, batches, lr, unit, units, steps, warmup=3, **kw):
    learner.model.train()
    learner.use_graphs(True)
    for i in range(8):
        learner.train1minibatch(*batches[i % len(batches)], lr, **kw)
    pend = [None]

    def step(i):
        r = learner.train1minibatch(*batches[i % len(batches)], lr, _defer=True, **kw)
        if pend[0] is not None:
            pend[0].result()
        pend[0] = r if hasattr(r, 'result') else None
    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(2000):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(22)


bh.run = run
bh.GRAPHS = True
bh.tabular(10) if which == 'tabular' else bh.collab(10, 64)
