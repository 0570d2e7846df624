"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden-vector generator (build container only).

Runs the REAL reference (imported read-only from /root/reference under oracle/_ref_import.py's shims, torch
2.10 CPU — the reference pins torch 1.2, README.md:21; see SURVEY.md §8c for the caveat) on small seeded
inputs and writes inputs + expected outputs to tests/golden/*.npz.  Only data is written: no reference
source text is copied.  Usage:  python oracle/gen_golden.py [g1 g2 ...]   (default: all)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
R = _ref_import.load()
Learner = R['General.Learner'].Learner
Optimizer = R['General.Optimizer'].Optimizer


def A(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, {k: getattr(v, 'shape', None) for k, v in arrays.items()}, '%.1f KB' % (os.path.getsize(path) / 1024))


class FakeData:
    """Minimal data object honouring the Learner's data protocol (General/Learner.py:103-110,504,557,596)."""

    def __init__(self, train_batches, val_batches, bs, target_type):
        self.train_dl, self.val_dl, self.bs, self.target_type = train_batches, val_batches, bs, target_type


# ---------------------------------------------------------------------------------------------------------
def g1_collab():
    """G1: CollabFilterNet fwd / MSE / 4 grads / params after 3 Adam train1minibatch steps (wd=1e-4)."""
    CF = R['Applications.CollabFiltering']
    torch.manual_seed(101)
    n_user, n_item, D, bs = 50, 40, 8, 64
    net = CF.CollabFilterNet(n_user, n_item, D, [0.8, 5.2])
    with torch.no_grad():                      # larger than the 0.01-std init so the dot term matters
        for p in net.parameters():
            p.mul_(30.0)
    g = torch.Generator().manual_seed(7)
    xs = [torch.stack([torch.randint(0, n_user, (bs,), generator=g), torch.randint(0, n_item, (bs,), generator=g)], 1)
          for _ in range(3)]
    ys = [torch.randint(1, 6, (bs,), generator=g).float() for _ in range(3)]
    out = {'n_user': n_user, 'n_item': n_item, 'D': D, 'lo': 0.8, 'hi': 5.2}
    for k, p in net.state_dict().items():
        out['init.' + k] = A(p)
    for i in range(3):
        out['x%d' % i], out['y%d' % i] = A(xs[i]), A(ys[i])
    pred = net(xs[0])
    loss = torch.nn.MSELoss()(pred, ys[0])
    loss.backward()
    out['pred0'], out['loss0'] = A(pred), A(loss)
    for k, p in net.named_parameters():
        out['grad0.' + k] = A(p.grad)
    # no-range variant of the forward (output_range=None branch, CollabFiltering.py:201)
    net.output_range = None
    out['pred0_norange'] = A(net(xs[0]))
    net.output_range = [0.8, 5.2]
    # 3 steps of the reference Learner.train1minibatch with Adam, lr 1e-2, wd 1e-4
    data = FakeData(list(zip(xs, ys)), list(zip(xs, ys)), bs, 'cont')
    learner = Learner('/tmp/nnl_golden_g1', data, net, optimizer='Adam')
    learner.init_optimizer(wd=1e-4)
    losses = [learner.train1minibatch(xs[i], ys[i], 1e-2) for i in range(3)]
    out['step_losses'] = np.array(losses, dtype=np.float64)
    for k, p in net.state_dict().items():
        out['after3.' + k] = A(p)
    save('g1_collab', **out)


GROUPS = {'g1': g1_collab}

if __name__ == '__main__':
    names = sys.argv[1:] or sorted(GROUPS)
    for n in names:
        GROUPS[n]()
