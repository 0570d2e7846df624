"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not imported by the product package.

CPU (torch fp32, eager) restatement of the arithmetic on the reference's Learner.fit() hot path, written as
plain functions of tensors so that torch autograd on the CPU gives the reference gradients.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the checker /
the CPU comparator.  Every function cites the reference lines it restates (paths relative to the reference
root).  The reference delegates all tensor math to PyTorch (README.md:21 pins torch 1.2), so
"the published algorithm" of that third-party dependency is restated with the same torch primitives on CPU.

Parity pin: these functions are checked against golden vectors produced by RUNNING THE REFERENCE ITSELF
(imported from /root/reference under oracle/_ref_import.py's shims, torch 2.10 CPU) by oracle/gen_golden.py;
the vectors live in tests/golden/*.npz and tests/test_oracle_golden.py is the pin.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------
# K4  CollabFilterNet.forward — Applications/CollabFiltering.py:196-204
# ---------------------------------------------------------------------------------------------------------
def embdotbias(x, U, M, bu, bi, output_range=None):
    """x int64 [n,2]; U [n_user,D]; M [n_item,D]; bu [n_user,1]; bi [n_item,1] -> y [n]."""
    u, it = x[:, 0], x[:, 1]
    res = (U[u] * M[it]).sum(dim=1) + bu[u].squeeze(1) + bi[it].squeeze(1)       # :198-200
    if output_range is not None:
        lo, hi = output_range[0], output_range[1]
        res = lo + (hi - lo) * torch.sigmoid(res)                                # :201-203
    return res


def mse_loss(pred, y):
    "nn.MSELoss(), loss_func_dict['cont'] — General/Learner.py:20"
    return ((pred - y) ** 2).mean()


# ---------------------------------------------------------------------------------------------------------
# Optimizer.step — General/Optimizer.py:58-70 with torch.optim.SGD(momentum) / Adam (General/Learner.py:17-19)
# ---------------------------------------------------------------------------------------------------------
class OptimState:
    """Per-tensor optimizer state for the restated SGD-momentum / Adam updates."""

    def __init__(self, params):
        self.step = 0
        self.buf = [None] * len(params)                       # SGD momentum buffer
        self.m = [torch.zeros_like(p) for p in params]        # Adam exp_avg
        self.v = [torch.zeros_like(p) for p in params]        # Adam exp_avg_sq


def clip_grad_norm(grads, max_norm):
    "torch.nn.utils.clip_grad_norm_ (General/Optimizer.py:54-56): scale all grads by min(1, c/(norm+1e-6))"
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [None if g is None else g * coef for g in grads]


def optimizer_step(params, grads, state, lrs, wds, kind, momentum=0.9, betas=(0.9, 0.999), eps=1e-8,
                   clip=None, decay_mask=None):
    """One Optimizer.step() on a flat list of tensors, in place.
    lrs / wds: per-tensor learning rate and decoupled weight-decay constant (wd_g*lr_g applied as
    X *= 1 - wd*lr, Optimizer.py:60-67; decay_mask[i]=False skips it, i.e. bn groups when bn_wd is False).
    kind: 'sgd' (torch.optim.SGD, dampening 0, no nesterov) or 'adam' (torch.optim.Adam, no amsgrad)."""
    with torch.no_grad():
        for i, p in enumerate(params):
            if wds is not None and wds[i] and (decay_mask is None or decay_mask[i]):
                p.mul_(1 - wds[i] * lrs[i])
        if clip:
            grads = clip_grad_norm(grads, clip)
        state.step += 1
        t = state.step
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is None:
                continue
            if kind == 'sgd':
                if momentum:
                    if state.buf[i] is None:
                        state.buf[i] = g.clone()
                    else:
                        state.buf[i].mul_(momentum).add_(g)
                    g = state.buf[i]
                p.add_(g, alpha=-lrs[i])
            elif kind == 'adam':
                b1, b2 = betas
                state.m[i].mul_(b1).add_(g, alpha=1 - b1)
                state.v[i].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (state.v[i].sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(state.m[i], denom, value=-lrs[i] / bc1)
            else:
                raise ValueError(kind)
