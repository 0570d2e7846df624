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


# ---------------------------------------------------------------------------------------------------------
# K6  detection loss — Applications/Vision.py:234-256, 1474-1511, 1513-1530, 1532-1566, 1568-1605, 1620-1644
# ---------------------------------------------------------------------------------------------------------
def jaccard(B1, B2):
    "IoU matrix [n,m] of min-max boxes, no +1 (Vision.py:234-256)"
    a1 = (B1[:, 2] - B1[:, 0]) * (B1[:, 3] - B1[:, 1])
    a2 = (B2[:, 2] - B2[:, 0]) * (B2[:, 3] - B2[:, 1])
    b1, b2 = B1.unsqueeze(1), B2.unsqueeze(0)
    iw = (torch.min(b1[:, :, 2], b2[:, :, 2]) - torch.max(b1[:, :, 0], b2[:, :, 0])).clamp(min=0)
    ih = (torch.min(b1[:, :, 3], b2[:, :, 3]) - torch.max(b1[:, :, 1], b2[:, :, 1])).clamp(min=0)
    inter = iw * ih
    return inter / (a1.unsqueeze(1) + a2.unsqueeze(0) - inter)


def match_anchors_objects(objects, anchors, pos_thresh=0.5, neg_thresh=0.4):
    "state per anchor: matched object index (>=0) if max IoU > .5, -1 if < .4, -2 otherwise (Vision.py:1474-1511)"
    N = len(anchors)
    if len(objects) == 0:
        return torch.full((N,), -1, dtype=torch.long)
    mx, arg = torch.max(jaccard(objects, anchors), dim=0)
    state = torch.full((N,), -2, dtype=torch.long)
    state[mx < neg_thresh] = -1
    state[mx > pos_thresh] = arg[mx > pos_thresh]
    return state


def focal_loss_retina(pred, target, alpha=0.25, gamma=2.0):
    "Vision.py:1513-1530"
    p = pred.clamp(1e-4, 1.0 - 1e-4)
    pt = p * target + (1 - p) * (1 - target)
    w = (alpha * target + (1 - alpha) * (1 - target)) * (1 - pt).pow(gamma)
    losses = -w * (target * torch.log(p) + (1 - target) * torch.log(1 - p))
    return losses.sum() / target.sum().clamp(min=1)


def smoothL1_loss_retina(anchs, pred_shift, target):
    "Vision.py:1532-1566"
    aw, ah = anchs[:, 2] - anchs[:, 0], anchs[:, 3] - anchs[:, 1]
    ax, ay = anchs[:, 0] + 0.5 * aw, anchs[:, 1] + 0.5 * ah
    tw, th = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1]
    tx, ty = target[:, 0] + 0.5 * tw, target[:, 1] + 0.5 * th
    tw, th = tw.clamp(min=1), th.clamp(min=1)
    true = torch.stack(((tx - ax) / aw, (ty - ay) / ah, torch.log(tw / aw), torch.log(th / ah))).t()
    true = true / torch.tensor([[0.1, 0.1, 0.2, 0.2]], dtype=true.dtype)
    diff = torch.abs(true - pred_shift)
    losses = 0.5 * 9 * diff.pow(2) * (diff < 1 / 9).to(diff.dtype) + (diff - 0.5 / 9) * (diff >= 1 / 9).to(diff.dtype)
    return losses.mean()


def ssd_loss(anchors, reg, clas, BBoxes, Cats, beta=0.5, alpha=0.25, gamma=2.0):
    """(total, reg_loss, clas_loss) of a batch: per image strip the -1 padding, match, build the one-hot targets of
    the positive anchors, focal over pos+neg anchors, smooth-L1 over pos anchors; batch means (Vision.py:1568-1644)."""
    bs, K = len(BBoxes), clas.shape[2]
    reg_loss, clas_loss = torch.zeros((), dtype=clas.dtype), torch.zeros((), dtype=clas.dtype)
    for i in range(bs):
        keep = Cats[i] >= 0
        boxes, cats = BBoxes[i][keep], Cats[i][keep]
        state = match_anchors_objects(boxes, anchors)
        pos, used = state >= 0, state != -2
        targ = torch.zeros(len(anchors), K, dtype=clas.dtype)
        if pos.any():
            targ[pos.nonzero().view(-1), cats[state[pos]]] = 1
        clas_loss = clas_loss + focal_loss_retina(clas[i][used], targ[used], alpha, gamma)
        if pos.any():
            reg_loss = reg_loss + smoothL1_loss_retina(anchors[pos], reg[i][pos], boxes[state[pos]])
    reg_loss, clas_loss = reg_loss / bs, clas_loss / bs
    return (1 - beta) * reg_loss + beta * clas_loss, reg_loss, clas_loss


def anchors_for(H, W, ratios=(0.5, 1, 2), scales=(2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3))):
    """float32 [sum_l H_l*W_l*9, 4] anchors of pyramid levels 3..7: stride 2^l, base size 2^(l+2), 3 ratios x 3 scales
    (ratio-major), centres (i+0.5)*stride, cell-major / anchor-minor order (retinanet.py:439-495; numpy fp64 -> fp32)."""
    S = np.tile(np.array(scales), len(ratios))
    Rt = np.repeat(np.array(ratios), len(scales))
    Hs, Ws = S / np.sqrt(Rt), S * np.sqrt(Rt)
    base = np.array([-Ws / 2, -Hs / 2, Ws / 2, Hs / 2]).T
    out = []
    for lvl in [3, 4, 5, 6, 7]:
        stride, size = 2 ** lvl, 2 ** (lvl + 2)
        gh, gw = (H + stride - 1) // stride, (W + stride - 1) // stride
        sx, sy = np.meshgrid((np.arange(gw) + 0.5) * stride, (np.arange(gh) + 0.5) * stride)
        shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], 1)
        out.append((shifts[:, None, :] + (size * base)[None, :, :]).reshape(-1, 4))
    return torch.from_numpy(np.concatenate(out).astype(np.float32))
