"""torch.autograd.Function wrappers around the C ABI of libnnl_hip.so (include/nnl.h).

Each Function is the device-side replacement of one group of eager torch ops in the reference (call sites
cited per class).  Tensors are handed over as raw device pointers + sizes on torch's CURRENT stream; there
is no CPU fallback — a non-CUDA tensor raises NnlError.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream

__all__ = ['mse_loss', 'scaled_sigmoid', 'embdotbias', 'index_error_flag', 'raise_if_index_error', 'conv2d', 'to_nhwc', 'from_nhwc', 'linear', 'bn_act', 'concat_pool2d', 'TabularPlan', 'tab_embed_concat', 'embedding_renorm_drop', 'retina_loss']

_ERR_FLAGS = {}


def _ab(name, default):
    """A/B hook of a CLOSED experiment: the shipped package runs its default; the variable is read only in an
    A/B session (NNL_AB=1 in the environment, with the `make AB=1` library for the C-side hooks)."""
    return os.environ.get(name, default) if os.environ.get('NNL_AB') == '1' else default



def _flags(device):
    key = (device.type, device.index)
    if key not in _ERR_FLAGS:
        _ERR_FLAGS[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return _ERR_FLAGS[key]


def index_error_flag(device):
    """Per-device int32 flag that gather kernels set when they meet an out-of-range index (the sample is
    skipped, nothing faults).  Checked without a per-step sync by `raise_if_index_error()`."""
    return _flags(device)[0:1]


def lstm_timeout_flag(device):
    """Its own word next to the index flag: the persistent LSTM kernels store 2 here when their grid barrier times out (a gather
    kernel's plain store of 1 into the index word can then no longer overwrite it)."""
    return _flags(device)[1:2]


def raise_for_flag(code):
    "bit 0: index out of range (torch's nn.Embedding failure); bit 1: persistent-LSTM barrier time-out"
    if code & 2:
        raise _lib.NnlError("the persistent LSTM kernel's grid barrier timed out (results of that step are invalid)")
    if code & 1:
        raise IndexError("index out of range in self")


def raise_if_index_error():
    """One D2H read per call: raises IndexError (torch's nn.Embedding failure) if any gather kernel since the
    last call saw an out-of-range index.  The Learner calls this at the end of every epoch, evaluate() and predict()
    (General/Learner.py `_raise_if_index_error`, which also makes the decision rank-uniform under data parallelism)."""
    for flag in _ERR_FLAGS.values():
        v = flag.tolist()
        code = (1 if v[0] else 0) | (2 if v[1] else 0)
        if code:
            flag.zero_()
            raise_for_flag(code)


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class _EmbDotBias(torch.autograd.Function):
    """CollabFilterNet.forward (reference Applications/CollabFiltering.py:196-204): four embedding gathers,
    row-wise dot, bias adds and the scaled sigmoid in ONE kernel; backward = the four dense scatter-adds."""

    @staticmethod
    def forward(ctx, x, U, M, bu, bi, lo, hi):
        require_cuda(x, U, M, bu, bi)
        x = x.contiguous()
        if x.dtype != torch.int64:
            x = x.long()
        U, M, bu, bi = _f32c(U), _f32c(M), _f32c(bu), _f32c(bi)
        n, D = x.shape[0], U.shape[1]
        y = torch.empty(n, dtype=torch.float32, device=x.device)
        z = torch.empty(n, dtype=torch.float32, device=x.device)
        has_range = lo is not None
        check(lib.nnl_embdotbias_fwd(ptr(x), ptr(U), ptr(M), ptr(bu), ptr(bi), ptr(y), ptr(z), n, U.shape[0],
                                     M.shape[0], D, int(has_range), float(lo or 0.), float(hi or 0.),
                                     ptr(index_error_flag(x.device)), stream()))
        ctx.save_for_backward(x, U, M, z)
        ctx.rng = (has_range, float(lo or 0.), float(hi or 0.))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, U, M, z = ctx.saved_tensors
        has_range, lo, hi = ctx.rng
        dy = _f32c(dy)
        nu, ni, D = U.shape[0], M.shape[0], U.shape[1]
        flat = torch.empty((nu + ni) * (D + 1), dtype=torch.float32, device=U.device)       # [dU | dM | dbu | dbi]: one fill in C
        dU, dM = flat[:nu * D].view(nu, D), flat[nu * D:(nu + ni) * D].view(ni, D)
        dbu, dbi = flat[(nu + ni) * D:(nu + ni) * D + nu].view(nu, 1), flat[(nu + ni) * D + nu:].view(ni, 1)
        wsb = int(lib.nnl_embdotbias_bwd_workspace_bytes(x.shape[0]))          # sample-order (deterministic) scatter-add
        ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=U.device)
        check(lib.nnl_embdotbias_bwd(ptr(x), ptr(U), ptr(M), ptr(z), ptr(dy), ptr(dU), ptr(dM), ptr(dbu), ptr(dbi),
                                     x.shape[0], U.shape[0], M.shape[0], U.shape[1], int(has_range), lo, hi,
                                     ptr(ws), wsb, stream()))
        return None, dU, dM, dbu, dbi, None, None


class _MSE(torch.autograd.Function):
    """nn.MSELoss() (reference General/Learner.py:20, the 'cont' loss): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, pred, target):
        require_cuda(pred, target)
        p, t = _f32c(pred).reshape(-1), _f32c(target).reshape(-1)
        n = p.numel()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        wsb = int(lib.nnl_mse_workspace_bytes(n))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=p.device) if wsb else None
        check(lib.nnl_mse_fwd(ptr(p), ptr(t), ptr(loss), n, ptr(ws), wsb, stream()))
        ctx.save_for_backward(p, t)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, dloss):
        p, t = ctx.saved_tensors
        g = _f32c(dloss).reshape(1)
        dp = torch.empty_like(p)
        check(lib.nnl_mse_bwd(ptr(p), ptr(t), ptr(g), ptr(dp), p.numel(), stream()))
        return dp.view(ctx.shape), None


def mse_loss(pred, target):
    "mean((pred - target)^2) as a 0-dim tensor; gradient with respect to `pred` only (targets are data)"
    return _MSE.apply(pred, target)


class _ScaledSigmoid(torch.autograd.Function):
    """lo + (hi - lo) * sigmoid(x): FullyConnectedNet's 'sigmoidal' output activation (reference General/Layers.py:150-152)"""

    @staticmethod
    def forward(ctx, x, lo, hi):
        require_cuda(x)
        xc = _f32c(x)
        y, sg = torch.empty_like(xc), torch.empty_like(xc)
        check(lib.nnl_scaled_sigmoid_fwd(ptr(xc), ptr(y), ptr(sg), xc.numel(), float(lo), float(hi), stream()))
        ctx.save_for_backward(sg)
        ctx.rng = (float(lo), float(hi))
        return y

    @staticmethod
    def backward(ctx, dy):
        sg, = ctx.saved_tensors
        dyc = _f32c(dy)
        dx = torch.empty_like(sg)
        check(lib.nnl_scaled_sigmoid_bwd(ptr(dyc), ptr(sg), ptr(dx), sg.numel(), ctx.rng[0], ctx.rng[1], stream()))
        return dx, None, None


def scaled_sigmoid(x, lo, hi):
    return _ScaledSigmoid.apply(x, float(lo), float(hi))


def embdotbias(x, U, M, bu, bi, output_range=None):
    """y = lo + (hi-lo)*sigmoid(<U[x[:,0]], M[x[:,1]]> + bu[x[:,0]] + bi[x[:,1]]); x int64 [n,2]."""
    lo, hi = (None, None) if output_range is None else (float(output_range[0]), float(output_range[1]))
    return _EmbDotBias.apply(x, U, M, bu, bi, lo, hi)


# ---------------------------------------------------------------------------------------------------------
# K1 conv2d.  Logical tensors stay NCHW / [K,C,R,S] (drop-in API, state_dict layout); the kernels see their
# channels_last physical images: activations NHWC, filters KRSC.
# ---------------------------------------------------------------------------------------------------------
def to_nhwc(x):
    """[N,C,H,W] logical -> contiguous [N,H,W,C] tensor (no copy when x is already channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


def from_nhwc(x):
    """contiguous [N,H,W,C] -> logical [N,C,H,W] view (channels_last strides)."""
    return x.permute(0, 3, 1, 2)


def _geom(N, H, W, C, K, R, S, stride, pad):
    P = (H + 2 * pad - R) // stride + 1
    Q = (W + 2 * pad - S) // stride + 1
    return _lib.ConvGeom(N, H, W, C, K, R, S, stride, pad, P, Q)


def _pad_c4(t_nhwc):
    """pad the innermost (channel) dim of an NHWC / KRSC tensor to a multiple of 4 with zeros."""
    C = t_nhwc.shape[-1]
    if C % 4 == 0:
        return t_nhwc
    if t_nhwc.is_cuda and t_nhwc.dtype == torch.float32 and t_nhwc.is_contiguous() and not (torch.is_grad_enabled() and t_nhwc.requires_grad):
        Cp = C + 4 - C % 4                                # one launch (torch's pad is a fill + a strided copy)
        out = torch.empty(t_nhwc.shape[:-1] + (Cp,), dtype=torch.float32, device=t_nhwc.device)
        check(lib.nnl_pad_cols(ptr(t_nhwc), ptr(out), t_nhwc.numel() // C, C, Cp, stream()))
        return out
    return torch.nn.functional.pad(t_nhwc, (0, 4 - C % 4))


# ---- all filter transposes of a backward pass in one launch --------------------------------------------------------------
_WT_ACTIVE = {}       # weight data_ptr -> W^T tensor [C,R,S,K]; valid ONLY between prepare_backward() and finish_backward()
_WT_PADDED = {}       # (weight data_ptr, shape, pad) -> transposed, channel-padded filter of a K % 16 != 0 dgrad; same window


# Prepared Winograd filters (include/nnl.h: nnl_wino_filter_multi).  The transformed filter of a 3x3 / stride 1 / pad 1 layer depends
# on its weights only; transforming it inside every convolution call costs one extra launch per call (58 per ResNet-34 step).
# `prepare_forward(model)` (Learner, before the forward of a training step) and `prepare_backward(model)` transform the filters of all
# layers that took the Winograd kernel at their LAST call (`_WINO_PREF`) in one launch each; the convolutions pick them up by weight
# address.  The window closes in finish_backward() (Learner: in a `finally`), before the optimizer touches the weights.
_WINO_U_FWD, _WINO_U_BWD = {}, {}     # (weight data_ptr, kernel mode) -> U [K,4|16,3|1,C] (forward) / U' [C,...,K] (dgrad) in that kernel's layout
_WINO_GEN = [0]                       # training-step counter (prepare_forward): the recorded modes of a weight are those of its LAST step
_WINO_PREF = {}                       # weight data_ptr -> [modes the forward calls took, modes the dgrad calls took] (sets; 1 = 1-D, 2 = 2-D kernel):
                                      # a weight shared between geometries (RetinaNet's heads: five pyramid levels) may need BOTH layouts


class _WinoBatch:
    """persistent U buffers + device descriptor tables for nnl_wino_filter_multi: items = [(key_ptr, src tensor [rows,3,3,ch], flip, mode)],
    mode as nnl_conv2d_wino_preferred: 1 -> U [rows,12,ch] (1-D kernel), 2 -> U [rows,16,ch] (2-D kernel)"""

    def __init__(self, items):
        import numpy as np
        dev = items[0][1].device
        self.key = _wino_batch_key(items)
        total = sum(_wino_u_numel(t.shape[0], t.shape[3], mode) for _, t, _, mode in items)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        desc = np.zeros(len(items), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('rows', '<i4'), ('ch', '<i4'), ('flip', '<i4'), ('first', '<i4'),
                                                    ('two_d', '<i4'), ('reserved', '<i4')]))
        block_desc, self.views, off, first = [], {}, 0, 0
        for i, (k, t, flip, mode) in enumerate(items):
            rows, ch = t.shape[0], t.shape[3]
            n = _wino_u_numel(rows, ch, mode)
            u = self.flat[off:off + n]
            off += n
            nb = (rows * ch + 255) // 256 if mode == 2 else (rows * 3 * ch + 255) // 256
            desc[i] = (t.data_ptr(), u.data_ptr(), rows, ch, flip, first, 1 if mode == 2 else 0, 0)
            block_desc += [i] * nb
            first += nb
            self.views[(k, mode)] = u
        self.n_blocks = first
        self.desc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        self.block_desc = torch.tensor(block_desc, dtype=torch.int32, device=dev)

    def run(self):
        check(lib.nnl_wino_filter_multi(ptr(self.desc), ptr(self.block_desc), self.n_blocks, stream()))


def _wino_u_numel(rows, ch, mode):
    "floats of the transformed filter in the layout of kernel `mode` (include/nnl.h: nnl_conv2d_wino_preferred)"
    return rows * 16 * ch if mode == 2 else rows * 12 * ch


def _wino_batch_key(items):
    return tuple((k, t.data_ptr(), tuple(t.shape), mode) for k, t, _, mode in items)


def _conv_mods(model):
    mods = getattr(model, '_nnl_conv_mods', None)
    if mods is None:                # the module walk costs ~0.1 ms per step on a host-bound model: done once.  A conv added
        mods = [m for m in model.modules() if getattr(m, 'nnl_hip_conv', False)]     # later transforms its own filter (slower, correct)
        object.__setattr__(model, '_nnl_conv_mods', mods)
    return mods


def _run_wino_batch(model, attr, items, out):
    if not items:
        return
    key = _wino_batch_key(items)
    batch = getattr(model, attr, None)
    if batch is None or batch.key != key:
        if torch.cuda.is_current_stream_capturing():
            return                                      # (buffers are built during eager steps)
        batch = _WinoBatch(items)
        object.__setattr__(model, attr, batch)
    batch.run()
    out.update(batch.views)


def _wino_modes(key_ptr, which):
    e = _WINO_PREF.get(key_ptr)
    return e[which][1] if e is not None else ()


def prepare_forward(model):
    """Call right before the forward of a TRAINING step (Learner does): the Winograd filters of every HipConv2d whose last forward
    took the Winograd kernel, in one launch; valid until finish_backward()."""
    _WINO_U_FWD.clear()
    _WINO_GEN[0] += 1                                    # a new training step: the modes recorded from here on replace the previous step's
    if os.environ.get('NNL_WINO_PREPARE', '1') == '0':
        return
    items = []
    for m in _conv_mods(model):
        w = m.weight
        modes = _wino_modes(w.data_ptr(), 0)
        if (modes and w.is_cuda and w.dim() == 4 and w.shape[2] == 3 and w.shape[3] == 3 and w.dtype == torch.float32
                and w.is_contiguous(memory_format=torch.channels_last)):
            for mode in sorted(modes):
                items.append((w.data_ptr(), w.permute(0, 2, 3, 1), 0, mode))    # KRSC view of the same memory
    _run_wino_batch(model, '_nnl_wino_fwd_batch', items, _WINO_U_FWD)


class _WtBatch:
    "persistent W^T buffers + device descriptor tables for the conv filters of one model (nnl_conv2d_weight_transpose_multi)"

    def __init__(self, weights):
        import numpy as np
        dev = weights[0].device
        self.key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
        total = sum(w.numel() for w in weights)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        desc = np.zeros(len(weights), dtype=np.dtype([('w', '<u8'), ('wt', '<u8'), ('K', '<i4'), ('RS', '<i4'), ('C', '<i4'), ('first', '<i4')]))
        tile_tensor, self.views, off, first = [], {}, 0, 0
        for i, w in enumerate(weights):
            K, C, R, S = w.shape
            wt = self.flat[off:off + w.numel()].view(C, R, S, K)
            off += w.numel()
            n_tiles = R * S * ((K + 31) // 32) * ((C + 31) // 32)
            desc[i] = (w.data_ptr(), wt.data_ptr(), K, R * S, C, first)
            tile_tensor += [i] * n_tiles
            first += n_tiles
            self.views[w.data_ptr()] = wt
        self.n_tiles, self.total = first, float(total)
        self.desc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        self.tile_tensor = torch.tensor(tile_tensor, dtype=torch.int32, device=dev)

    def run(self):
        check(lib.nnl_conv2d_weight_transpose_multi(ptr(self.desc), ptr(self.tile_tensor), self.n_tiles, self.total, stream()))


def prepare_backward(model):
    """Call right before `loss.backward()` (Learner does): transposes the filters of every HipConv2d of `model` in ONE launch
    (one small launch per layer otherwise) and exposes them to the convolutions' backward until finish_backward().  The
    window is deliberately that short: a filter modified later can never meet a stale transpose."""
    mods = _conv_mods(model)
    ws = [m.weight for m in mods
          if m.weight.is_cuda and m.weight.dim() == 4 and m.weight.shape[1] % 4 == 0
          and m.weight.shape[0] % 4 == 0 and m.weight.dtype == torch.float32
          and m.weight.is_contiguous(memory_format=torch.channels_last)]
    if len(ws) < 2 or sum(w.requires_grad for w in ws) * 2 < len(ws):
        return
    key = tuple((w.data_ptr(), tuple(w.shape)) for w in ws)
    batch = getattr(model, '_nnl_wt_batch', None)
    if batch is None or batch.key != key:
        if torch.cuda.is_current_stream_capturing():
            return                                      # (buffers are built during the eager warm-up steps)
        batch = _WtBatch(ws)
        object.__setattr__(model, '_nnl_wt_batch', batch)
    batch.run()
    _WT_ACTIVE.clear()
    _WT_ACTIVE.update(batch.views)
    # the dgrad Winograd filters U' [C,4,3,K] of the layers whose last dgrad took the Winograd kernel, from the transposes above
    _WINO_U_BWD.clear()
    if os.environ.get('NNL_WINO_PREPARE', '1') != '0':
        items = [(w.data_ptr(), batch.views[w.data_ptr()], 1, mode) for w in ws if w.shape[2] == 3 and w.shape[3] == 3
                 for mode in sorted(_wino_modes(w.data_ptr(), 1))]
        _run_wino_batch(model, '_nnl_wino_bwd_batch', items, _WINO_U_BWD)


def finish_backward():
    side_join()
    _finish_backward_impl()


def _finish_backward_impl():
    "closes the window of everything prepare_forward / prepare_backward exposed (the optimizer is about to change the weights)"
    _WT_ACTIVE.clear()
    _WT_PADDED.clear()
    _WINO_U_FWD.clear()
    _WINO_U_BWD.clear()


def _wino_pref(key_ptr, which, g):
    """which kernel nnl_conv2d_fwd (which = 0) / nnl_conv2d_dgrad (1) takes for g: 0 direct, 1 Winograd 1-D, 2 Winograd 2-D.  Remembered
    per weight for the next step's batch (a weight shared between geometries keeps the mode of its LAST call; calls whose mode
    differs transform their own filter: the prepared buffer is only handed over when its size is that of the mode's layout)"""
    pref = int(lib.nnl_conv2d_wino_preferred(g, which)) if (g.R == 3 and g.S == 3 and g.stride == 1) else 0
    if pref or key_ptr in _WINO_PREF:
        if len(_WINO_PREF) > 8192:
            _WINO_PREF.clear()
        e = _WINO_PREF.setdefault(key_ptr, [[-1, set()], [-1, set()]])[which]
        if e[0] != _WINO_GEN[0]:                         # first call of this training step (prepare_forward opens a new one): forget the old modes
            e[0], e[1] = _WINO_GEN[0], set()
        if pref:
            e[1].add(pref)
    return pref


# Gradient buffers whose rows are ALREADY padded with zeros to the GEMM granularity (the softmax-CE backward at V % 16 != 0, the
# LSTM's dgates at 4H % 32 != 0): the producer registers the padded base tensor, hands autograd the [:, :K] view, and the linear
# layer that receives it (through whatever reshape / permute views autograd inserts) recognises its base and uses the padded rows
# directly.  Keyed by id() and guarded by a weak reference, so a recycled id can never match.
_PADDED_GRADS = {}


def register_padded_grad(base, ld):
    import weakref
    if len(_PADDED_GRADS) > 64:
        for k in [k for k, (r, _) in _PADDED_GRADS.items() if r() is None]:
            del _PADDED_GRADS[k]
    _PADDED_GRADS[id(base)] = (weakref.ref(base), int(ld))


def _padded_rows(dy, K):
    "dy logical [N, K, 1, 1]: the [N,1,1,ld] tensor over its registered zero-padded base, or None"
    base = dy._base if dy._is_view() else None
    ent = _PADDED_GRADS.get(id(base)) if base is not None else None
    if ent is None or ent[0]() is not base or dy.dim() != 4 or dy.shape[2] != 1 or dy.shape[3] != 1:
        return None
    ld = ent[1]
    if dy.dtype != torch.float32 or dy.stride(1) != 1 or dy.stride(0) != ld or ld < K or ld % 16 != 0 or dy.storage_offset() % ld != 0:
        return None
    if dy.storage_offset() + dy.shape[0] * ld > base.numel():
        return None
    return torch.as_strided(base, (dy.shape[0], 1, 1, ld), (ld, ld, ld, 1), dy.storage_offset())


_TILE_COUNTERS = {}


def _tile_counters(device):
    """persistent zero-at-rest ticket counters of the conv kernels' in-kernel split-tile fix-up (include/nnl.h), one buffer per
    device: every op of this package runs on torch's current stream, one conv at a time (a captured step replays on that same
    stream).  Created on first use outside stream capture (eager warm-up steps precede every capture)."""
    import os
    if _ab('NNL_IGEMM_FIXUP', '1') == '0':
        return None
    t = _TILE_COUNTERS.get(device.index)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = _TILE_COUNTERS[device.index] = torch.zeros(int(lib.nnl_conv2d_tile_counters()), dtype=torch.int32, device=device)
    return t


class GradSlot:
    """Side channel between two autograd Functions of one residual block: the BN that closes the block parks the gradient of
    its identity shortcut here (instead of returning it to autograd) and the block's FIRST convolution adds it in the
    epilogue of its dgrad kernel — one accumulation kernel and three tensor passes less per block.  Valid because the first
    convolution's backward always runs after the closing BN's backward (it depends on it through the main path) and both
    consume the same tensor x.
    A block with a projection shortcut uses it the same way: the downsample convolution (`give_slot`) parks ITS input gradient
    here and returns none to autograd.  There is no dependency between the two convolutions' backward nodes (the engine runs
    the later-created downsample first), so the hand-over is guarded: the consumer closes the slot when it runs, and a
    producer that finds it closed returns its gradient to autograd as usual."""
    __slots__ = ('tensor', 'closed')

    def __init__(self):
        self.tensor = None
        self.closed = False



# ---- weight gradients on a SIDE STREAM (round 4) ---------------------------------------------------------------------------------
# The language model's backward is a chain of latency-bound launches (210 BPTT timesteps x 2 kernels: <= 256 small workgroups and ~6 us
# of dependent-launch gap each) with five big weight-gradient GEMMs hanging off it (decoder 170 GFLOP, dW_ih / dW_hh 47 GFLOP each)
# that nothing downstream in backward depends on.  Call sites that opt in (`linear(..., wgrad_side=True)`, ops_text._LSTMRecurrence)
# launch those GEMMs on a second HIP stream, where they fill the CUs the recurrence leaves idle; the main stream joins it in
# finish_backward() / before the fused optimizer step (side_join).  Safety: only when the parameter has no gradient yet (autograd then
# takes the produced tensor as `.grad` without touching its data — an accumulation kernel on the main stream would race), not under
# data parallelism (the bucket hooks copy on the main stream), not while a hipGraph is being captured; every tensor the side launches
# touch is recorded on that stream for the caching allocator; a gradient that needs un-padding is made dense ON the side stream (a strided view would be cloned by autograd
# on the main stream: a race — found by tests/test_text.py).  Measured: the GEMMs do run beside the recurrence, but they take its CUs: kernel times
# LSTM 7.08 -> 9.07 ms, wgrad 3.73 -> 4.39 ms, wall 17.33 -> 17.21 ms — hence opt-in (NNL_WGRAD_SIDE_STREAM=1), kept for the record.
class _Side:
    stream = None
    used = False
    enabled = _ab('NNL_WGRAD_SIDE_STREAM', '0') == '1'      # OPT-IN: measured +0.7 % on the LM step (profiles/r4_lm_side_stream_ab.log)
    pending_param = None          # set by linear(..., wgrad_side=True) for the _Conv2d.forward that follows


def side_ok(param):
    return (_Side.enabled and param is not None and param.is_cuda and param.grad is None and getattr(param, '_nnl_grad_dst', None) is None
            and not torch.cuda.is_current_stream_capturing())


def side_run(fn, tensors):
    "fn() launches its kernels on the current stream: run it with the side stream current, after everything queued on the main stream so far"
    main = torch.cuda.current_stream()
    if _Side.stream is None:
        _Side.stream = torch.cuda.Stream()
    s = _Side.stream
    s.wait_stream(main)
    with torch.cuda.stream(s):
        fn()
    for t in tensors:
        if t is not None:
            t.record_stream(s)
    _Side.used = True


def _side_wgrad(run_w, tensors, dwn, K, c_in):
    "a linear layer's weight gradient on the side stream; returns a DENSE [K, c_in, 1, 1] tensor (un-padded there, not by autograd on the main stream)"
    if K == dwn.shape[0] and c_in == dwn.shape[3]:
        side_run(run_w, tensors)
        return from_nhwc(dwn)
    dense = torch.empty((K, c_in, 1, 1), dtype=torch.float32, device=dwn.device)

    def run():
        run_w()
        dense.view(K, c_in).copy_(dwn.view(dwn.shape[0], dwn.shape[3])[:K, :c_in])
    side_run(run, tuple(tensors) + (dense,))
    return dense


def side_join():
    "the main stream waits for the side-stream weight gradients (before anything reads them: optimizer, accumulation into a tied weight)"
    if _Side.used:
        torch.cuda.current_stream().wait_stream(_Side.stream)
        _Side.used = False


# ---- parameters shared by several forward calls of one step (RetinaNet's heads on the five pyramid levels) ----------------------------
# Autograd sums the per-call gradients of such a parameter with a chain of accumulation kernels (4 adds per tensor and step; 91 ATen
# launches per RetinaNet step, 0.54 ms: profiles/r4_retinanet_kernel_stats.csv).  `shared_params(modules, n)` hands every parameter
# of `modules` out as n ALIASES (same storage) produced by one autograd node whose backward adds the n gradients in call order with ONE
# launch (nnl_sum_tensors); HipConv2d.forward takes the next alias of its weight / bias (`fan_param`).  The same sum as autograd's
# accumulation up to the order of the additions (call order here, backward order there); bitwise reproducible run to run.
_FAN = {}


class _FanOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, n):
        ctx.n = n
        return tuple(p.detach().view(p.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        have = [g for g in grads if g is not None]
        if not have:
            return None, None
        if len(have) == 1:
            return have[0], None
        g0 = have[0]
        dense = all(g.is_cuda and g.dtype == torch.float32 and g.shape == g0.shape and g.stride() == g0.stride() and g.data_ptr() % 16 == 0
                    and (g.is_contiguous() or (g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last))) for g in have)
        if not dense or len(have) > 8:
            out = have[0]
            for g in have[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(g0)                       # (preserves the dense layout of the operands)
        arr = (ctypes.c_void_p * len(have))(*[g.data_ptr() for g in have])
        check(lib.nnl_sum_tensors(arr, len(have), ptr(out), g0.numel(), stream()))
        return out, None


class shared_params:
    """with shared_params([module, ...], n_calls): ... the n_calls forward calls of the modules ...   (training with autograd only:
    outside it — evaluation, no_grad — the modules use their parameters directly)"""

    def __init__(self, modules, n):
        self.modules, self.n, self.keys = modules, int(n), []

    def __enter__(self):
        if self.n > 1 and torch.is_grad_enabled():
            for m in self.modules:
                for p in m.parameters():
                    if p.requires_grad and p.is_cuda and id(p) not in _FAN:
                        _FAN[id(p)] = list(_FanOut.apply(p, self.n))
                        self.keys.append(id(p))
        return self

    def __exit__(self, *exc):
        for k in self.keys:
            _FAN.pop(k, None)
        return False


def fan_param(p):
    "the next alias of a parameter inside `shared_params`, else the parameter itself"
    if p is None or not _FAN:
        return p
    lst = _FAN.get(id(p))
    return lst.pop(0) if lst else p


class _Conv2d(torch.autograd.Function):
    """nn.Conv2d forward/backward (reference Applications/VisionModels/retinanet.py:26-28,66-71,106-124,169-185,
    241-257,304,345) on the fp32-MFMA implicit-GEMM kernels; optional fused bias + ReLU epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, slot=None, bn_pivot=None, give_slot=None):
        require_cuda(x, weight, bias)
        ctx.slot = slot
        ctx.give_slot = give_slot
        ctx.side_param, _Side.pending_param = _Side.pending_param, None      # linear(..., wgrad_side=True): the parameter whose gradient may go to the side stream
        ctx.grad_dst = getattr(weight, '_nnl_grad_dst', None)     # data parallel: the flat all-reduce bucket (dist.GradSync)
        ctx.uses = getattr(weight, '_nnl_uses', None)             # forward uses of this weight in the current step
        if ctx.uses is not None:
            ctx.uses[0] += 1
        xn = _pad_c4(to_nhwc(_f32c(x) if x.dim() != 4 else x.float()))
        wn = _pad_c4(to_nhwc(weight.float()))
        N, H, W, C = xn.shape
        K, R, S, _ = wn.shape
        g = _geom(N, H, W, C, K, R, S, stride, pad)
        y = torch.empty((N, g.P, g.Q, K), dtype=torch.float32, device=x.device)
        b = None if bias is None else _f32c(bias)
        wsb = int(lib.nnl_conv2d_fwd_workspace_bytes(g))         # balanced-schedule slabs (0 when the plain launch is used)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
        part, rows = None, _lib.i32(0)
        if bn_pivot is not None:                                  # BatchNorm statistics from the conv epilogue (include/nnl.h)
            part = torch.empty(((N * g.P * g.Q + 63) // 64) * K * 2, dtype=torch.float32, device=x.device)
        wmode = _wino_pref(wn.data_ptr(), 0, g)
        u = _WINO_U_FWD.get((wn.data_ptr(), wmode)) if wmode else None   # the filter prepared for this step, if any — in THIS call's layout
        if u is not None and u.numel() != _wino_u_numel(K, C, wmode):
            u = None
        check(lib.nnl_conv2d_fwd_pre(ptr(xn), ptr(wn), ptr(b), ptr(y), g, int(relu), ptr(ws), wsb, ptr(_tile_counters(x.device) if wsb else None),
                                     ptr(part), ptr(bn_pivot), ctypes.byref(rows) if part is not None else None, ptr(u), stream()))
        if part is None or rows.value == 0:
            part = torch.empty(0, dtype=torch.float32, device=x.device)
        else:
            part = part[:rows.value * K * 2]
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)              # no zero-filled gradient tensor for `part` in backward
        ctx.g, ctx.relu, ctx.has_bias = g, relu, bias is not None
        ctx.c_in = x.shape[1]
        ctx.w_layout = (tuple(weight.shape), tuple(weight.stride()))
        ctx.save_for_backward(xn, wn, y if relu else None)        # relu: 0 none, 1 ReLU, 2 sigmoid (both gates need the OUTPUT y)
        return from_nhwc(y), part

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        if dy is None:
            return (None,) * 9
        xn, wn, y = ctx.saved_tensors
        g = ctx.g
        pre = _padded_rows(dy, g.K) if (not ctx.relu and g.R == 1 and g.S == 1 and g.H == 1 and g.W == 1) else None
        if pre is not None and pre.shape[-1] != g.K:
            # the gradient arrived in a zero-padded buffer: run every pass on the padded K (zero channels contribute nothing)
            dyn = pre
            K = g.K
            padk = dyn.shape[-1] - K
            wn = torch.nn.functional.pad(wn, (0, 0, 0, 0, 0, 0, 0, padk))
            g = _geom(g.N, g.H, g.W, g.C, K + padk, g.R, g.S, g.stride, g.pad)
            return _Conv2d._backward_padded(ctx, dyn, wn, xn, g, K)
        dyn = to_nhwc(dy.float())
        db_gated = None
        if ctx.relu == 1 and _ab('NNL_RELU_GATE', '1') == '0':
            dyn = dyn * (y > 0)
        elif ctx.relu:
            # ReLU / sigmoid gate (+ the bias gradient of the gated dy) in one pass
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            rows = g.N * g.P * g.Q
            gated = torch.empty_like(dyn)
            cb = int(lib.nnl_colsum_workspace_bytes(rows, g.K)) if want_db else 0
            cws = torch.empty(max(cb // 4, 1), dtype=torch.float32, device=dyn.device) if want_db else None
            db_gated = torch.empty(g.K, dtype=torch.float32, device=dyn.device) if want_db else None
            check(lib.nnl_act_gate_colsum(ptr(dyn), ptr(y), ptr(gated), ptr(db_gated), rows, g.K, int(ctx.relu), ptr(cws), cb, stream()))
            dyn = gated
        K = g.K
        if K % 4 != 0:                      # e.g. RetinaNet 36/180-channel output convs: pad dy's channels
            dyn = _pad_c4(dyn)
            wn = torch.nn.functional.pad(wn, (0, 0, 0, 0, 0, 0, 0, dyn.shape[-1] - K))
            g = _geom(g.N, g.H, g.W, g.C, dyn.shape[-1], g.R, g.S, g.stride, g.pad)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            g_w, dyn_w, wn_w = g, dyn, wn                   # (the weight gradient keeps the unpadded operands)
            ktail = g.R == 1 and g.S == 1 and g.K % 4 == 0 and os.environ.get('NNL_IGEMM_KTAIL', '1') != '0'     # the tap kernel masks the k tail itself
            if g.K % 16 != 0 and g.K >= 32 and not ktail and _ab('NNL_DGRAD_PAD16', '1') != '0':
                # dgrad reduces over K: the tap-table kernel needs K % 16 == 0 (RetinaNet's 36- / 180-channel output convs would
                # fall back to the first-generation kernel, ~2.5x slower); zero channels cost one copy of dy
                padk = 16 - g.K % 16
                dyn = torch.nn.functional.pad(dyn, (0, padk))
                wkey = (wn.data_ptr(), tuple(wn.shape), padk)
                g = _geom(g.N, g.H, g.W, g.C, g.K + padk, g.R, g.S, g.stride, g.pad)
                wt = _WT_PADDED.get(wkey) if _WT_ACTIVE else None      # a shared filter (RetinaNet's output convs on five levels): padded + transposed ONCE per backward pass
                if wt is None:
                    wn = torch.nn.functional.pad(wn, (0, 0, 0, 0, 0, 0, 0, padk))
            else:
                wkey, wt = None, (_WT_ACTIVE.get(wn.data_ptr()) if g.K == K else None)
            if wt is None or tuple(wt.shape) != (g.C, g.R, g.S, g.K):
                wt = torch.empty((g.C, g.R, g.S, g.K), dtype=torch.float32, device=dyn.device)
                check(lib.nnl_conv2d_weight_transpose(ptr(wn), ptr(wt), g.K, g.R, g.S, g.C, stream()))
                if wkey is not None and _WT_ACTIVE:                   # (only inside the prepare_backward .. finish_backward window: weights are fixed there)
                    _WT_PADDED[wkey] = wt
            dxn = torch.empty((g.N, g.H, g.W, g.C), dtype=torch.float32, device=dyn.device)
            wsb = int(lib.nnl_conv2d_dgrad_workspace_bytes(g))
            dws = torch.empty(wsb // 4, dtype=torch.float32, device=dyn.device) if wsb else None
            shortcut = None
            if ctx.slot is not None:
                shortcut, ctx.slot.tensor, ctx.slot.closed = ctx.slot.tensor, None, True
            # stride 2: every output-parity class of a 3x3 / pad 1 filter has a tap, so every dx pixel passes through the epilogue
            fuse = shortcut is not None and g.K % 16 == 0 and shortcut.numel() == dxn.numel() \
                and (g.stride == 1 or (g.stride == 2 and g.R == 3 and g.S == 3 and g.pad == 1))
            wmode = _wino_pref(wn.data_ptr() if g.K == K else 0, 1, g)      # (a channel-padded dgrad transforms its own filter: not recorded under the weight)
            u = _WINO_U_BWD.get((wn.data_ptr(), wmode)) if (wmode and g.K == K) else None
            if u is not None and u.numel() != _wino_u_numel(g.C, g.K, wmode):
                u = None
            check(lib.nnl_conv2d_dgrad_pre(ptr(dyn), ptr(wt), ptr(dxn), g, ptr(shortcut) if fuse else None, ptr(dws), wsb,
                                           ptr(_tile_counters(dyn.device) if wsb else None), ptr(u), stream()))
            if shortcut is not None and not fuse:
                dxn += shortcut.view_as(dxn)
            give = ctx.give_slot
            if give is not None and not give.closed and ctx.c_in == g.C:
                give.tensor = dxn                           # the block's first conv adds it inside its dgrad kernel
            else:
                dx = from_nhwc(dxn[..., :ctx.c_in] if ctx.c_in != g.C else dxn)
            g, dyn, wn = g_w, dyn_w, wn_w
        elif ctx.slot is not None:
            ctx.slot.tensor, ctx.slot.closed = None, True
        if ctx.needs_input_grad[1]:
            dst = ctx.grad_dst
            # in place only for a weight used ONCE this step: the gradients of a shared weight (RetinaNet heads on 5 pyramid
            # levels) are summed by autograd and must not alias each other
            if dst is not None and ctx.uses is not None and ctx.uses[0] == 1 and dst.dim() == 4 and g.K == K and ctx.c_in == g.C and dst.permute(0, 2, 3, 1).is_contiguous() \
                    and tuple(dst.shape) == (g.K, g.C, g.R, g.S):
                dwn = dst.permute(0, 2, 3, 1)               # the bucket segment, viewed KRSC: the kernel writes it in place
            else:
                dwn = torch.empty((g.K, g.R, g.S, g.C), dtype=torch.float32, device=dyn.device)
            ws_bytes = int(lib.nnl_conv2d_wgrad_workspace_bytes(g))
            ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=dyn.device)
            run_w = lambda: check(lib.nnl_conv2d_wgrad(ptr(xn), ptr(dyn), ptr(dwn), g, ptr(ws), ws_bytes, stream()))
            if side_ok(getattr(ctx, 'side_param', None)) and dst is None and g.R == 1 and g.S == 1:
                dw = _side_wgrad(run_w, (xn, dyn, dwn, ws), dwn, K, ctx.c_in)
            else:
                run_w()
                dw = from_nhwc(dwn[:K, :, :, :ctx.c_in])
                # a 1x1 filter's [K, C, 1, 1] gradient: give it EXACTLY the parameter's strides (the size-1 dimensions make them ambiguous);
                # otherwise AccumulateGrad sees a layout mismatch and clones it into the parameter's layout — one device copy per 1x1
                # convolution and step (53 of RetinaNet's 68 rocclr_copyBuffer launches, profiles/r5_retinanet_kernel_stats.csv)
                wshape, wstride = getattr(ctx, 'w_layout', (None, None))
                if (g.R == 1 and g.S == 1 and wshape == tuple(dw.shape) and wstride != tuple(dw.stride()) and dw.is_contiguous(memory_format=torch.channels_last)
                        and wstride[1] == 1 and wstride[0] == dw.shape[1]):
                    dw = dw.as_strided(wshape, wstride)
        if db_gated is not None:
            db = db_gated
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db_full = torch.empty(g.K, dtype=torch.float32, device=dyn.device)
            cb = int(lib.nnl_colsum_workspace_bytes(g.N * g.P * g.Q, g.K))
            cws = torch.empty(max(cb // 4, 1), dtype=torch.float32, device=dyn.device)
            check(lib.nnl_colsum(ptr(dyn), ptr(db_full), g.N * g.P * g.Q, g.K, ptr(cws), cb, stream()))
            db = db_full[:K]
        return dx, dw, db, None, None, None, None, None, None


def _conv2d_backward_padded(ctx, dyn, wn, xn, g, K):
    """backward of a linear layer (1x1 "image") whose output gradient dyn [N,1,1,Kp] is already zero-padded from K to Kp = g.K
    channels: dgrad and wgrad on the padded K, bias gradient from the first K columns"""
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        wt = torch.empty((g.C, g.R, g.S, g.K), dtype=torch.float32, device=dyn.device)
        check(lib.nnl_conv2d_weight_transpose(ptr(wn), ptr(wt), g.K, g.R, g.S, g.C, stream()))
        dxn = torch.empty((g.N, g.H, g.W, g.C), dtype=torch.float32, device=dyn.device)
        wsb = int(lib.nnl_conv2d_dgrad_workspace_bytes(g))
        dws = torch.empty(wsb // 4, dtype=torch.float32, device=dyn.device) if wsb else None
        check(lib.nnl_conv2d_dgrad(ptr(dyn), ptr(wt), ptr(dxn), g, None, ptr(dws), wsb, ptr(_tile_counters(dyn.device) if wsb else None), stream()))
        dx = from_nhwc(dxn[..., :ctx.c_in] if ctx.c_in != g.C else dxn)
    if ctx.needs_input_grad[1]:
        dwn = torch.empty((g.K, g.R, g.S, g.C), dtype=torch.float32, device=dyn.device)
        ws_bytes = int(lib.nnl_conv2d_wgrad_workspace_bytes(g))
        ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=dyn.device)
        run_w = lambda: check(lib.nnl_conv2d_wgrad(ptr(xn), ptr(dyn), ptr(dwn), g, ptr(ws), ws_bytes, stream()))
        if side_ok(getattr(ctx, 'side_param', None)) and g.R == 1 and g.S == 1:
            dw = _side_wgrad(run_w, (xn, dyn, dwn, ws), dwn, K, ctx.c_in)
        else:
            run_w()
            dw = from_nhwc(dwn[:K, :, :, :ctx.c_in])
    if ctx.has_bias and ctx.needs_input_grad[2]:
        db_full = torch.empty(g.K, dtype=torch.float32, device=dyn.device)
        cb = int(lib.nnl_colsum_workspace_bytes(g.N * g.P * g.Q, g.K))
        cws = torch.empty(max(cb // 4, 1), dtype=torch.float32, device=dyn.device)
        check(lib.nnl_colsum(ptr(dyn), ptr(db_full), g.N * g.P * g.Q, g.K, ptr(cws), cb, stream()))
        db = db_full[:K]
    return dx, dw, db, None, None, None, None, None, None


_Conv2d._backward_padded = staticmethod(_conv2d_backward_padded)


class _NeedsView:
    "a Function ctx seen through another needs_input_grad (so that _Conv2d.backward can serve Functions with other input lists)"

    def __init__(self, ctx, needs):
        object.__setattr__(self, '_ctx', ctx)
        object.__setattr__(self, 'needs_input_grad', tuple(needs) + (False,) * 6)

    def __getattr__(self, name):
        return getattr(self._ctx, name)

    def __setattr__(self, name, value):
        setattr(self._ctx, name, value)


def _conv_backward_core(ctx, dy, needs):
    "(dx, dw, db) of a convolution whose ctx carries the fields _Conv2d.forward sets"
    out = _Conv2d.backward(_NeedsView(ctx, needs), dy)
    return out[0], out[1], out[2]


def conv2d(x, weight, bias=None, stride=1, pad=0, relu=False, grad_slot=None, give_slot=None):
    """y = act(conv2d(x, weight, bias, stride, padding=pad)); relu: False / 0 none, True / 1 ReLU, 2 sigmoid (fused into the
    kernel epilogue; the backward gate and the bias gradient are one pass); x logical [N,C,H,W], weight [K,C,R,S].
    grad_slot: see GradSlot (the shortcut gradient of a residual block, added to dx inside the dgrad kernel)."""
    return _Conv2d.apply(x, weight, bias, int(stride), int(pad), int(relu), grad_slot, None, give_slot)[0]


class _ConvAddUp2(torch.autograd.Function):
    """conv2d(x, weight, bias) + nearest-x2-upsample(small) in ONE kernel (reference PyramidFeatures.forward,
    retinanet.py:126-148: `P5_upsampled + P4_1(C4)`, `P3_1(C3) + P4_upsampled`); backward = the convolution's backward on dy plus
    the 2x2 block sums of dy for `small` (nnl_upsample2_bwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, small, stride, pad):
        require_cuda(x, weight, bias, small)
        ctx.slot = ctx.give_slot = None
        ctx.grad_dst = getattr(weight, '_nnl_grad_dst', None)
        ctx.uses = getattr(weight, '_nnl_uses', None)
        if ctx.uses is not None:
            ctx.uses[0] += 1
        xn = _pad_c4(to_nhwc(x.float()))
        wn = _pad_c4(to_nhwc(weight.float()))
        sn = to_nhwc(small.float())
        N, H, W, C = xn.shape
        K, R, S, _ = wn.shape
        g = _geom(N, H, W, C, K, R, S, stride, pad)
        if tuple(sn.shape) != (N, g.P // 2, g.Q // 2, K) or g.P % 2 or g.Q % 2:
            raise _lib.NnlError('conv_add_upsampled: `small` %s is not [N, K, P/2, Q/2] of the %dx%d output' % (tuple(small.shape), g.P, g.Q))
        y = torch.empty((N, g.P, g.Q, K), dtype=torch.float32, device=x.device)
        b = None if bias is None else _f32c(bias)
        check(lib.nnl_conv2d_fwd_add_up2(ptr(xn), ptr(wn), ptr(b), ptr(sn), ptr(y), g, stream()))
        ctx.g, ctx.relu, ctx.has_bias = g, 0, bias is not None
        ctx.c_in = x.shape[1]
        ctx.small_shape = (N, g.P // 2, g.Q // 2, K)
        ctx.save_for_backward(xn, wn, None)
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, dy):
        dsmall = None
        if ctx.needs_input_grad[3]:
            dyn = to_nhwc(dy.float())
            N, h, w, K = ctx.small_shape
            ds = torch.empty(ctx.small_shape, dtype=torch.float32, device=dy.device)
            check(lib.nnl_upsample2_bwd(ptr(dyn), ptr(ds), N, h, w, K, stream()))
            dsmall = from_nhwc(ds)
        need = ctx.needs_input_grad
        ctx_needs = (need[0], need[1], need[2])
        dx, dw, db = _conv_backward_core(ctx, dy, ctx_needs)
        return dx, dw, db, dsmall, None, None


def conv_add_upsampled(x, weight, bias, small, stride=1, pad=0):
    "conv2d(x, weight, bias) + F.interpolate(small, scale_factor=2, mode='nearest'), fused (FPN top-down merge)"
    return _ConvAddUp2.apply(x, weight, bias, small, int(stride), int(pad))


def conv2d_with_bn_stats(x, weight, bias, stride, pad, bn_pivot, grad_slot=None, give_slot=None):
    """conv2d whose epilogue also reduces the BatchNorm batch statistics of its output against `bn_pivot` [K].  Returns (y, partials);
    partials is empty when this launch could not produce them (the BatchNorm then runs its own statistics pass)."""
    return _Conv2d.apply(x, weight, bias, int(stride), int(pad), False, grad_slot, bn_pivot, give_slot)


def _rows_in_place(x):
    "a 2-D fp32 matrix whose rows are dense (any row stride): usable by the kernels that take a leading dimension"
    return x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1]


class _LinearSmall(torch.autograd.Function):
    """nn.Linear with 1 - 4 output features (FullyConnectedNet.final_lin of the regression heads, General/Layers.py:146): one wave
    per row forward, fixed-order column reductions backward (csrc/linear_small.hip) instead of a 64-wide MFMA tile per column."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        require_cuda(x, weight, bias)
        x = x if _rows_in_place(x) else _f32c(x)
        w = _f32c(weight)
        b = None if bias is None else _f32c(bias)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        check(lib.nnl_linear_small_fwd(ptr(x), ptr(w), ptr(b), ptr(y), M, K, x.stride(0) if M > 1 else K, N, stream()))
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dy = _f32c(dy)
        dev = dy.device
        dx = torch.empty(M, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dw = torch.empty(N, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        db = torch.empty(N, dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        wsb = int(lib.nnl_linear_small_bwd_workspace_bytes(M, K, N)) if (dw is not None or db is not None) else 0
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=dev) if wsb else None
        check(lib.nnl_linear_small_bwd(ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(dw), ptr(db), M, K, x.stride(0) if M > 1 else K, N, ptr(ws), wsb,
                                       stream()))
        return dx, dw, db


def linear(x, weight, bias=None, relu=False, wgrad_side=False):
    """y = x @ weight.T + bias [+ ReLU] (nn.Linear; reference General/Layers.py:39,146; Text.py:572) on the same
    fp32-MFMA implicit-GEMM kernels: a Linear is the 1x1 convolution of a 1x1 'image' per sample.  Leading dims of x
    are flattened into rows.  wgrad_side: the weight gradient may be computed on the side stream (see _Side above)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if (not relu and weight.shape[0] <= 4 and x2.is_cuda and x2.shape[0] > 0
            and _ab('NNL_LINEAR_SMALL', '1') != '0'):
        return _LinearSmall.apply(x2, weight, bias).reshape(*lead, weight.shape[0])
    _Side.pending_param = weight if (wgrad_side and weight.requires_grad) else None
    y = _Conv2d.apply(x2[:, :, None, None], weight[:, :, None, None], bias, 1, 0, int(relu), None, None)[0]
    return y.reshape(*lead, weight.shape[0])


def _rows_view(x):
    """logical [N,C,H,W] (or [rows,C]) -> contiguous [rows, C] matrix + a function mapping such a matrix back."""
    if x.dim() == 4:
        xn = to_nhwc(x.float())
        N, H, W, C = xn.shape
        return xn.view(-1, C), (lambda m: from_nhwc(m.view(N, H, W, C)))
    if x.dim() == 2:
        return _f32c(x), (lambda m: m)
    if x.dim() == 3:                                    # BatchNorm1d on [N, C, L]
        xn = x.float().permute(0, 2, 1).contiguous()
        N, L, C = xn.shape
        return xn.view(-1, C), (lambda m: m.view(N, L, C).permute(0, 2, 1))
    raise ValueError('bn_act: unsupported input rank %d' % x.dim())


def _relu_mask(rows, C, device):
    "uninitialised keep-bit buffer for nnl_bn_fwd (1 bit per element, + 2 words of slack)"
    return torch.empty((rows * C + 31) // 32 + 2, dtype=torch.int32, device=device)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, training, momentum, eps, relu, nbt, slot=None, ext=None,
                pivot_out=None):
        require_cuda(x, residual, gamma, beta)
        ctx.slot = slot
        xm, back = _rows_view(x)
        rows, C = xm.shape
        rm = None if residual is None else _rows_view(residual)[0]
        y = torch.empty_like(xm)
        mean = torch.empty(C, dtype=torch.float32, device=xm.device)
        invstd = torch.empty(C, dtype=torch.float32, device=xm.device)
        wsb = int(lib.nnl_bn_workspace_bytes(rows, C))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=xm.device)
        mask = _relu_mask(rows, C, xm.device) if relu else None      # 1 bit / element for the backward's ReLU gate
        check(lib.nnl_bn_fwd(ptr(xm), ptr(gamma), ptr(beta), ptr(rm), ptr(y), ptr(mean), ptr(invstd), ptr(running_mean),
                             ptr(running_var), rows, C, float(eps), float(momentum), int(training), int(relu), ptr(nbt),
                             ptr(mask), ptr(ext[0]) if ext else None, (ext[0].numel() // (2 * C)) if ext else 0,
                             ptr(ext[1]) if ext else None, ptr(pivot_out), ptr(ws), wsb, stream()))
        ctx.save_for_backward(xm, mask, gamma, mean, invstd)
        ctx.cfg = (training, relu, residual is not None, back)
        return back(y)

    @staticmethod
    def backward(ctx, dy):
        xm, mask, gamma, mean, invstd = ctx.saved_tensors
        training, relu, has_res, back = ctx.cfg
        dym = _rows_view(dy)[0]
        rows, C = xm.shape
        dx = torch.empty_like(xm)
        dres = torch.empty_like(xm) if (has_res and ctx.needs_input_grad[1]) else None
        dgamma = torch.empty(C, dtype=torch.float32, device=xm.device) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=xm.device) if gamma is not None else None
        wsb = int(lib.nnl_bn_workspace_bytes(rows, C))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=xm.device)
        check(lib.nnl_bn_bwd(ptr(dym), None, ptr(mask), ptr(xm), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx), ptr(dres),
                             ptr(dgamma), ptr(dbeta), rows, C, int(training), int(relu), ptr(ws), wsb, stream()))
        if ctx.slot is not None and dres is not None:
            ctx.slot.tensor, dres = dres, None              # the block's first conv adds it to its dx (GradSlot)
        return (back(dx), None if dres is None else back(dres), dgamma, dbeta) + (None,) * 10


def _sync_group_size(group):
    import torch.distributed as dist
    return dist.get_world_size(group)


class _SyncBNAct(torch.autograd.Function):
    """Training-mode BatchNorm with statistics over the GLOBAL batch of a data-parallel job: two kernels +
    one small all_gather forward (2C+2 floats per rank), two kernels + one all_reduce backward (2C floats).  `group` is a
    torch.distributed process group (None = WORLD); gather_fn / reduce_fn are injectable for single-process tests."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, nbt, group, comm, slot=None):
        require_cuda(x, residual, gamma, beta)
        ctx.slot = slot
        xm, back = _rows_view(x)
        rows, C = xm.shape
        rm = None if residual is None else _rows_view(residual)[0]
        dev = xm.device
        wsb = int(lib.nnl_bn_workspace_bytes(rows, C))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
        stats = torch.empty(2 * C + 2, dtype=torch.float32, device=dev)
        check(lib.nnl_bn_sync_stats(ptr(xm), ptr(stats), rows, C, ptr(ws), wsb, stream()))
        all_stats = comm.all_gather(stats, group)                  # [world, 2C+2], rank order
        world = all_stats.shape[0]
        y = torch.empty_like(xm)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        mask = _relu_mask(rows, C, dev) if relu else None
        check(lib.nnl_bn_sync_fwd(ptr(xm), ptr(all_stats), world, ptr(gamma), ptr(beta), ptr(rm), ptr(y), ptr(mean), ptr(invstd),
                                  ptr(running_mean), ptr(running_var), rows, C, float(eps), float(momentum), int(relu), ptr(nbt),
                                  ptr(mask), ptr(ws), wsb, stream()))
        ctx.save_for_backward(xm, mask, gamma, mean, invstd, all_stats)
        ctx.cfg = (relu, residual is not None, back, group, comm)
        return back(y)

    @staticmethod
    def backward(ctx, dy):
        xm, mask, gamma, mean, invstd, all_stats = ctx.saved_tensors
        relu, has_res, back, group, comm = ctx.cfg
        dym = _rows_view(dy)[0]
        rows, C = xm.shape
        dev = xm.device
        wsb = int(lib.nnl_bn_workspace_bytes(rows, C))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
        sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
        check(lib.nnl_bn_sync_bwd_reduce(ptr(dym), None, ptr(mask), ptr(xm), ptr(mean), ptr(invstd), ptr(sums), rows, C, int(relu),
                                         ptr(ws), wsb, stream()))
        total = comm.all_reduce_sum(sums, group)                   # a new tensor; `sums` keeps the local values
        dx = torch.empty_like(xm)
        dres = torch.empty_like(xm) if (has_res and ctx.needs_input_grad[1]) else None
        dgamma = torch.empty(C, dtype=torch.float32, device=dev) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=dev) if gamma is not None else None
        check(lib.nnl_bn_sync_bwd(ptr(dym), None, ptr(mask), ptr(xm), ptr(gamma), ptr(mean), ptr(invstd), ptr(sums), ptr(total),
                                  ptr(all_stats), int(all_stats.shape[0]), ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), rows, C,
                                  int(relu), ptr(ws), wsb, stream()))
        if ctx.slot is not None and dres is not None:
            ctx.slot.tensor, dres = dres, None
        return (back(dx), None if dres is None else back(dres), dgamma, dbeta) + (None,) * 9


class DistComm:
    "the two collectives SyncBN needs, over torch.distributed (RCCL on the GPUs of a node; gloo in tests)"

    @staticmethod
    def all_gather(t, group):
        import torch.distributed as dist
        w = dist.get_world_size(group)
        out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather(list(out.unbind(0)), t, group=group)
        return out

    @staticmethod
    def all_reduce_sum(t, group):
        import torch.distributed as dist
        out = t.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out


def bn_act(bn, x, residual=None, relu=True, grad_slot=None, ext_stats=None, pivot_out=None):
    """BatchNorm (train: batch statistics + running-stat update; eval: running stats) -> (+ residual) -> ReLU in the HIP
    kernels of batchnorm.hip: the bn -> `out += residual` -> relu tail of BasicBlock / Bottleneck (reference
    retinanet.py:47-48,53-57,81-95), the stem (:372-373) and the BatchNorm1d layers (General/Layers.py:40).
    `bn` is the nn.BatchNorm{1,2}d module holding weight / bias / running stats (state_dict unchanged).  A module marked by
    dist.enable_sync_bn (`bn.nnl_sync = (group, comm)`) uses global-batch statistics in training mode."""
    training = bn.training or (bn.running_mean is None)
    momentum, nbt = 0.0, None
    if bn.training and bn.track_running_stats:
        nbt = bn.num_batches_tracked                       # incremented inside the finalize kernel
        if bn.momentum is not None:
            momentum = bn.momentum
        else:                                              # cumulative moving average: the factor needs the count on the host
            nbt.add_(1)
            momentum, nbt = 1.0 / float(nbt.item()), None
    rmean = bn.running_mean if (not training or bn.track_running_stats) else None
    rvar = bn.running_var if (not training or bn.track_running_stats) else None
    sync = getattr(bn, 'nnl_sync', None)
    if sync is not None and training:
        return _SyncBNAct.apply(x, residual, bn.weight, bn.bias, rmean, rvar, momentum, bn.eps, relu, nbt, sync[0], sync[1],
                                grad_slot)
    ext = ext_stats if (ext_stats is not None and training and ext_stats[0].numel() > 0) else None
    return _BNAct.apply(x, residual, bn.weight, bn.bias, rmean, rvar, training, momentum, bn.eps, relu, nbt, grad_slot, ext,
                        pivot_out if training else None)


class _ConcatPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        xn = to_nhwc(x.float())
        N, H, W, C = xn.shape
        out = torch.empty(N, 2 * C, dtype=torch.float32, device=x.device)
        am = torch.empty(N, C, dtype=torch.int32, device=x.device)
        check(lib.nnl_concat_pool_fwd(ptr(xn), ptr(out), ptr(am), N, H * W, C, stream()))
        ctx.save_for_backward(am)
        ctx.shape = (N, H, W, C)
        return out.view(N, 2 * C, 1, 1)

    @staticmethod
    def backward(ctx, dout):
        (am,) = ctx.saved_tensors
        N, H, W, C = ctx.shape
        dx = torch.empty(N, H, W, C, dtype=torch.float32, device=dout.device)
        check(lib.nnl_concat_pool_bwd(ptr(_f32c(dout.reshape(N, 2 * C))), ptr(am), ptr(dx), N, H * W, C, stream()))
        return from_nhwc(dx)


def conv_bn_act(conv, bn, x, residual=None, relu=True, conv_slot=None, bn_slot=None, conv_give=None):
    """bn_act(bn, conv(x), residual, relu) for a HipConv2d followed by BatchNorm — the conv -> bn -> (+shortcut) -> relu unit of the
    ResNet blocks (retinanet.py:43-59,77-97,304-306).  In training mode the convolution's epilogue reduces the batch statistics
    as shifted sums sum(y - pivot), sum((y - pivot)^2), so the BatchNorm does not re-read the activation for them.  The pivot is
    the batch mean of the PREVIOUS training step of this layer (kept in `bn._nnl_pivot`, written by the finalize kernel): it
    sits within a fraction of a standard deviation of the new mean, which keeps the variance free of cancellation; the first
    step (no pivot yet) runs the stand-alone statistics pass."""
    fuse = (bn.training and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None
            and getattr(bn, 'nnl_sync', None) is None and x.is_cuda and conv.bias is None and not conv.fuse_relu
            and torch.is_grad_enabled() and os.environ.get('NNL_BN_EPI_STATS', '1') != '0')
    if not fuse:
        return bn_act(bn, conv(x, grad_slot=conv_slot, give_slot=conv_give), residual=residual, relu=relu, grad_slot=bn_slot)
    pivot = getattr(bn, '_nnl_pivot', None)
    if pivot is None or pivot.device != x.device or pivot.numel() != bn.num_features:
        if torch.cuda.is_current_stream_capturing():
            return bn_act(bn, conv(x, grad_slot=conv_slot, give_slot=conv_give), residual=residual, relu=relu, grad_slot=bn_slot)
        pivot = torch.empty(bn.num_features, dtype=torch.float32, device=x.device)
        object.__setattr__(bn, '_nnl_pivot', pivot)                  # plain attribute: not a buffer, not in the state_dict
        return bn_act(bn, conv(x, grad_slot=conv_slot, give_slot=conv_give), residual=residual, relu=relu, grad_slot=bn_slot, pivot_out=pivot)
    y, part = conv2d_with_bn_stats(x, conv.weight, None, conv.stride[0], conv.padding[0], pivot, conv_slot, conv_give)
    return bn_act(bn, y, residual=residual, relu=relu, grad_slot=bn_slot, ext_stats=(part, pivot), pivot_out=pivot)


def linear_relu_bn(lin, bn, x):
    """bn(relu(lin(x))) — the Linear block of FullyConnectedNet (reference General/Layers.py:37-41: Linear -> ReLU -> BatchNorm1d, BN
    AFTER the ReLU).  In training mode the GEMM's epilogue reduces the batch statistics of relu(x W^T + b) as shifted sums (as
    conv_bn_act does for convolutions), so the BatchNorm1d does not re-read the activation for them."""
    fuse = (bn.training and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None
            and getattr(bn, 'nnl_sync', None) is None and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled()
            and os.environ.get('NNL_BN_EPI_STATS', '1') != '0')
    pivot = getattr(bn, '_nnl_pivot', None) if fuse else None
    if not fuse or pivot is None or pivot.device != x.device or pivot.numel() != bn.num_features:
        y = linear(x, lin.weight, lin.bias, relu=True)
        if not fuse or torch.cuda.is_current_stream_capturing():
            return bn_act(bn, y, relu=False)
        pivot = torch.empty(bn.num_features, dtype=torch.float32, device=x.device)
        object.__setattr__(bn, '_nnl_pivot', pivot)                  # plain attribute: not a buffer, not in the state_dict
        return bn_act(bn, y, relu=False, pivot_out=pivot)
    y, part = _Conv2d.apply(x[:, :, None, None], lin.weight[:, :, None, None], lin.bias, 1, 0, 1, None, pivot, None)
    return bn_act(bn, y.reshape(x.shape[0], lin.weight.shape[0]), relu=False, ext_stats=(part, pivot), pivot_out=pivot)


def concat_pool2d(x):
    """cat([AdaptiveMaxPool2d(1), AdaptiveAvgPool2d(1)], 1) -> [N, 2C, 1, 1]  (AdaptiveConcatPool2d, General/Layers.py:78-87);
    the max gradient goes to the first arg-max pixel, as torch's adaptive max pool does (pool.hip)."""
    return _ConcatPool.apply(x)


class _MaxPool2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ksize, stride, pad):
        require_cuda(x)
        xn = to_nhwc(x.float())
        N, H, W, C = xn.shape
        P, Q = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        y = torch.empty(N, P, Q, C, dtype=torch.float32, device=x.device)
        idx = torch.empty(N, P, Q, C, dtype=torch.uint8, device=x.device)
        check(lib.nnl_maxpool2d_fwd(ptr(xn), ptr(y), ptr(idx), N, H, W, C, P, Q, ksize, stride, pad, stream()))
        ctx.save_for_backward(idx)
        ctx.cfg = (N, H, W, C, P, Q, ksize, stride, pad)
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C, P, Q, ksize, stride, pad = ctx.cfg
        dyn = to_nhwc(dy.float())
        dx = torch.empty(N, H, W, C, dtype=torch.float32, device=dy.device)
        check(lib.nnl_maxpool2d_bwd(ptr(dyn), ptr(idx), ptr(dx), N, H, W, C, P, Q, ksize, stride, pad, stream()))
        return from_nhwc(dx), None, None, None


class _BNReLUMaxPool(torch.autograd.Function):
    """maxpool(relu(batch_norm(x))) — the ResNet stem tail (reference retinanet.py:372-374) — without materialising the
    normalised activation: batchnorm.hip nnl_bn_relu_maxpool_{fwd,bwd}.  Saves x (the conv output), the uint8 window index and
    the per-channel affine."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, nbt, ksize, stride, pad):
        require_cuda(x, gamma, beta)
        xn = to_nhwc(x.float())
        N, H, W, C = xn.shape
        P, Q = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        dev = x.device
        y = torch.empty(N, P, Q, C, dtype=torch.float32, device=dev)
        idx = torch.empty(N, P, Q, C, dtype=torch.uint8, device=dev)
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)               # mean, invstd, scale, shift
        wsb = int(lib.nnl_bn_workspace_bytes(N * H * W, C))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
        check(lib.nnl_bn_relu_maxpool_fwd(ptr(xn), ptr(gamma), ptr(beta), ptr(y), ptr(idx), ptr(stats[0]), ptr(stats[1]),
                                          ptr(stats[2]), ptr(stats[3]), ptr(running_mean), ptr(running_var), N, H, W, C, P, Q,
                                          ksize, stride, pad, float(eps), float(momentum), int(training), ptr(nbt), ptr(ws), wsb,
                                          stream()))
        ctx.save_for_backward(xn, idx, gamma, beta, stats, y)
        ctx.cfg = (N, H, W, C, P, Q, ksize, stride, pad, training)
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, dy):
        xn, idx, gamma, beta, stats, y = ctx.saved_tensors
        N, H, W, C, P, Q, ksize, stride, pad, training = ctx.cfg
        dyn = to_nhwc(dy.float())
        dx = torch.empty_like(xn)
        dgamma = torch.empty(C, dtype=torch.float32, device=xn.device) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=xn.device) if gamma is not None else None
        wsb = int(lib.nnl_bn_workspace_bytes(N * H * W, C))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=xn.device)
        check(lib.nnl_bn_relu_maxpool_bwd(ptr(dyn), ptr(y), ptr(idx), ptr(xn), ptr(gamma), ptr(beta), ptr(stats[0]), ptr(stats[1]),
                                          ptr(stats[2]), ptr(stats[3]), ptr(dx), ptr(dgamma), ptr(dbeta), N, H, W, C, P, Q, ksize, stride, pad,
                                          int(training), ptr(ws), wsb, stream()))
        return (from_nhwc(dx), dgamma, dbeta) + (None,) * 9


def conv_bn_relu_maxpool(conv, bn, pool, x):
    """pool(relu(bn(conv(x)))) for the ResNet stem (conv7x7/2 -> BatchNorm -> ReLU -> MaxPool 3/2/1; reference
    retinanet.py:304-307,371-374).  BatchNorm, ReLU and the pooling run as one pass over the convolution output whenever the
    pooling is the plain floor-mode square max-pool and the BatchNorm is a local (non-synchronised) one; otherwise the three
    stages run one after the other."""
    k, s, p = pool.kernel_size, pool.stride, pool.padding
    fuse = (x.is_cuda and isinstance(pool, torch.nn.MaxPool2d) and all(isinstance(v, int) for v in (k, s, p))
            and pool.dilation == 1 and not pool.ceil_mode and not pool.return_indices and 2 * p <= k and k * k <= 255
            and not (bn.training and getattr(bn, 'nnl_sync', None) is not None)
            and (bn.running_mean is not None or bn.training)
            and not (bn.training and bn.track_running_stats and bn.momentum is None)
            and bool(lib.nnl_bn_relu_maxpool_supported(bn.num_features)))
    if not fuse:
        return pool(conv_bn_act(conv, bn, x, relu=True))
    y = conv(x)
    training = bn.training or (bn.running_mean is None)
    momentum, nbt = 0.0, None
    if bn.training and bn.track_running_stats:
        nbt, momentum = bn.num_batches_tracked, bn.momentum
    rmean = bn.running_mean if (not training or bn.track_running_stats) else None
    rvar = bn.running_var if (not training or bn.track_running_stats) else None
    return _BNReLUMaxPool.apply(y, bn.weight, bn.bias, rmean, rvar, training, momentum, bn.eps, nbt, k, s, p)


def maxpool2d(x, ksize=3, stride=2, pad=1):
    """nn.MaxPool2d(ksize, stride, pad) (floor mode, no dilation) — the ResNet stem pool (retinanet.py:307,374) — NHWC,
    channel count a multiple of 4; torch's first-maximum tie rule, deterministic gather-style backward (pool.hip)."""
    return _MaxPool2d.apply(x, int(ksize), int(stride), int(pad))


# ---------------------------------------------------------------------------------------------------------
# K3 tabular front end
# ---------------------------------------------------------------------------------------------------------
class TabularPlan:
    """Device-side descriptor arrays for a list of embedding tables (see include/nnl.h, K3).  Rebuilt when a table
    moves (data_ptr / device change)."""

    def __init__(self, weights):
        import numpy as np
        dev = weights[0].device
        self.key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights) + (str(dev),)
        self.sync = None
        card = [int(w.shape[0]) for w in weights]
        dim = [int(w.shape[1]) for w in weights]
        self.ncat, self.cat_width, self.total_rows = len(weights), sum(dim), sum(card)
        col_off = np.concatenate([[0], np.cumsum(dim)[:-1]]).astype(np.int32)
        row_off = np.concatenate([[0], np.cumsum(card)[:-1]]).astype(np.int32)
        sizes = [c * d for c, d in zip(card, dim)]
        self.grad_sizes, self.shapes = sizes, [tuple(w.shape) for w in weights]
        grad_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.grad_elems = int(sum(sizes))
        t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(dev)
        self.ptrs = t(np.array([w.data_ptr() for w in weights], dtype=np.int64), torch.int64)
        self.card, self.dim = t(card, torch.int32), t(dim, torch.int32)
        self.col_off, self.row_off, self.grad_off = t(col_off, torch.int32), t(row_off, torch.int32), t(grad_off, torch.int64)
        self.col_table = t(np.repeat(np.arange(self.ncat), dim), torch.int32)
        self.row_table = t(np.repeat(np.arange(self.ncat), card), torch.int32)
        self.flags = torch.zeros(self.total_rows, dtype=torch.int32, device=dev)
        # sort-free backward (nnl_tab_scan_bwd): 256 flat gradient elements of ONE column per block
        self.max_dim = max(dim) if dim else 0
        blk_col, blk_first = [], []
        for j, n in enumerate(sizes):
            for f in range(0, n, 256):
                blk_col.append(j)
                blk_first.append(f)
        self.n_scan_blocks = len(blk_col)
        self.blk_col, self.blk_first = t(blk_col or [0], torch.int32), t(blk_first or [0], torch.int32)

    @staticmethod
    def for_weights(weights, cached=None):
        key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights) + (str(weights[0].device),)
        return cached if (cached is not None and cached.key == key) else TabularPlan(weights)


class _TabEmbedConcat(torch.autograd.Function):
    """StructuredDataNet.forward front end (reference Applications/StructuredData.py:1074-1082 + General/Layers.py:74-76)."""

    @staticmethod
    def forward(ctx, xcat, cont, row_mask, cont_mask, plan, max_norm, *weights):
        require_cuda(xcat, cont, *weights)
        for w in weights:
            if not (w.is_contiguous() and w.dtype == torch.float32):
                raise _lib.NnlError('tab_embed_concat: embedding tables must be contiguous fp32 parameters')
        xcat = xcat.contiguous().long()
        bs = xcat.shape[0]
        n_cont = 0 if cont is None else cont.shape[1]
        cont = None if cont is None else _f32c(cont)
        row_mask = None if row_mask is None else _f32c(row_mask)
        cont_mask = None if cont_mask is None else _f32c(cont_mask)
        if max_norm is not None:
            # data parallel: the in-place renorm must hit the rows looked up by ANY rank, or the replicas' tables diverge
            # (Learner's replayed data-parallel step gathers them BEFORE the graph — StructuredDataNet.nnl_dp_prepare — and hands the
            # static buffer over as a fourth element: no collective inside the captured forward)
            if plan.sync is None:
                xren = xcat
            elif len(plan.sync) > 3 and plan.sync[3] is not None:
                xren = plan.sync[3]
            else:
                xren = _global_lookup_indices(xcat, plan.sync[:3])
            check(lib.nnl_tab_renorm(ptr(xren), ptr(plan.ptrs), ptr(plan.card), ptr(plan.dim), ptr(plan.row_off),
                                     ptr(plan.row_table), ptr(plan.flags), xren.shape[0], plan.ncat, plan.total_rows,
                                     float(max_norm), ptr(index_error_flag(xcat.device)), stream()))
        ld = plan.cat_width + n_cont
        out = torch.empty(bs, ld, dtype=torch.float32, device=xcat.device)
        check(lib.nnl_tab_gather_fwd(ptr(xcat), ptr(plan.ptrs), ptr(plan.card), ptr(plan.dim), ptr(plan.col_off),
                                     ptr(plan.col_table), ptr(row_mask), ptr(cont), ptr(cont_mask), ptr(out), bs, plan.ncat,
                                     plan.cat_width, n_cont, ld, stream()))
        ctx.save_for_backward(xcat, row_mask, cont_mask)
        ctx.plan, ctx.n_cont = plan, n_cont
        return out

    @staticmethod
    def backward(ctx, dout):
        xcat, row_mask, cont_mask = ctx.saved_tensors
        plan, n_cont = ctx.plan, ctx.n_cont
        bs = xcat.shape[0]
        flat = torch.empty(max(plan.grad_elems, 1), dtype=torch.float32, device=dout.device)
        dcont = torch.empty(bs, n_cont, dtype=torch.float32, device=dout.device) if (n_cont and ctx.needs_input_grad[1]) else None
        # (the scan costs gradient elements x minibatch compares — Rossmann: 7e4 x 1024; tables of 1e5+ rows go to the sorted scatter)
        if plan.ncat and bs and 0 < plan.max_dim <= 32 and plan.grad_elems * bs <= (1 << 28) and os.environ.get('NNL_TAB_SCAN', '1') != '0':
            # one launch, no sort, no zero fill; a row-strided gradient (the slice of a channel-padded buffer) is read in place
            if not (dout.dtype == torch.float32 and dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= dout.shape[1]):
                dout = _f32c(dout)
            check(lib.nnl_tab_scan_bwd(ptr(xcat), ptr(plan.card), ptr(plan.dim), ptr(plan.col_off), ptr(plan.grad_off), ptr(row_mask),
                                       ptr(cont_mask), ptr(dout), ptr(flat), ptr(dcont), ptr(plan.blk_col), ptr(plan.blk_first),
                                       plan.n_scan_blocks, plan.max_dim, bs, plan.ncat, plan.cat_width, n_cont, dout.stride(0), stream()))
        else:
            dout = _f32c(dout)
            wsb = int(lib.nnl_tab_scatter_bwd_workspace_bytes(bs, plan.ncat))            # sample-order (deterministic) scatter-add
            ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dout.device)
            check(lib.nnl_tab_scatter_bwd(ptr(xcat), ptr(plan.card), ptr(plan.dim), ptr(plan.col_off), ptr(plan.col_table),
                                          ptr(plan.grad_off), ptr(row_mask), ptr(cont_mask), ptr(dout), ptr(flat), plan.grad_elems,
                                          ptr(dcont), bs, plan.ncat, plan.cat_width, n_cont, dout.shape[1], ptr(ws), wsb, stream()))
        grads, o = [], 0
        for n, shp in zip(plan.grad_sizes, plan.shapes):
            grads.append(flat[o:o + n].view(shp))
            o += n
        return (None, dcont, None, None, None, None) + tuple(grads)


def _global_lookup_indices(xcat, sync):
    """All ranks' categorical indices of this step ([sum of local batch sizes (padded), ncat]) — SURVEY.md §8e: "all-gather of
    touched embedding indices so max_norm renorm hits the same rows on every rank".  sync = (group, comm, capacity): shards
    are padded to `capacity` rows (the per-rank full batch size) so the collective has a fixed shape and needs no host
    synchronisation; padding rows are replaced by a real looked-up row (renorm is idempotent per row)."""
    group, comm, cap = sync[:3]
    bs, ncat = xcat.shape
    if bs > cap:
        raise _lib.NnlError(f'tab_embed_concat: local batch {bs} exceeds the data-parallel capacity {cap}')
    pad = torch.zeros(cap + 1, ncat, dtype=torch.int64, device=xcat.device)
    pad[:bs] = xcat
    pad[cap, 0] = bs
    g = comm.all_gather(pad, group)                                   # [world, cap + 1, ncat]
    counts, rows = g[:, cap, 0], g[:, :cap]
    valid = torch.arange(cap, device=xcat.device)[None, :] < counts[:, None]
    flat = rows.reshape(-1, ncat)
    fill = flat[valid.reshape(-1).to(torch.uint8).argmax()]             # first real row of any rank
    return torch.where(valid.reshape(-1, 1), flat, fill).contiguous()


def keep_masks(shape_a, keep_a, shape_b, keep_b, device):
    """Two scaled Bernoulli keep masks (value 0 or 1/keep) from one uniform draw and one launch; a shape may be None."""
    na = 0 if shape_a is None else int(torch.Size(shape_a).numel())
    nb = 0 if shape_b is None else int(torch.Size(shape_b).numel())
    u = torch.rand(na + nb, dtype=torch.float32, device=device)
    a = torch.empty(shape_a, dtype=torch.float32, device=device) if na else None
    b = torch.empty(shape_b, dtype=torch.float32, device=device) if nb else None
    check(lib.nnl_keep_masks(ptr(u), ptr(a), na, float(keep_a), ptr(b), nb, float(keep_b), stream()))
    return a, b


def tab_embed_concat(xcat, weights, row_mask=None, cont=None, cont_mask=None, max_norm=None, plan=None, sync=None):
    """[bs, sum(d_j) + n_cont] = cat_j( W_j[xcat[:,j]] * row_mask[j][:,None] ) ++ cont*cont_mask, after the in-place
    max_norm renorm of the looked-up rows (of all ranks' lookups when `sync` = (group, comm, capacity) is given).
    Returns (out, plan)."""
    plan = TabularPlan.for_weights(weights, plan)
    plan.sync = sync
    return _TabEmbedConcat.apply(xcat, cont, row_mask, cont_mask, plan, max_norm, *weights), plan


def embedding_renorm_drop(x, weight, mask, max_norm):
    """EmbeddingDrop.forward for ONE column (General/Layers.py:74-76): emb(x) * mask.unsqueeze(1)."""
    out, _ = tab_embed_concat(x.view(-1, 1), [weight], None if mask is None else mask.view(1, -1), None, None, max_norm)
    return out


# ---------------------------------------------------------------------------------------------------------
# K6 fused RetinaNet loss
# ---------------------------------------------------------------------------------------------------------
class _RetinaLoss(torch.autograd.Function):
    """SSD_loss.__call__ (reference Applications/Vision.py:1620-1644 and everything below it) as one fused forward
    kernel and one fused backward kernel; returns [total, reg_loss, clas_loss].  Gradient flows through `total`."""

    @staticmethod
    def forward(ctx, anchors, reg, clas, boxes, cats, beta, alpha, gamma):
        require_cuda(anchors, reg, clas, boxes, cats)
        anchors, reg, clas, boxes = _f32c(anchors), _f32c(reg), _f32c(clas), _f32c(boxes)
        cats = cats.contiguous().long()
        bs, A, K = clas.shape
        M = boxes.shape[1]
        dev = clas.device
        state = torch.empty(bs, A, dtype=torch.int32, device=dev)
        npos = torch.empty(bs, dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        wsb = int(lib.nnl_retina_loss_workspace_bytes(bs, A))
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=dev)
        check(lib.nnl_retina_loss_fwd(ptr(anchors), ptr(reg), ptr(clas), ptr(boxes), ptr(cats), ptr(state), ptr(npos), ptr(out),
                                      bs, A, K, M, float(beta), float(alpha), float(gamma), ptr(ws), wsb, stream()))
        ctx.save_for_backward(anchors, reg, clas, boxes, cats, state, npos)
        ctx.hyper = (float(beta), float(alpha), float(gamma))
        return out

    @staticmethod
    def backward(ctx, dout):
        anchors, reg, clas, boxes, cats, state, npos = ctx.saved_tensors
        beta, alpha, gamma = ctx.hyper
        bs, A, K = clas.shape
        gup = _f32c(dout)[0:1].contiguous()
        dreg, dclas = torch.empty_like(reg), torch.empty_like(clas)
        check(lib.nnl_retina_loss_bwd(ptr(anchors), ptr(reg), ptr(clas), ptr(boxes), ptr(cats), ptr(state), ptr(npos), ptr(gup),
                                      ptr(dreg), ptr(dclas), bs, A, K, boxes.shape[1], beta, alpha, gamma, stream()))
        return None, dreg, dclas, None, None, None, None, None


def retina_loss(anchors, reg, clas, boxes, cats, beta=0.5, alpha=0.25, gamma=2.0):
    """[ (1-beta)*reg_loss + beta*clas_loss, reg_loss, clas_loss ] for a batch (see include/nnl.h, K6)."""
    return _RetinaLoss.apply(anchors, reg, clas, boxes, cats, beta, alpha, gamma)


from .ops_text import (lstm_layer, embedding_rowmask, softmax_cross_entropy, cross_entropy_nd, seq_activation_reg)  # noqa: E402,F401
__all__ += ['lstm_layer', 'embedding_rowmask', 'softmax_cross_entropy', 'cross_entropy_nd', 'seq_activation_reg', 'conv_add_upsampled']
