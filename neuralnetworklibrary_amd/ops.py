"""torch.autograd.Function wrappers around the C ABI of libnnl_hip.so (include/nnl.h).

Each Function is the device-side replacement of one group of eager torch ops in the reference (call sites
cited per class).  Tensors are handed over as raw device pointers + sizes on torch's CURRENT stream; there
is no CPU fallback — a non-CUDA tensor raises NnlError.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream

__all__ = ['embdotbias', 'index_error_flag', 'raise_if_index_error']

_ERR_FLAGS = {}


def index_error_flag(device):
    """Per-device int32 flag that gather kernels set when they meet an out-of-range index (the sample is
    skipped, nothing faults).  Checked without a per-step sync by `raise_if_index_error()`."""
    key = (device.type, device.index)
    if key not in _ERR_FLAGS:
        _ERR_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _ERR_FLAGS[key]


def raise_if_index_error():
    """One D2H read per call: raises IndexError (torch's nn.Embedding failure) if any gather kernel since the
    last call saw an out-of-range index.  The Learner calls this once per epoch."""
    for flag in _ERR_FLAGS.values():
        if int(flag.item()) != 0:
            flag.zero_()
            raise IndexError("index out of range in self")


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
        dU, dM = torch.empty_like(U), torch.empty_like(M)
        dbu = torch.empty(U.shape[0], 1, dtype=torch.float32, device=U.device)
        dbi = torch.empty(M.shape[0], 1, dtype=torch.float32, device=U.device)
        check(lib.nnl_embdotbias_bwd(ptr(x), ptr(U), ptr(M), ptr(z), ptr(dy), ptr(dU), ptr(dM), ptr(dbu), ptr(dbi),
                                     x.shape[0], U.shape[0], M.shape[0], U.shape[1], int(has_range), lo, hi,
                                     stream()))
        return None, dU, dM, dbu, dbi, None, None


def embdotbias(x, U, M, bu, bi, output_range=None):
    """y = lo + (hi-lo)*sigmoid(<U[x[:,0]], M[x[:,1]]> + bu[x[:,0]] + bi[x[:,1]]); x int64 [n,2]."""
    lo, hi = (None, None) if output_range is None else (float(output_range[0]), float(output_range[1]))
    return _EmbDotBias.apply(x, U, M, bu, bi, lo, hi)
