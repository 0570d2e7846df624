"""Language-model ops on the C ABI (K5 / K5b): LSTM layer, embedding with vocabulary-row dropout, fused softmax-CE.
Imported into `ops` (use `ops.lstm_layer`, `ops.embedding_rowmask`, `ops.softmax_cross_entropy`)."""
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _ceil4(n):
    return (n + 3) // 4 * 4


class _LSTMRecurrence(torch.autograd.Function):
    """The time recurrence of nn.LSTM(num_layers=1) (reference Applications/Text.py:483,513) given the input
    projections gx [T,B,4H]: forward = nnl_lstm_fwd, backward = nnl_lstm_bwd (BPTT) + one GEMM for dW_hh."""

    @staticmethod
    def forward(ctx, gx, w_raw, h0, c0, wmask=None, p=0.0, seed=0):
        """w_raw [4H,H]: the RAW recurrent matrix; the weight drop W = w_raw * m (Text.py:495-513) — m = `wmask` (explicit, already
        scaled) or Bernoulli(1-p)/(1-p) from (seed, element index) — and the zero padding to the kernels' k granularity are ONE
        pass (nnl_weight_drop); the backward re-derives m for dW_raw = dW * m."""
        require_cuda(gx, w_raw, h0, c0, wmask)
        gx, w_raw = _f32c(gx), _f32c(w_raw)
        T, B, G = gx.shape
        H = G // 4
        h0, c0 = _f32c(h0).view(B, H), _f32c(c0).view(B, H)
        Hp = int(lib.nnl_lstm_padded_hidden(H))
        dev = gx.device
        wm = None if wmask is None else _f32c(wmask)
        ctx.drop = (wm is not None or p > 0.0, float(p), int(seed))
        if Hp == H and not ctx.drop[0]:
            w_pad = w_raw
        else:
            w_pad = torch.empty(G, Hp, dtype=torch.float32, device=dev)
            check(lib.nnl_weight_drop(ptr(w_raw), H, ptr(wm), ptr(w_pad), Hp, G, H, int(seed), float(p), stream()))
        w_hh = w_pad                                        # [4H, Hp]: columns >= H are zero
        y = torch.empty(T, B, H, dtype=torch.float32, device=dev)
        cy = torch.empty(T, B, H, dtype=torch.float32, device=dev)
        from .ops import lstm_timeout_flag
        gates = torch.empty(T, B, G, dtype=torch.float32, device=dev)
        wsb = int(lib.nnl_lstm_workspace_bytes(T, B, H))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
        check(lib.nnl_lstm_fwd(ptr(gx), ptr(w_pad), ptr(h0), ptr(c0), ptr(y), ptr(cy), ptr(gates), T, B, H, ptr(ws), wsb,
                               ptr(lstm_timeout_flag(dev)), stream()))
        ctx.save_for_backward(w_hh, h0, c0, y, cy, gates, wm)
        ctx.side_param = w_raw if w_raw.requires_grad else None      # dW_hh may be computed on the side stream (ops._Side)
        return y, y[-1].clone(), cy[-1].clone()

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        w_hh, h0, c0, y, cy, gates, wm = ctx.saved_tensors
        T, B, H = y.shape
        G = 4 * H
        dev = y.device
        Hw = w_hh.shape[1]                                  # H, or the padded row length of the dropped matrix
        dy = None if dy is None else _f32c(dy)
        dhT = None if dhT is None else _f32c(dhT)
        dcT = None if dcT is None else _f32c(dcT)
        Gp = int(lib.nnl_lstm_padded_gates(H))
        w_t = torch.empty(Hw, G, dtype=torch.float32, device=dev)        # rows >= H (the zero pad columns of w_hh) are never read
        check(lib.nnl_conv2d_weight_transpose(ptr(w_hh), ptr(w_t), G, 1, 1, Hw, stream()))
        if Gp != G:
            w_t = F.pad(w_t, (0, Gp - G))
        dgates = torch.empty(T, B, Gp, dtype=torch.float32, device=dev)
        if Gp != G:
            dgates[:, :, G:].zero_()                         # only the pad columns must be zero (round 4 zero-filled all 82 MB per layer)
        dh0 = torch.empty(B, H, dtype=torch.float32, device=dev)
        dc0 = torch.empty(B, H, dtype=torch.float32, device=dev)
        wsb = int(lib.nnl_lstm_workspace_bytes(T, B, H))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
        from .ops import lstm_timeout_flag
        check(lib.nnl_lstm_bwd(ptr(dy), ptr(dhT), ptr(dcT), ptr(gates), ptr(cy), ptr(c0), ptr(w_t), ptr(dgates), ptr(dh0),
                               ptr(dc0), T, B, H, ptr(ws), wsb, ptr(lstm_timeout_flag(dev)), stream()))
        dw = None
        if ctx.needs_input_grad[1]:
            # dW_hh[4H,H] = sum_t dgates_t^T h_{t-1}: the wgrad kernel on a 1x1 "conv" over T*B "pixels"
            Hp = _ceil4(H)
            # h_{t-1} for t = 0 .. T-1, rows padded to Hp: ONE big strided copy (round 4: cat + pad = a copy, a fill and another copy)
            hprev = torch.empty(T, B, Hp, dtype=torch.float32, device=dev)
            hprev[0, :, :H] = h0
            if T > 1:
                hprev[1:, :, :H] = y[:-1]
            if Hp != H:
                hprev[:, :, H:].zero_()
            hprev = hprev.view(T * B, Hp)
            g = _lib.ConvGeom(T * B, 1, 1, Hp, Gp, 1, 1, 1, 0, 1, 1)
            dwp = torch.empty(Gp, Hp, dtype=torch.float32, device=dev)
            wb = int(lib.nnl_conv2d_wgrad_workspace_bytes(g))
            wws = torch.empty(max(wb // 4, 1), dtype=torch.float32, device=dev)
            use, p, seed = ctx.drop
            from . import ops as _ops
            side = _ops.side_ok(getattr(ctx, 'side_param', None))
            dense = use or (side and (Gp != G or Hp != H))   # (side stream: the un-padding copy happens THERE, not in autograd on the main stream)
            dw = torch.empty(G, H, dtype=torch.float32, device=dev) if dense else dwp[:G, :H]

            def run_w():
                check(lib.nnl_conv2d_wgrad(ptr(hprev), ptr(dgates.view(T * B, Gp)), ptr(dwp), g, ptr(wws), wb, stream()))
                if use:                                      # dW_raw = dW * m, un-padded in the same pass
                    check(lib.nnl_weight_drop(ptr(dwp), Hp, ptr(wm), ptr(dw), H, G, H, seed, p, stream()))
                elif dense:
                    dw.copy_(dwp[:G, :H])
            if side:
                # the 47-GFLOP GEMM runs beside the latency-bound BPTT of the layer below (which needs only dgates, returned now)
                _ops.side_run(run_w, (hprev, dgates, dwp, wws, dw, wm))
            else:
                run_w()
        if Gp != G:
            from .ops import register_padded_grad
            register_padded_grad(dgates, Gp)                 # rows already padded with zeros: the input GEMM's backward uses them as is
        return dgates[:, :, :G], dw, dh0.view_as(h0), dc0.view_as(c0), None, None, None


def lstm_layer(x, h0, c0, w_ih, w_hh, b_ih, b_hh, weight_mask=None, weight_p=0.0):
    """One-layer LSTM over x [T,B,I] with initial state (h0, c0) [1,B,H] (or [B,H]).  w_hh is the RAW recurrent matrix: the
    weight drop of WeightDropLSTM1 (Text.py:495-513) happens inside (`weight_mask`: an explicit scaled mask; else a fresh
    Bernoulli(1 - weight_p) mask per call, its seed drawn from torch's CPU generator so `torch.manual_seed` governs it).
    Returns y [T,B,H], (hT [1,B,H], cT [1,B,H]) like nn.LSTM."""
    from . import ops
    T, B, _ = x.shape
    H = w_hh.shape[1]
    gx = ops.linear(x.reshape(T * B, -1), w_ih, b_ih + b_hh, wgrad_side=True).view(T, B, 4 * H)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (weight_mask is None and weight_p > 0) else 0
    y, hT, cT = _LSTMRecurrence.apply(gx, w_hh, h0.reshape(B, H), c0.reshape(B, H), weight_mask, float(weight_p), seed)
    return y, (hT.view(1, B, H), cT.view(1, B, H))


class _EmbeddingRowMask(torch.autograd.Function):
    """F.embedding(x, W * mask[V,1], padding_idx) (reference Applications/Text.py:473-474) without materialising W*mask."""

    @staticmethod
    def forward(ctx, x, W, rowmask, padding_idx):
        from .ops import index_error_flag
        require_cuda(x, W, rowmask)
        xi = x.contiguous().long()
        W = _f32c(W)
        rm = None if rowmask is None else _f32c(rowmask).view(-1)
        V, D = W.shape
        out = torch.empty(xi.numel(), D, dtype=torch.float32, device=W.device)
        check(lib.nnl_embedding_rowmask_fwd(ptr(xi), ptr(W), ptr(rm), ptr(out), xi.numel(), V, D,
                                            ptr(index_error_flag(W.device)), stream()))
        ctx.save_for_backward(xi, rm)
        ctx.meta = (V, D, -1 if padding_idx is None else int(padding_idx))
        return out.view(*x.shape, D)

    @staticmethod
    def backward(ctx, dout):
        from . import ops as _ops
        _ops.side_join()          # the tied decoder weight's gradient (side stream) is about to be accumulated into by autograd
        xi, rm = ctx.saved_tensors
        V, D, pad = ctx.meta
        dout = _f32c(dout)
        dW = torch.empty(V, D, dtype=torch.float32, device=dout.device)
        wsb = int(lib.nnl_embedding_rowmask_bwd_workspace_bytes(xi.numel()))         # sample-order (deterministic) scatter-add
        ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dout.device)
        check(lib.nnl_embedding_rowmask_bwd(ptr(xi), ptr(rm), ptr(dout), ptr(dW), xi.numel(), V, D, pad, ptr(ws), wsb, stream()))
        return None, dW, None, None


def embedding_rowmask(x, W, rowmask=None, padding_idx=None):
    return _EmbeddingRowMask.apply(x, W, rowmask, padding_idx)


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        from .ops import index_error_flag
        require_cuda(logits, target)
        logits = _f32c(logits)
        target = target.contiguous().long()
        rows, V = logits.shape
        dev = logits.device
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        loss_rows = torch.empty(rows, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        check(lib.nnl_softmax_ce_fwd(ptr(logits), ptr(target), ptr(lse), ptr(loss_rows), ptr(loss), rows, V,
                                     ptr(index_error_flag(dev)), stream()))
        ctx.save_for_backward(logits, target, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target, lse = ctx.saved_tensors
        from .ops import register_padded_grad
        g = _f32c(dloss).view(1)
        rows, V = logits.shape
        Vp = (V + 15) // 16 * 16
        # rows padded to the GEMM granularity (zeros in the pad columns): the producing linear layer's backward uses the buffer as
        # is instead of re-padding it (V = 47 343: an 848 MB fill + copy per step)
        buf = torch.empty(rows, Vp, dtype=torch.float32, device=logits.device)
        check(lib.nnl_softmax_ce_bwd(ptr(logits), ptr(target), ptr(lse), ptr(g), ptr(buf), rows, V, Vp, stream()))
        if Vp == V:
            return buf, None
        register_padded_grad(buf, Vp)
        return buf[:, :V], None


def softmax_cross_entropy(logits, target):
    """mean_r( logsumexp(logits[r]) - logits[r, target[r]] ) for logits [rows, V], target [rows]."""
    return _SoftmaxCE.apply(logits, target)


def cross_entropy_nd(preds, target):
    """F.cross_entropy(preds, target) for class-dim-1 inputs: [N,C] with target [N], or [N,C,d1..] with [N,d1..]."""
    if preds.dim() == 2:
        return softmax_cross_entropy(preds, target)
    C = preds.shape[1]
    if preds.dim() == 3 and preds.permute(2, 0, 1).is_contiguous():
        # the language-model decoder returns lin(x).permute(1,2,0): a [bs,V,seq] VIEW of a contiguous [seq,bs,V]
        # buffer (Text.py:572) — use that buffer as is (the mean does not depend on the row order)
        return softmax_cross_entropy(preds.permute(2, 0, 1).reshape(-1, C), target.transpose(0, 1).reshape(-1))
    perm = [0] + list(range(2, preds.dim())) + [1]
    return softmax_cross_entropy(preds.permute(*perm).reshape(-1, C), target.reshape(-1))


class _SeqReg(torch.autograd.Function):
    """alpha * mean(h^2) + beta * mean((h[1:] - h[:-1])^2) for h = enc_out [T, bs, E] (reference Text.py:765-777: the AR / TAR
    terms of RegSeqCrossEntropyLoss) — one reduction pass forward, one elementwise pass backward."""

    @staticmethod
    def forward(ctx, h, alpha, beta):
        require_cuda(h)
        h = _f32c(h)
        T = h.shape[0]
        R = h.numel() // max(T, 1)
        out = torch.empty(3, dtype=torch.float32, device=h.device)
        wsb = int(lib.nnl_seq_reg_workspace_bytes(T, R))
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=h.device)
        check(lib.nnl_seq_reg_fwd(ptr(h), ptr(out), T, R, float(alpha), float(beta), ptr(ws), wsb, stream()))
        ctx.save_for_backward(h)
        ctx.ab = (float(alpha), float(beta))
        return out[0]

    @staticmethod
    def backward(ctx, dout):
        h, = ctx.saved_tensors
        T = h.shape[0]
        R = h.numel() // max(T, 1)
        g = _f32c(dout).reshape(1)
        dh = torch.empty_like(h)
        check(lib.nnl_seq_reg_bwd(ptr(h), ptr(g), ptr(dh), T, R, ctx.ab[0], ctx.ab[1], stream()))
        return dh, None, None


def seq_activation_reg(enc_out, alpha, beta):
    "AR + TAR regulariser of RegSeqCrossEntropyLoss as a 0-dim tensor (see _SeqReg)"
    return _SeqReg.apply(enc_out, float(alpha), float(beta))
