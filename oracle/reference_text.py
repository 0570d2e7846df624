"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not imported by the product package.

CPU (torch fp32, eager) restatement of the reference's language-model path (Applications/Text.py): the LSTM layer is
written out as explicit gate equations (the reference delegates to torch's nn.LSTM, Text.py:483,513 — restating
torch's published cell equations, gate order i,f,g,o), every dropout mask is an explicit argument.  Same sub-module
/ parameter names as the reference after clear_non_raw().  Pinned by tests/golden/g7_text.npz (reference run).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def lstm_layer(x, h0, c0, w_ih, w_hh, b_ih, b_hh):
    "x [T,B,I], h0/c0 [1,B,H] -> y [T,B,H], (hT, cT) [1,B,H]; torch nn.LSTM semantics"
    h, c = h0[0], c0[0]
    ys = []
    for t in range(x.shape[0]):
        g = x[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys.append(h)
    return torch.stack(ys), (h.unsqueeze(0), c.unsqueeze(0))


class _LSTMParams(nn.Module):
    def __init__(self, I, H):
        super().__init__()
        k = 1.0 / math.sqrt(H)
        mk = lambda *s: nn.Parameter(torch.empty(*s).uniform_(-k, k))
        self.weight_ih_l0, self.bias_ih_l0, self.bias_hh_l0 = mk(4 * H, I), mk(4 * H), mk(4 * H)
        self.weight_hh_l0_raw = mk(4 * H, H)


class WeightDropLSTM1(nn.Module):
    "Text.py:477-513; weight_mask = the dropout mask (already scaled by 1/(1-p)) on weight_hh_l0_raw, or None"
    def __init__(self, I, H):
        super().__init__()
        self.lstm = _LSTMParams(I, H)

    def forward(self, x, h0c0, weight_mask=None):
        p = self.lstm
        w_hh = p.weight_hh_l0_raw if weight_mask is None else p.weight_hh_l0_raw * weight_mask
        return lstm_layer(x, h0c0[0], h0c0[1], p.weight_ih_l0, w_hh, p.bias_ih_l0, p.bias_hh_l0)


class _Embed(nn.Module):
    def __init__(self, V, E, pad):
        super().__init__()
        self.embed = nn.Embedding(V, E, pad)


class LSTM_Encoder(nn.Module):
    """Text.py:515-551 (+ EmbeddingDropout :454-475, LockedDropout :443-452).  masks: dict with 'emb_rows' [V,1],
    'emb_locked' [1,B,E], 'weights'[l], 'hidden'[l] (None entries / missing keys = no dropout)."""
    def __init__(self, V, E, Hh, L, pad, bs):
        super().__init__()
        self.word_embed = _Embed(V, E, pad)
        self.pad = pad
        self.sizes = [E] + (L - 1) * [Hh] + [E]
        self.lstms = nn.ModuleList([WeightDropLSTM1(self.sizes[i], self.sizes[i + 1]) for i in range(L)])
        self.reset(bs)

    def reset(self, bs):
        self.h = [torch.zeros(1, bs, s) for s in self.sizes[1:]]
        self.c = [torch.zeros(1, bs, s) for s in self.sizes[1:]]

    def forward(self, x, masks=None):
        masks = masks or {}
        x = x.transpose(1, 0)
        W = self.word_embed.embed.weight
        if masks.get('emb_rows') is not None:
            W = W * masks['emb_rows']                                           # :473-474
        x = F.embedding(x, W, self.pad)
        if masks.get('emb_locked') is not None:
            x = x * masks['emb_locked']
        hn, cn = [], []
        for i, l in enumerate(self.lstms):
            wm = masks['weights'][i] if masks.get('weights') else None
            x, (h, c) = l(x, (self.h[i], self.c[i]), wm)
            if masks.get('hidden'):
                x = x * masks['hidden'][i]                                      # after every layer, incl. the last (:546)
            hn.append(h.detach()); cn.append(c.detach())
        self.h, self.c = hn, cn
        return x


class _Dec(nn.Module):
    def __init__(self, V, E, tied):
        super().__init__()
        self.lin = nn.Linear(E, V, bias=False)
        self.lin.weight = tied


class LanguageModelNet(nn.Module):
    "Text.py:611-653 with dropout masks as arguments (dec_mask [1,B,E])"
    def __init__(self, V, pad, bs, E=400, Hh=1150, L=3):
        super().__init__()
        self.enc = LSTM_Encoder(V, E, Hh, L, pad, bs)
        self.dec = _Dec(V, E, self.enc.word_embed.embed.weight)

    def forward(self, x, masks=None, dec_mask=None):
        enc_out = self.enc(x, masks)
        d = enc_out if dec_mask is None else enc_out * dec_mask
        return self.dec.lin(d).permute(1, 2, 0), enc_out                         # :572


def reg_seq_cross_entropy(outputs, target, alpha=2.0, beta=1.0):
    "RegSeqCrossEntropyLoss.__call__ (Text.py:765-777): returns (loss, plain cross-entropy)"
    preds, enc_out = outputs
    ce = F.cross_entropy(preds, target)
    loss = ce
    if alpha > 0:
        loss = loss + alpha * enc_out.pow(2).mean()
    if beta > 0:
        loss = loss + beta * (enc_out[1:] - enc_out[:-1]).pow(2).mean()
    return loss, ce.detach()
