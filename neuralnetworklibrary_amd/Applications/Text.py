"""Text application of the drop-in API: the model / loss side of the reference's Applications/Text.py
(§3 models :441-752, §4 losses :754-808) plus the language-model batch loader (:231-332).

HIP-backed hot path (K5/K5b): the embedding gather with vocabulary-row dropout (ops.embedding_rowmask), every LSTM
layer (ops.lstm_layer: one fp32-MFMA GEMM for all input projections + the recurrence kernels, BPTT in C), the tied
decoder GEMM (ops.linear) and the fused softmax-cross-entropy over the vocabulary (ops.cross_entropy_nd).
Same classes, constructor arguments, state_dict keys (enc.word_embed.embed.weight, enc.lstms.{l}.lstm.{weight_ih_l0,
bias_ih_l0,bias_hh_l0,weight_hh_l0_raw}, dec.lin.weight tied) and quirks as the reference: hidden dropout is applied
after EVERY LSTM layer including the last (Text.py:546), hidden state is carried (detached) across batches and never
reset by the Learner (Text.py:547-550).  spaCy tokenisation / numericalisation (:19-229) is CPU text preparation and
out of scope.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..General.Core import *          # noqa: F401,F403
from ..General.Layers import *        # noqa: F401,F403
from ..General.Learner import *       # noqa: F401,F403
from ..General.LossesMetrics import * # noqa: F401,F403
from ..General.Optimizer import *     # noqa: F401,F403
from ..General.Core import TEN, correct_foldername, default_device, list_mult, separate_bn_layers
from ..General.Layers import FullyConnectedNet
from .. import ops
from ..dist import keyed_mask


# ---- data: language-model batches (Text.py:231-332) ----------------------------------------------------------------

class LanguageModelDataLoader(object):
    """All texts concatenated, split into bs parallel streams [bs, seqlen+1]; yields consecutive (x, y) windows of
    `bptt` tokens (train: 5 % of batches halved, minus U{0..9} jitter), y = x shifted by one (Text.py:231-290).
    `ds` needs `.texts` (sequence of token-id lists, or a pandas Series) and `.num_tokens`.
    device_resident=True (MI355X addition): the `[bs, seqlen+1]` token matrix is uploaded once per epoch
    and the (x, y) windows are device slices — no per-step numpy -> tensor -> H2D copy."""

    def __init__(self, ds, bs, bptt, random=True, device_resident=False):
        self.bs, self.bptt, self.random, self.device_resident = bs, bptt, random, device_resident
        self.texts, self.ntexts = ds.texts, len(ds.texts)
        self.seqlen = (ds.num_tokens // bs) - 1
        self.ntoks = bs * (self.seqlen + 1)
        self.concat_texts()
        self.set_batch_lengths()

    def _text(self, i):
        return self.texts.iloc[i] if hasattr(self.texts, 'iloc') else self.texts[i]

    def concat_texts(self):
        idxs = list(range(self.ntexts))
        if self.random:
            np.random.shuffle(idxs)
        flat = np.concatenate([np.asarray(self._text(i), dtype=np.int64) for i in idxs])[:self.ntoks]
        self.combined_text = flat.reshape(self.bs, self.seqlen + 1)
        self.combined_dev = TEN(self.combined_text) if self.device_resident else None

    def set_batch_lengths(self):
        self.batch_lengths = []
        i, used = 0, 0
        while used < self.seqlen:
            bptt = self.bptt
            if self.random and i > 0 and np.random.random() < 0.05:
                bptt = bptt // 2
            if self.random and i > 0:
                bptt = bptt - np.random.randint(0, 10)
            n = min(self.seqlen - used, bptt)
            used += n
            i += 1
            self.batch_lengths.append(n)

    def __len__(self):
        return len(self.batch_lengths)

    def __iter__(self):
        used = 0
        for bl in self.batch_lengths:
            if self.combined_dev is not None:
                yield (self.combined_dev[:, used:used + bl].contiguous(), self.combined_dev[:, used + 1:used + bl + 1].contiguous())
            else:
                yield (TEN(self.combined_text[:, used:used + bl]), TEN(self.combined_text[:, used + 1:used + bl + 1]))
            used += bl
        if self.random:
            self.concat_texts()


class LanguageModelDataObj(object):
    "train / val / (test) LanguageModelDataLoaders, target_type 'lang_model' (Text.py:292-304)"

    def __init__(self, train_ds, val_ds, test_ds, bs, bptt, device_resident=False):
        self.bs, self.bptt, self.stoi, self.target_type = bs, bptt, train_ds.stoi, 'lang_model'
        self.train_ds, self.val_ds, self.test_ds = train_ds, val_ds, test_ds
        self.train_dl = LanguageModelDataLoader(train_ds, bs, bptt, True, device_resident)
        self.val_dl = LanguageModelDataLoader(val_ds, bs, bptt, False, device_resident)
        if test_ds:
            self.test_dl = LanguageModelDataLoader(test_ds, bs, bptt, False, device_resident)


# ---- models --------------------------------------------------------------------------------------------------------

class LockedDropout(nn.Module):
    "One dropout mask [1, bs, C] shared by every timestep of x [T, bs, C] (Text.py:443-452)"

    def __init__(self, drop):
        super().__init__()
        self.drop = nn.Dropout(drop)

    def forward(self, x, mask=None):
        if mask is None and self.training and self.drop.p > 0:
            mask = keyed_mask((1, x.size(1), x.size(2)), self.drop.p, x.device, sample_dim=1)      # None unless use_keyed_dropout()
        if mask is None:
            mask = self.drop(torch.ones(1, x.size(1), x.size(2), device=x.device))
        return mask * x


class EmbeddingDropout(nn.Module):
    "Word embedding with whole-row (per vocabulary entry) dropout, then locked dropout (Text.py:454-475)"

    def __init__(self, vocab_size, emb_dim, drop1, drop2, pad_token):
        super().__init__()
        self.vocab_size, self.pad_token = vocab_size, pad_token
        self.drop1, self.drop2 = nn.Dropout(drop1), LockedDropout(drop2)
        self.embed = nn.Embedding(vocab_size, emb_dim, pad_token)
        nn.init.uniform_(self.embed.weight, -0.1, 0.1)
        with torch.no_grad():
            self.embed.weight[pad_token].zero_()

    def forward(self, x, row_mask=None, locked_mask=None):
        # x: [seqlen, bs] -> [seqlen, bs, emb_dim]
        if not self.training:
            return ops.embedding_rowmask(x, self.embed.weight, None, self.pad_token)
        if row_mask is None and self.drop1.p > 0:
            row_mask = keyed_mask((self.vocab_size, 1), self.drop1.p, x.device)     # parameter-shaped: the same on every rank
        if row_mask is None:
            row_mask = self.drop1(torch.ones(self.vocab_size, 1, device=x.device))
        out = ops.embedding_rowmask(x, self.embed.weight, row_mask, self.pad_token)
        return self.drop2(out, locked_mask)


class _LSTMParams(nn.Module):
    """Parameter holder with nn.LSTM's names and init (uniform(-1/sqrt(H), 1/sqrt(H))) in the order the reference ends
    up with after clear_non_raw(): weight_ih_l0, bias_ih_l0, bias_hh_l0, weight_hh_l0_raw (Text.py:486-493)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        mk = lambda *s: nn.Parameter(torch.empty(*s).uniform_(-k, k))
        self.weight_ih_l0 = mk(4 * hidden_size, input_size)
        self.bias_ih_l0 = mk(4 * hidden_size)
        self.bias_hh_l0 = mk(4 * hidden_size)
        self.weight_hh_l0_raw = mk(4 * hidden_size, hidden_size)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # reference checkpoints saved before clear_non_raw() also carry `weight_hh_l0` (the same tensor): ignore it
        state_dict.pop(prefix + 'weight_hh_l0', None)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class WeightDropLSTM1(nn.Module):
    """Single-layer LSTM whose hidden-to-hidden matrix gets elementwise dropout (one mask per forward call, shared by
    all timesteps) — Text.py:477-513."""

    nnl_stateful_forward = True        # the weight-drop seed of a call is drawn on the host (ops_text.lstm_layer): not replayable

    def __init__(self, input_size, hidden_size, drop):
        super().__init__()
        self.weight_drop = nn.Dropout(drop)
        self.lstm = _LSTMParams(input_size, hidden_size)

    def setup_raw(self):
        pass

    def clear_non_raw(self):
        pass          # there is no non-raw copy to clear: weight_hh_l0_raw is the only recurrent parameter

    def forward(self, x, h0c0, weight_mask=None):
        p = self.lstm
        if weight_mask is None and self.training and self.weight_drop.p > 0:
            weight_mask = keyed_mask(p.weight_hh_l0_raw.shape, self.weight_drop.p, p.weight_hh_l0_raw.device)   # None unless keyed
        # W = Dropout_p(W_raw) (Text.py:511) is applied INSIDE the layer: one pass that also pads W for the recurrence kernels
        drop_p = self.weight_drop.p if (self.training and weight_mask is None) else 0.0
        return ops.lstm_layer(x, h0c0[0], h0c0[1], p.weight_ih_l0, p.weight_hh_l0_raw, p.bias_ih_l0, p.bias_hh_l0,
                              weight_mask=weight_mask, weight_p=drop_p)


class LSTM_Encoder(nn.Module):
    """Embedding dropout -> num_layers x (WeightDropLSTM1 -> locked hidden dropout), carried hidden state
    (Text.py:515-551).  `fixed_masks` (dict with 'emb_rows', 'emb_locked', 'weights'[l], 'hidden'[l]) pins the dropout
    masks for parity tests."""

    nnl_stateful_forward = True        # carries (h, c) between minibatches on the Python side: Learner.use_graphs() keeps such steps eager

    def __init__(self, vocab_size, emb_dim, hidden_size, num_layers, pad_token, drops, bs):
        super().__init__()
        emb_drop1, emb_drop2, weight_drop, hidden_drop = drops
        self.word_embed = EmbeddingDropout(vocab_size, emb_dim, emb_drop1, emb_drop2, pad_token)
        self.hidden_drop = LockedDropout(hidden_drop)
        self.sizes = [emb_dim] + (num_layers - 1) * [hidden_size] + [emb_dim]
        self.lstms = nn.ModuleList([WeightDropLSTM1(self.sizes[i], self.sizes[i + 1], weight_drop) for i in range(num_layers)])
        self.fixed_masks = None
        self.reset(bs)

    def reset(self, bs):
        dev = self.word_embed.embed.weight.device
        self.h = [torch.zeros(1, bs, self.sizes[i], device=dev) for i in range(1, len(self.sizes))]
        self.c = [torch.zeros(1, bs, self.sizes[i], device=dev) for i in range(1, len(self.sizes))]

    def forward(self, x):
        # x: [bs, seqlen] -> [seqlen, bs, emb_dim]
        fm = self.fixed_masks or {}
        dev = self.word_embed.embed.weight.device
        if self.h[0].device != dev:
            self.h, self.c = [t.to(dev) for t in self.h], [t.to(dev) for t in self.c]
        x = self.word_embed(x.transpose(1, 0), fm.get('emb_rows'), fm.get('emb_locked'))
        h_n, c_n = [], []
        for i, lstm in enumerate(self.lstms):
            wm = fm['weights'][i] if 'weights' in fm else None
            x, (hn, cn) = lstm(x, (self.h[i], self.c[i]), wm)
            x = self.hidden_drop(x, fm['hidden'][i] if 'hidden' in fm else None)     # also after the last layer (:546)
            h_n.append(hn.detach())
            c_n.append(cn.detach())
        self.h, self.c = h_n, c_n
        return x


class LanguageModelDecoder(nn.Module):
    "Locked dropout -> tied linear -> [bs, vocab, seqlen] (Text.py:553-573)"

    def __init__(self, vocab_size, emb_dim, drop, tied_weight):
        super().__init__()
        self.lin = nn.Linear(emb_dim, vocab_size, bias=False)
        self.drop = LockedDropout(drop)
        self.lin.weight = tied_weight
        self.fixed_mask = None

    def forward(self, enc_out):
        pred = ops.linear(self.drop(enc_out, self.fixed_mask), self.lin.weight, None, wgrad_side=True)      # [seq, bs, V]; its 170-GFLOP dW runs beside the BPTT
        return pred.permute(1, 2, 0), enc_out


class TextClassificationDecoder(nn.Module):
    "Attention pooling over encoder outputs -> FullyConnectedNet (Text.py:575-609)"

    def __init__(self, emb_dim, num_classes, attn_size, fc_layer_sizes, fc_drops):
        super().__init__()
        self.fc = FullyConnectedNet([emb_dim] + fc_layer_sizes + [num_classes], fc_drops)
        self.attn1 = nn.Linear(emb_dim, attn_size)
        self.attn2 = nn.Linear(attn_size, 1)
        for m in (self.attn1, self.attn2):
            nn.init.kaiming_normal_(m.weight)
            nn.init.constant_(m.bias, 0)

    def forward(self, enc_in, enc_out):
        attn = ops.linear(enc_out, self.attn1.weight, self.attn1.bias, relu=True)       # seqlen x bs x attn_size
        attn = ops.linear(attn, self.attn2.weight, self.attn2.bias).squeeze()          # seqlen x bs
        attn = F.softmax(attn, dim=0)
        attn = attn * (enc_in.transpose(1, 0) != 1).float()                            # ignore the pad token
        attn = attn / attn.sum(dim=0).unsqueeze(0)
        combined = (attn.unsqueeze(2) * enc_out).sum(0)                                # bs x emb_dim
        return self.fc(combined), attn


class _Vocab:
    "minimal stand-in for a data object: stoi + bs"
    def __init__(self, stoi, bs):
        self.stoi, self.bs = stoi, bs


class LanguageModelNet(nn.Module):
    """AWD-LSTM language model: LSTM_Encoder (400 / 1150 / 3 layers) + tied LanguageModelDecoder (Text.py:611-702).
    `pretrained` weights (wt103) are LFS blobs absent from the reference snapshot -> only None is supported here."""

    def __init__(self, data, enc_drops=[0.05, 0.25, 0.2, 0.15], dec_drop=0.1, drop_scaling=0.7, pretrained=None,
                 emb_dim=400, hidden_size=1150, num_layers=3):
        super().__init__()
        enc_drops, dec_drop = list_mult(list(enc_drops), drop_scaling), dec_drop * drop_scaling
        vocab_size, pad_token = len(data.stoi), data.stoi['_pad_']
        self.bs, self.stoi, self.itos = data.bs, data.stoi, {i: s for s, i in data.stoi.items()}
        self.enc = LSTM_Encoder(vocab_size, emb_dim, hidden_size, num_layers, pad_token, enc_drops, self.bs)
        if pretrained:
            raise NotImplementedError('wt103 pretrained weights are not part of the reference snapshot (LFS pointers)')
        self.dec = LanguageModelDecoder(vocab_size, emb_dim, dec_drop, tied_weight=self.enc.word_embed.embed.weight)
        self.head = self.dec
        self.layer_groups = [self.enc.lstms, self.head]
        self.param_groups = separate_bn_layers(self.layer_groups)

    def reset(self):
        self.enc.reset(self.bs)

    def clear_non_raw(self):
        self.to(default_device())
        for lstm in self.enc.lstms:
            lstm.clear_non_raw()

    def forward(self, x):
        return self.dec(self.enc(x))


class TextClassificationNet(nn.Module):
    "LSTM_Encoder initialised from a language model + attention decoder (Text.py:704-751)"

    def __init__(self, PATH, language_model, num_classes, attn_size=100, enc_drops=[0.05, 0.25, 0.2, 0.15],
                 drop_scaling=0.7, fc_layer_sizes=[100], fc_drops=[0.25, 0.25]):
        super().__init__()
        import os
        enc_drops = list_mult(list(enc_drops), drop_scaling)
        lm_enc = language_model.enc
        emb_dim, hidden_size, num_layers = lm_enc.sizes[0], lm_enc.sizes[1], len(lm_enc.lstms)
        vocab_size, pad_token = len(language_model.stoi), language_model.stoi['_pad_']
        self.bs, self.stoi = language_model.bs, language_model.stoi
        PATH = correct_foldername(PATH)
        os.makedirs(PATH + 'models', exist_ok=True)
        torch.save(lm_enc.state_dict(), PATH + 'models/lang_model_enc.pt')
        self.enc = LSTM_Encoder(vocab_size, emb_dim, hidden_size, num_layers, pad_token, enc_drops, self.bs)
        self.enc.load_state_dict(torch.load(PATH + 'models/lang_model_enc.pt'), strict=False)
        self.dec = TextClassificationDecoder(emb_dim, num_classes, attn_size, fc_layer_sizes, fc_drops)
        self.head = self.dec
        self.layer_groups = [self.enc.lstms, self.enc.word_embed, self.head]
        self.param_groups = separate_bn_layers(self.layer_groups)

    def clear_non_raw(self):
        self.to(default_device())

    def forward(self, x, attn_vals=False):
        self.enc.reset(len(x))
        enc_out = self.enc(x)
        pred_classes, attn_values = self.dec(x, enc_out)
        return (pred_classes, enc_out, attn_values) if attn_vals else (pred_classes, enc_out)


# ---- losses / metrics (Text.py:756-808) --------------------------------------------------------------------------

class RegSeqCrossEntropyLoss(object):
    """CE(preds, target) + alpha*mean(enc_out^2) + beta*mean((enc_out[1:]-enc_out[:-1])^2) (Text.py:756-777).
    `.cross_entropy` keeps the unregularised CE as a detached 0-dim device tensor (the reference re-wraps
    loss.item(): same value, without the host sync)."""

    def __init__(self, alpha=2.0, beta=1.0):
        self.alpha, self.beta = alpha, beta
        self.cross_entropy = torch.zeros(())

    def __call__(self, outputs, target):
        preds, enc_out = outputs
        loss = ops.cross_entropy_nd(preds, target)
        self.cross_entropy = loss.detach()
        if enc_out.is_cuda and enc_out.dim() == 3 and (self.alpha > 0 or self.beta > 0):
            # alpha * mean(h^2) + beta * mean((h[1:] - h[:-1])^2) in one reduction kernel (ops_text.seq_activation_reg)
            return loss + ops.seq_activation_reg(enc_out, max(self.alpha, 0.0), max(self.beta, 0.0))
        if self.alpha > 0:
            loss = loss + self.alpha * enc_out.pow(2).mean()
        if self.beta > 0:
            loss = loss + self.beta * (enc_out[1:] - enc_out[:-1]).pow(2).mean()
        return loss


class SeqCrossEntropyLoss(object):
    "The unregularised CE of a RegSeqCrossEntropyLoss instance (Text.py:779-788)"
    def __init__(self, regularized_loss):
        self.regularized_loss = regularized_loss

    def __call__(self, outputs, target):
        return self.regularized_loss.cross_entropy


class LanguageModelAccuracy(object):
    "Next-token accuracy ignoring the 4 special tokens (Text.py:791-799)"
    def __call__(self, outputs, target):
        preds, _ = outputs
        preds[:, :4, :] = 0
        return (preds.max(dim=1)[1] == target).sum().float() / (target.shape[0] * target.shape[1])


class TextClassificationAccuracy(object):
    "Text.py:801-808"
    def __call__(self, outputs, target):
        preds, _ = outputs
        return (preds.max(dim=1)[1] == target).sum().float() / len(target)
