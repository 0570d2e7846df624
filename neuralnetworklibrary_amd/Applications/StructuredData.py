"""Structured (tabular) data application of the drop-in API: model side of Applications/StructuredData.py §2
(StructuredDataset :803, StructuredDataCollater :849, StructuredDataObj :871, embedding_dim :970,
StructuredDataNet :979-1096, StructuredDataEnsembleNet :1098).

HIP-backed hot path: the categorical front end (per-column max_norm renorm + gather + per-sample dropout + both
torch.cat calls) is ONE fused gather kernel (ops.tab_embed_concat, K3) and one scatter-add kernel in backward; the
FC head runs on the fp32-MFMA GEMM (ops.linear) and the fused BatchNorm kernels (ops.bn_act).
Out of scope (offline pandas feature engineering / EDA plots, never inside fit(): SURVEY.md §2.1 rows 7-8):
ProcessDataFrame and the §1 helpers — `StructuredDataObj` is built from ready arrays / StructuredDatasets.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import Dataset, DataLoader

from ..General.Core import *          # noqa: F401,F403
from ..General.Layers import *        # noqa: F401,F403
from ..General.Learner import *       # noqa: F401,F403
from ..General.LossesMetrics import * # noqa: F401,F403
from ..General.Optimizer import *     # noqa: F401,F403
from ..General.Core import TEN, separate_bn_layers
from ..General.Layers import EmbeddingDrop, Flatten1d, FullyConnectedNet
from .. import ops
from ..dist import keyed_mask


class StructuredDataset(Dataset):
    """x_cat int64 [N, n_cat'], x_cont float32 [N, n_cont'], y [N]; a missing block becomes a single zero column
    (Applications/StructuredData.py:803-847).  Accepts DataFrames or arrays."""

    def __init__(self, xcat_df, xcont_df, y, target_type):
        self.target_type = target_type
        L = len(xcat_df) if (xcat_df is not None) else len(xcont_df)
        self.y = y if (y is not None) else np.zeros(L).astype('float32')
        if xcat_df is not None:
            self.n_cat, self.x_cat = xcat_df.shape[1], np.array(xcat_df)
        else:
            self.n_cat, self.x_cat = 0, np.zeros((L, 1), 'int64')
        if xcont_df is not None:
            self.n_cont, self.x_cont = xcont_df.shape[1], np.array(xcont_df)
        else:
            self.n_cont, self.x_cont = 0, np.zeros((L, 1), 'float32')

    def __len__(self):
        return len(self.x_cat)

    def __getitem__(self, idx):
        return self.x_cat[idx], self.x_cont[idx], self.y[idx]

    def y_range(self):
        return [np.min(self.y), np.max(self.y)]


def StructuredDataCollater(batch):
    "list of (x_cat[i], x_cont[i], y[i]) -> [xcat, xcont], y as CPU tensors (StructuredData.py:849-869)"
    xcat = TEN(np.array([z[0] for z in batch]), GPU=False)
    xcont = TEN(np.array([z[1] for z in batch]), GPU=False)
    y = TEN(np.array([z[2] for z in batch]), GPU=False)
    return [xcat, xcont], y


class StructuredDataObj(object):
    """train / val / (test) StructuredDatasets + dataloaders (StructuredData.py:871-911).
    device_resident=True (MI355X addition): x_cat / x_cont / y live in HBM and `[xcat, xcont], y`
    minibatches are gathered on the device (device_data.DeviceBatches): no DataLoader workers, collate or per-step H2D."""

    def __init__(self, train_ds, val_ds, category_labels, scaling_values, bs, num_workers=6, test_ds=None,
                 device_resident=False, seed=0):
        self.train_ds, self.val_ds, self.test_ds = train_ds, val_ds, test_ds
        self.category_labels, self.scaling_values = category_labels, scaling_values
        self.bs, self.num_workers, self.target_type = bs, num_workers, train_ds.target_type
        if device_resident:
            from .. import dist as nnl_dist
            from ..device_data import DeviceBatches
            def mk(ds, shuffle, shard):
                y = np.asarray(ds.y)
                y = y.astype('int64') if ds.target_type == 'cat' else y.astype('float32')
                return DeviceBatches([np.asarray(ds.x_cat).astype('int64'), np.asarray(ds.x_cont).astype('float32')], y, bs,
                                     shuffle=shuffle, seed=seed, rank=nnl_dist.rank() if shard else 0,
                                     world=nnl_dist.world_size() if shard else 1)
            self.train_dl, self.val_dl = mk(train_ds, True, True), mk(val_ds, False, False)
            if self.test_ds:
                self.test_dl = mk(test_ds, False, False)
            return
        kw = dict(batch_size=bs, collate_fn=StructuredDataCollater, num_workers=num_workers, pin_memory=True)
        self.train_dl = DataLoader(train_ds, shuffle=True, **kw)
        self.val_dl = DataLoader(val_ds, shuffle=False, **kw)
        if self.test_ds:
            self.test_dl = DataLoader(test_ds, shuffle=False, **kw)


def embedding_dim(n):
    "A 'reasonable' embedding dimension for n classes (StructuredData.py:970-977)"
    if 2 <= n <= 8: return int(np.ceil(n / 2))
    if 9 <= n <= 12: return 5
    if 13 <= n <= 18: return 6
    if 19 <= n <= 27: return 7
    if 28 <= n <= 100: return int(np.ceil(n / 4))
    if n > 100: return 25


class StructuredDataNet(nn.Module):
    """Categorical embeddings (+ row dropout) ++ BatchNorm1d'ed, dropped-out continuous inputs -> FullyConnectedNet
    (StructuredData.py:979-1084).  Same constructor, sub-module names and layer groups as the reference.
    `inject_masks(row_masks [n_cat, bs], cont_mask [bs, n_cont])` pins the dropout masks (parity tests)."""

    nnl_default_graphs = True        # launch-bound at notebook batch sizes: Learner replays the whole step as a hipGraph by default

    def __init__(self, target_type, n_cat, n_cont, category_labels, fc_layer_sizes,
                 emb_sizes='default', output_range=None, dropout_levels=None):
        super().__init__()
        self.n_cat, self.n_cont = n_cat, n_cont
        if dropout_levels is None:
            dropout_levels = (0, 0, None)
        self.cont_bn = nn.BatchNorm1d(n_cont)
        self.cont_drop = nn.Dropout(dropout_levels[1])
        if emb_sizes == 'default':
            labels = category_labels if target_type == 'cont' else category_labels[0:-1]
            emb_sizes = [(len(D), embedding_dim(len(D))) for D in labels]
        self.embeddings = nn.ModuleList([EmbeddingDrop(c, d, dropout_levels[0], std=1 / d ** 0.5, max_norm=1.5)
                                         for c, d in emb_sizes])
        layer_sizes = [sum(d for c, d in emb_sizes) + n_cont] + fc_layer_sizes
        final_activ = 'sigmoidal' if (target_type == 'cont' and output_range) else None
        fc = FullyConnectedNet(layer_sizes, dropout_levels[2], final_activ, output_range, pre_bn=False)
        self.head = fc if target_type == 'cat' else nn.Sequential(fc, Flatten1d())
        self.layer_groups = [nn.ModuleList([self.embeddings, self.cont_bn, self.cont_drop]), self.head]
        self.param_groups = separate_bn_layers(self.layer_groups)
        self._plan, self._injected = None, None

    def inject_masks(self, row_masks=None, cont_mask=None):
        self._injected = (row_masks, cont_mask)

    def _masks(self, bs, device):
        if self._injected is not None:
            return self._injected
        row_masks = cont_mask = None
        p_emb = self.embeddings[0].drop.p if self.n_cat > 0 else 0
        p_cont = self.cont_drop.p
        want_rows = self.training and p_emb > 0
        want_cont = self.training and p_cont > 0 and self.n_cont > 0
        if want_rows:          # Layers.py:75-76: drop(ones(len(x))) per column
            row_masks = keyed_mask((self.n_cat, bs), p_emb, device, sample_dim=1)       # None unless Learner.use_keyed_dropout()
        if want_cont:
            cont_mask = keyed_mask((bs, self.n_cont), p_cont, device, sample_dim=0)
        need_rows, need_cont = want_rows and row_masks is None, want_cont and cont_mask is None
        if (need_rows or need_cont) and device.type == 'cuda':
            # both masks from one uniform draw + one launch (Bernoulli(keep) / keep each, as nn.Dropout on ones / on the inputs)
            a, b = ops.keep_masks((self.n_cat, bs) if need_rows else None, 1 - p_emb, (bs, self.n_cont) if need_cont else None, 1 - p_cont, device)
            row_masks = a if need_rows else row_masks
            cont_mask = b if need_cont else cont_mask
        else:
            if need_rows:
                row_masks = torch.empty(self.n_cat, bs, device=device).bernoulli_(1 - p_emb).div_(1 - p_emb)
            if need_cont:
                cont_mask = torch.empty(bs, self.n_cont, device=device).bernoulli_(1 - p_cont).div_(1 - p_cont)
        return row_masks, cont_mask

    def forward(self, xcat_batch, xcont_batch):
        bs, dev = len(xcat_batch), xcat_batch.device
        row_masks, cont_mask = self._masks(bs, dev)
        cont = ops.bn_act(self.cont_bn, xcont_batch, relu=False) if self.n_cont > 0 else None
        if self.n_cat > 0:
            weights = [e.emb.weight for e in self.embeddings]
            sync = getattr(self, 'nnl_dp', None)
            if sync is not None and getattr(self, '_nnl_xren_ready', False):
                sync = tuple(sync[:3]) + (self._nnl_xren,)           # all ranks' indices, gathered before the (captured) step
                self._nnl_xren_ready = False
            combined, self._plan = ops.tab_embed_concat(xcat_batch, weights, row_masks, cont, cont_mask,
                                                        self.embeddings[0].emb.max_norm, self._plan, sync=sync)
        else:
            combined = cont if cont_mask is None else cont * cont_mask
        return self.head(combined)

    def nnl_dp_prepare(self, x_batch):
        """Data-parallel replay (Learner.use_graphs under distribute()): the renorm sync's all-gather of every rank's looked-up indices
         run EAGERLY on the step's static input, into a static buffer the captured forward reads — the collective stays
        outside the hipGraph.  Returns True when the next forward will use the buffer."""
        sync = getattr(self, 'nnl_dp', None)
        if sync is None or self.n_cat == 0 or self.embeddings[0].emb.max_norm is None:
            return False
        xcat = x_batch[0] if isinstance(x_batch, (list, tuple)) else x_batch
        xren = ops._global_lookup_indices(xcat.contiguous().long(), sync[:3])
        buf = getattr(self, '_nnl_xren', None)
        if buf is None or buf.shape != xren.shape or buf.device != xren.device:
            self._nnl_xren = xren.clone()
        else:
            buf.copy_(xren)
        self._nnl_xren_ready = True
        return True

    @classmethod
    def from_dataobj(cls, data, fc_layer_sizes, emb_sizes='default', output_range=None, dropout_levels=None):
        return cls(data.target_type, data.train_ds.n_cat, data.train_ds.n_cont, data.category_labels, fc_layer_sizes,
                   emb_sizes, output_range, dropout_levels)


class StructuredDataEnsembleNet(nn.Module):
    "Weighted average of several tabular models, optional softmax correction (StructuredData.py:1098-1133)"

    def __init__(self, models, weights=None, correction=None):
        super().__init__()
        n = len(models)
        self.weights = weights if weights else [1 / n] * n
        self.correction = correction
        self.models = nn.ModuleList(models)
        self.layer_groups = models
        self.param_groups = separate_bn_layers(self.layer_groups)

    def forward(self, xcat, xcont):
        if self.correction is None:
            return sum(w * m(xcat, xcont) for w, m in zip(self.weights, self.models))
        if self.correction == 'cat':
            return sum(w * F.log_softmax(m(xcat, xcont), dim=1).exp() for w, m in zip(self.weights, self.models))
