"""Collaborative filtering application of the drop-in API (mirror of Applications/CollabFiltering.py).

`CollabFilterNet` keeps the reference's four nn.Embedding sub-modules (=> state_dict keys
user_emb.weight, item_emb.weight, user_bias.weight, item_bias.weight; CollabFiltering.py:189-190) but its
forward is ONE fused HIP kernel (ops.embdotbias, K4) instead of 4 gathers + ~8 elementwise launches
(CollabFiltering.py:196-204), and its backward one scatter-add kernel.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset, DataLoader

from ..General.Core import *          # noqa: F401,F403
from ..General.Layers import *        # noqa: F401,F403
from ..General.Learner import *       # noqa: F401,F403
from ..General.LossesMetrics import * # noqa: F401,F403
from ..General.Optimizer import *     # noqa: F401,F403
from ..General.Core import SplitTrainVal, separate_bn_layers
from ..General.Layers import get_embedding
from .. import ops


class CollabFilterDataset(Dataset):
    """(user, item) -> rating rows of a DataFrame, relabelled to dense integer ids
    (Applications/CollabFiltering.py:29-72).  x: int64 [N,2]; y: float32 [N]; y_range = [min, max]."""

    def __init__(self, df, user_col, item_col, rating_col, labels):
        users = df[user_col].map(labels[0]).astype('int64')
        items = df[item_col].map(labels[1]).astype('int64')
        self.x = np.stack([users.to_numpy(), items.to_numpy()], axis=1)
        if rating_col is None:
            self.y = np.zeros(len(df), 'float32')
        else:
            self.y = df[rating_col].to_numpy().astype('float32')
        self.y_range = [float(np.min(self.y)), float(np.max(self.y))]

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx]


class CollabFilterDataObj(object):
    """train / val / (test) datasets + dataloaders, target_type 'cont' (CollabFiltering.py:75-116).
    device_resident=True (MI355X addition): the id / rating arrays live in HBM and minibatches are
    gathered on the device (device_data.DeviceBatches) instead of going through DataLoader workers and per-step H2D copies;
    under torch.distributed every rank takes its slice of each global minibatch."""

    def __init__(self, train_df, val_df, user_col, item_col, rating_col,
                 labels, bs, num_workers=6, test_df=None, device_resident=False, seed=0):
        self.bs, self.labels, self.target_type = bs, labels, 'cont'
        self.train_ds = CollabFilterDataset(train_df, user_col, item_col, rating_col, labels)
        self.val_ds = CollabFilterDataset(val_df, user_col, item_col, rating_col, labels)
        if test_df is not None:
            self.test_ds = CollabFilterDataset(test_df, user_col, item_col, None, labels)
        if device_resident:
            from .. import dist as nnl_dist
            from ..device_data import DeviceBatches
            mk = lambda ds, shuffle, shard: DeviceBatches(
                ds.x, ds.y, bs, shuffle=shuffle, seed=seed, rank=nnl_dist.rank() if shard else 0,
                world=nnl_dist.world_size() if shard else 1)
            self.train_dl, self.val_dl = mk(self.train_ds, True, True), mk(self.val_ds, False, False)
            if test_df is not None:
                self.test_dl = mk(self.test_ds, False, False)
            return
        kw = dict(batch_size=bs, num_workers=num_workers, pin_memory=True)
        self.train_dl = DataLoader(self.train_ds, shuffle=True, **kw)
        self.val_dl = DataLoader(self.val_ds, shuffle=False, **kw)
        if test_df is not None:
            self.test_dl = DataLoader(self.test_ds, shuffle=False, **kw)

    @classmethod
    def from_csv(cls, train_csv, user_col, item_col, rating_col, bs, val_csv=None, test_csv=None,
                 val_idxs=None, val_frac=0.2, num_workers=6):
        """Build from csv file(s); labels come from the training csv (CollabFiltering.py:118-165)."""
        import pandas as pd
        train_df = pd.read_csv(train_csv).reindex(columns=[user_col, item_col, rating_col])
        users, items = train_df[user_col].unique(), train_df[item_col].unique()
        labels = [{u: i for i, u in enumerate(users)}, {m: i for i, m in enumerate(items)}]
        if val_csv is not None:
            val_df = pd.read_csv(val_csv).reindex(columns=[user_col, item_col, rating_col])
        else:
            train_df, val_df = SplitTrainVal(train_df, val_idxs, val_frac)
        test_df = pd.read_csv(test_csv).reindex(columns=[user_col, item_col]) if test_csv is not None else None
        return cls(train_df, val_df, user_col, item_col, rating_col, labels, bs, num_workers, test_df)


class CollabFilterNet(nn.Module):
    """y = lo + (hi-lo)*sigmoid(<user_emb[u], item_emb[i]> + user_bias[u] + item_bias[i])
    (Applications/CollabFiltering.py:168-213).  One layer group."""

    nnl_default_graphs = True        # launch-bound at notebook batch sizes: Learner replays the whole step as a hipGraph by default

    def __init__(self, n_user, n_item, emb_dim, output_range):
        super().__init__()
        self.output_range = output_range
        self.user_emb, self.item_emb = get_embedding(n_user, emb_dim), get_embedding(n_item, emb_dim)
        self.user_bias, self.item_bias = get_embedding(n_user, 1), get_embedding(n_item, 1)
        self.layer_groups = [nn.ModuleList([self.user_emb, self.item_emb, self.user_bias, self.item_bias])]
        self.param_groups = separate_bn_layers(self.layer_groups)

    def forward(self, x_batch):
        return ops.embdotbias(x_batch, self.user_emb.weight, self.item_emb.weight,
                              self.user_bias.weight, self.item_bias.weight, self.output_range)

    @classmethod
    def from_dataobj(cls, data, emb_dim, output_range='default'):
        n_user, n_item = len(data.labels[0]), len(data.labels[1])
        if output_range == 'default':
            lo, hi = data.train_ds.y_range
            output_range = [lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo)]
        return cls(n_user, n_item, emb_dim, output_range)


class CollabFilterEnsembleNet(nn.Module):
    "Weighted average of the outputs of several collab-filter models (CollabFiltering.py:216-242)."

    def __init__(self, models, weights=None):
        super().__init__()
        n = len(models)
        self.weights = weights if weights else [1 / n] * n
        self.models = nn.ModuleList(models)
        self.layer_groups = models
        self.param_groups = separate_bn_layers(self.layer_groups)

    def forward(self, x):
        return sum(w * m(x) for w, m in zip(self.weights, self.models))
