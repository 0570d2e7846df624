"""Core helpers of the drop-in API (mirror of the reference's General/Core.py).

Same public names, arguments and results as the reference (cited per function), but device-agnostic:
the reference hard-codes `.cuda()` (General/Core.py:70,140-144); here every placement goes through
`default_device()`, which is the local MI355X (`cuda:<LOCAL_RANK>`) when one is visible and the CPU
otherwise (host-logic tests).  Heavy, hot-path-irrelevant imports of the reference prelude (seaborn,
spacy, cv2, skimage, GPUtil: General/Core.py:9-22) are not made.
"""
import copy
import os

import numpy as np
import torch
import torch.nn as nn

try:  # pandas is only needed by SplitTrainVal / the DataObj constructors
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None

__all__ = ['TEN', 'ARR', 'LIST', 'list_del', 'list_mult', 'outer_mult', 'linear_space', 'joint_sort',
           'correct_foldername', 'bn_types', 'linconv_types', 'to_cuda', 'trainable_params', 'num_children',
           'flatten_module', 'initialize_module', 'initialize_modules', 'separate_bn_layers',
           'make_model_basic', 'SaveFeatures', 'SplitTrainVal', 'combine_models', 'combine_preds',
           'default_device', 'set_default_device']

_DEVICE = None


def default_device():
    """Device every `TEN(..., GPU=True)` / `to_cuda` / `Learner` places tensors on."""
    global _DEVICE
    if _DEVICE is None:
        if torch.cuda.is_available():
            _DEVICE = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))
        else:
            _DEVICE = torch.device('cpu')
    return _DEVICE


def set_default_device(device):
    global _DEVICE
    _DEVICE = torch.device(device)
    return _DEVICE


# ---- variable type conversion (reference General/Core.py:46-83) ------------------------------------

_FLOAT_NP = (np.float32, np.float64)
_INT_NP = (np.int32, np.int64)


def TEN(x, GPU=True):
    """list / ndarray / python or numpy scalar -> FloatTensor or LongTensor (General/Core.py:46-71)."""
    if isinstance(x, list):
        x = np.array(x)
    if isinstance(x, np.ndarray):
        if x.dtype in _FLOAT_NP:
            x = torch.as_tensor(x, dtype=torch.float32)
        elif x.dtype in _INT_NP:
            x = torch.as_tensor(x, dtype=torch.int64)
    elif isinstance(x, (float,) + _FLOAT_NP):
        x = torch.tensor(float(x), dtype=torch.float32)
    elif isinstance(x, (int,) + _INT_NP) and not isinstance(x, bool):
        x = torch.tensor(int(x), dtype=torch.int64)
    if GPU:
        x = x.to(default_device())
    return x


def ARR(x):
    """Tensor (any device) -> numpy array on the host (General/Core.py:73-76)."""
    return x.detach().cpu().numpy() if x.requires_grad or x.is_cuda else x.numpy()


def LIST(x, N, Tuple=True, Array=True):
    """Broadcast x to a length-N list 'in a natural way' (General/Core.py:78-83)."""
    if isinstance(x, list) and len(x) == N:
        return x
    if Tuple and isinstance(x, tuple) and len(x) == N:
        return list(x)
    if Array and isinstance(x, np.ndarray) and len(x) == N:
        return list(x)
    return [x] * N


# ---- regular utilities (General/Core.py:88-133) ------------------------------------------------------

def list_del(L, idxs):
    drop = set(idxs)
    return [v for i, v in enumerate(L) if i not in drop]


def list_mult(L, c):
    return [v * c for v in L] if type(L) == list else L * c


def outer_mult(A, B):
    return np.array([A * b for b in B])


def linear_space(A, B, N):
    if isinstance(A, (float, int)):
        return np.linspace(A, B, N)
    return np.array([np.linspace(a, b, N) for a, b in zip(A, B)]).transpose()


def joint_sort(lists, reverse=False):
    key = lists[0]
    order = sorted(range(len(key)), key=key.__getitem__, reverse=reverse)
    return [[L[i] for i in order] for L in lists]


def correct_foldername(folder_name):
    return folder_name if folder_name.endswith('/') else folder_name + '/'


# ---- torch utilities (General/Core.py:137-215) -------------------------------------------------------

bn_types = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)
linconv_types = (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d)


def to_cuda(x):
    """Move a tensor or (nested) list of tensors to the default device (General/Core.py:140-144).
    Non-blocking: batches handed over in pinned memory overlap with compute."""
    if isinstance(x, (list, tuple)):
        return [to_cuda(v) for v in x]
    dev = default_device()
    return x if x.device == dev else x.to(dev, non_blocking=True)


def trainable_params(m):
    return [p for p in m.parameters() if p.requires_grad]


def num_children(m):
    return sum(1 for _ in m.children())


def flatten_module(m):
    """Leaves (child-less modules) of m in definition order (General/Core.py:154-157)."""
    kids = list(m.children())
    if not kids:
        return [m]
    out = []
    for k in kids:
        out += flatten_module(k)
    return out


def initialize_module(m, init_func, bn_init=False):
    """init_func on linear/conv weights, zero their biases, optional BN (1,0) (General/Core.py:159-175)."""
    for l in m.modules():
        if isinstance(l, linconv_types):
            init_func(l.weight)
            if l.bias is not None:
                nn.init.constant_(l.bias, 0)
        elif bn_init and isinstance(l, bn_types):
            nn.init.constant_(l.weight, 1)
            nn.init.constant_(l.bias, 0)


def initialize_modules(L, init_func, bn_init=False):
    for m in L:
        initialize_module(m, init_func, bn_init)


def separate_bn_layers(layer_groups):
    """[G_1..G_N] -> [nonBN(G_1)..nonBN(G_N), BN(G_1)..BN(G_N)] as ModuleLists — the param-group contract
    `Optimizer` relies on (General/Core.py:181-197)."""
    reg, bn = [], []
    for G in layer_groups:
        leaves = flatten_module(G)
        reg.append(nn.ModuleList([l for l in leaves if not isinstance(l, bn_types)]))
        bn.append(nn.ModuleList([l for l in leaves if isinstance(l, bn_types)]))
    return reg + bn


def make_model_basic(model):
    model.layer_groups = [model]
    model.param_groups = separate_bn_layers(model.layer_groups)
    return model


class SaveFeatures():
    """Forward hook that keeps the last output of module m (General/Core.py:209-215)."""
    features = None

    def __init__(self, m):
        self.hook = m.register_forward_hook(self.hook_fn)

    def hook_fn(self, module, input, output):
        self.features = output

    def close(self):
        self.hook.remove()


# ---- data splitting (General/Core.py:220-247) ---------------------------------------------------------

def SplitTrainVal(datapoints, val_idxs=None, val_frac=0.2):
    N = len(datapoints)
    if val_idxs is None:
        val_idxs = list(np.random.choice(np.arange(N), int(N * val_frac), replace=False))
    train_idxs = list(set(np.arange(N)) - set(val_idxs))
    if pd is not None and type(datapoints) == pd.DataFrame:
        return datapoints.iloc[train_idxs].copy(), datapoints.iloc[val_idxs].copy()
    if type(datapoints) == list:
        return [datapoints[i] for i in train_idxs], [datapoints[i] for i in val_idxs]


# ---- combining models / predictions (General/Core.py:252-309) ----------------------------------------

def combine_models(model_list, weights=None):
    """Weighted average of parameters and buffers of same-architecture models (SWA)."""
    n = len(model_list)
    if weights is None:
        weights = [1 / n] * n
    avg = copy.deepcopy(model_list[0])
    states = [m.state_dict() for m in model_list]
    merged = {}
    for name in states[0]:
        merged[name] = sum(w * s[name] for w, s in zip(weights, states))
    avg.load_state_dict(merged)
    return avg


def combine_preds(preds, target_type, weights=None):
    n = len(preds)
    if weights is None:
        weights = [1 / n] * n
    combined = sum(w * p for w, p in zip(weights, preds))
    if target_type == 'cont':
        return combined
    if target_type in ['cat', 'single_label']:
        return combined, combined.argmax(axis=1)
    if target_type == 'multi_label':
        return combined, combined.round().astype(int)
