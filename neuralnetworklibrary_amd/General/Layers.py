"""Building-block layers of the drop-in API (mirror of the reference's General/Layers.py).

Same class names, constructor arguments, sub-module names (=> identical state_dict keys) and forward
results.  The arithmetic of the hot-path layers runs in hand-written HIP kernels through `ops` — there is no
CPU branch: these modules raise NnlError on non-CUDA tensors (CPU parity lives in oracle/, test-only):
  * `EmbeddingDrop.forward`  (Layers.py:63-76)  -> ops.embedding_renorm_drop (single column) /
    the fused multi-column kernel used by StructuredDataNet;
  * `Linear.forward`         (Layers.py:37-41)  -> ops.linear (MFMA GEMM + bias + ReLU epilogue), BN1d after
    the ReLU as in the reference;
  * `AdaptiveConcatPool2d`   (Layers.py:78-87)  -> ops.concat_pool2d.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .Core import initialize_modules
from ..dist import KeyedDropout, keyed_mask

__all__ = ['Flatten', 'Flatten1d', 'Linear', 'Conv2d', 'get_embedding', 'EmbeddingDrop', 'AdaptiveConcatPool2d',
           'FullyConnectedNet']


def _ops():
    from .. import ops
    return ops


class Flatten(nn.Module):
    "(bs, n1, ..., nk) -> (bs, n1*...*nk)   (General/Layers.py:20-23)"
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class Flatten1d(nn.Module):
    "(bs, 1) -> (bs,)   (General/Layers.py:25-28)"
    def forward(self, x):
        return x.reshape(-1)


def get_embedding(num_cats, emb_dim, std=0.01, max_norm=None):
    """nn.Embedding whose weight is std * (N(0,1) fmod 2), i.e. a truncated normal (General/Layers.py:56-61)."""
    emb = nn.Embedding(num_cats, emb_dim, max_norm=max_norm)
    with torch.no_grad():
        emb.weight.normal_().fmod_(2).mul_(std)
    return emb


class Linear(nn.Module):
    "dropout -> nn.Linear -> ReLU -> BatchNorm1d (BN AFTER the ReLU)   (General/Layers.py:30-41)"
    def __init__(self, nin, nout, bn=True, drop=0):
        super().__init__()
        self.lin = nn.Linear(nin, nout)
        self.bn = nn.BatchNorm1d(nout) if bn else None
        self.drop = KeyedDropout(drop) if drop else None

    def forward(self, x):
        if self.drop:
            x = self.drop(x)
        if self.bn and x.dim() == 2:
            return _ops().linear_relu_bn(self.lin, self.bn, x)           # BatchNorm statistics from the GEMM epilogue
        x = _ops().linear(x, self.lin.weight, self.lin.bias, relu=True)
        if self.bn:
            x = _ops().bn_act(self.bn, x, relu=False)
        return x


class Conv2d(nn.Module):
    "dropout -> nn.Conv2d -> ReLU -> BatchNorm2d   (General/Layers.py:43-54)"
    def __init__(self, nin, nout, ks=3, stride=1, pad=1, bn=True, drop=0):
        super().__init__()
        self.conv = nn.Conv2d(nin, nout, ks, stride, pad)
        self.bn = nn.BatchNorm2d(nout) if bn else None
        self.drop = KeyedDropout(drop) if drop else None

    def forward(self, x):
        if self.drop:
            x = self.drop(x)
        x = _ops().conv2d(x, self.conv.weight, self.conv.bias, self.conv.stride[0], self.conv.padding[0], relu=True)
        if self.bn:
            x = _ops().bn_act(self.bn, x, relu=False)
        return x


class EmbeddingDrop(nn.Module):
    """Embedding (truncated-normal init, in-place max_norm renorm of the looked-up rows) times a per-SAMPLE
    Bernoulli keep mask scaled by 1/(1-p)   (General/Layers.py:63-76)."""
    def __init__(self, num_cats, emb_dim, drop, std, max_norm):
        super().__init__()
        self.drop = nn.Dropout(drop)
        self.emb = nn.Embedding(num_cats, emb_dim, max_norm=max_norm)
        with torch.no_grad():
            self.emb.weight.normal_().fmod_(2).mul_(std)

    def row_mask(self, n, device):
        "the reference's `drop(ones(len(x)))`: 0 or 1/(1-p) per sample row"
        if self.training and self.drop.p > 0:
            m = keyed_mask((n,), self.drop.p, device, sample_dim=0)
            if m is not None:
                return m
        return self.drop(torch.ones(n, device=device))

    def forward(self, x):
        mask = self.row_mask(len(x), x.device)
        return _ops().embedding_renorm_drop(x, self.emb.weight, mask, self.emb.max_norm)


class AdaptiveConcatPool2d(nn.Module):
    "cat([AdaptiveMaxPool2d(sz)(x), AdaptiveAvgPool2d(sz)(x)], 1)   (General/Layers.py:78-87)"
    def __init__(self, sz=None):
        super().__init__()
        sz = sz or (1, 1)
        self.sz = sz
        self.ap = nn.AdaptiveAvgPool2d(sz)
        self.mp = nn.AdaptiveMaxPool2d(sz)

    def forward(self, x):
        if tuple(self.sz) == (1, 1):
            return _ops().concat_pool2d(x)
        return torch.cat([self.mp(x), self.ap(x)], 1)   # non-global pooling: not on any BASELINE config


class FullyConnectedNet(nn.Module):
    """Multi-layer fully connected net: [pre_bn] -> (Linear block)* -> dropout -> final nn.Linear ->
    optional 'softmax' / 'sigmoidal'(output_range) activation   (General/Layers.py:89-154).
    kaiming-normal weights, zero biases (Layers.py:137)."""
    def __init__(self, layer_sizes, drops=None, final_activ=None, output_range=None, bn=True, pre_bn=True):
        super().__init__()
        N = len(layer_sizes) - 1
        if drops is None:
            drops = [0] * N
        self.final_activ = final_activ
        self.output_range = output_range
        self.pre_bn = nn.BatchNorm1d(layer_sizes[0]) if pre_bn else None
        self.lins = nn.ModuleList([Linear(layer_sizes[i], layer_sizes[i + 1], bn, drops[i]) for i in range(N - 1)])
        self.final_drop = KeyedDropout(drops[N - 1])
        self.final_lin = nn.Linear(layer_sizes[N - 1], layer_sizes[N])
        initialize_modules([self.lins, self.final_lin], nn.init.kaiming_normal_, False)

    def forward(self, x):
        if self.pre_bn:
            x = _ops().bn_act(self.pre_bn, x, relu=False)
        for lin in self.lins:
            x = lin(x)
        x = self.final_drop(x)
        x = _ops().linear(x, self.final_lin.weight, self.final_lin.bias, relu=False)
        if self.final_activ == 'softmax':
            x = F.log_softmax(x, dim=1).exp()
        elif self.final_activ == 'sigmoidal':
            lo, hi = float(self.output_range[0]), float(self.output_range[1])
            x = _ops().scaled_sigmoid(x, lo, hi) if (x.is_cuda and hi != lo) else lo + (hi - lo) * x.sigmoid()
        return x
