"""Pooling kernels (csrc/pool.hip) against torch CPU: MaxPool2d(3,2,1) / other windows and AdaptiveConcatPool2d
(General/Layers.py:78-87), forward and backward, INCLUDING ties (post-ReLU zeros are common: the gradient must go to the first
maximum, as torch's max-pool backward does) and NaN propagation.  Bit-exact: pooling only selects / adds a few values."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _x(shape, seed, ties):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    if ties:                                    # quantise + ReLU: many equal values and zero plateaus
        x = (x * 2).round().clamp_(min=0) / 2
    return x


@pytest.mark.parametrize('shape,k,s,p', [((3, 8, 13, 11), 3, 2, 1), ((2, 64, 112, 112), 3, 2, 1), ((2, 4, 7, 7), 2, 2, 0),
                                         ((1, 12, 9, 10), 3, 1, 1), ((2, 4, 5, 5), 5, 3, 2)])
@pytest.mark.parametrize('ties', [False, True])
def test_maxpool2d(shape, k, s, p, ties):
    from neuralnetworklibrary_amd import ops
    x = _x(shape, 1, ties)
    xc = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xc, k, s, p)
    dy = _x(tuple(ref.shape), 2, False)
    ref.backward(dy)
    xg = x.to(DEV).requires_grad_(True)
    out = ops.maxpool2d(xg, k, s, p)
    out.backward(dy.to(DEV))
    assert torch.equal(out.cpu(), ref), 'maxpool forward'
    assert_close(xg.grad, xc.grad, 1e-6, 1e-6, 'maxpool backward')   # sums of <= 4 values: order may differ from torch's atomics


def test_maxpool2d_nan_and_module():
    from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import HipMaxPool2d
    x = _x((1, 4, 6, 6), 3, False)
    x[0, 1, 2, 3] = float('nan')
    ref = F.max_pool2d(x, 3, 2, 1)
    out = HipMaxPool2d(3, 2, 1)(x.to(DEV)).cpu()
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert torch.equal(out[~torch.isnan(ref)], ref[~torch.isnan(ref)])


@pytest.mark.parametrize('shape', [(4, 512, 7, 7), (3, 40, 5, 3), (2, 8, 1, 1), (2, 130, 16, 16)])
@pytest.mark.parametrize('ties', [False, True])
def test_concat_pool(shape, ties):
    from neuralnetworklibrary_amd import ops
    x = _x(shape, 4, ties)
    if ties:
        x[:, 0] = 0.0                            # a dead channel: all-equal plateau
    xc = x.clone().requires_grad_(True)
    ref = torch.cat([nn.AdaptiveMaxPool2d(1)(xc), nn.AdaptiveAvgPool2d(1)(xc)], 1)
    dy = _x(tuple(ref.shape), 5, False)
    ref.backward(dy)
    xg = x.to(DEV).requires_grad_(True)
    out = ops.concat_pool2d(xg)
    out.backward(dy.to(DEV))
    C = shape[1]
    assert torch.equal(out[:, :C].cpu(), ref[:, :C].detach()), 'max half'
    assert_close(out[:, C:], ref[:, C:].detach(), 1e-6, 1e-6, 'avg half')
    assert_close(xg.grad, xc.grad, 1e-6, 1e-7, 'backward')


def test_concat_pool_nan():
    from neuralnetworklibrary_amd import ops
    x = _x((1, 4, 3, 3), 6, False)
    x[0, 2, 1, 1] = float('nan'); x[0, 2, 2, 0] = float('nan')
    ref = nn.AdaptiveMaxPool2d(1, return_indices=True)(x)
    out = ops.concat_pool2d(x.to(DEV)).cpu()
    assert torch.isnan(out[0, 2, 0, 0]) and torch.isnan(ref[0][0, 2, 0, 0])
    assert torch.equal(out[0, [0, 1, 3], 0, 0], ref[0][0, [0, 1, 3], 0, 0])
