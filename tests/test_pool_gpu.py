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


@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('shape', [(4, 64, 32, 32), (2, 16, 15, 13)], ids=str)
def test_bn_relu_maxpool_fused_matches_the_three_stages(training, shape):
    """ops.conv_bn_relu_maxpool's fused BatchNorm -> ReLU -> MaxPool (one pass, no normalised activation in memory) against the
    same three stages run one after the other on the HIP kernels (bitwise equal forward: same arithmetic, same tie rule) and
    against torch in fp64 (running statistics, gradients)."""
    from neuralnetworklibrary_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    x[:, :, ::3, ::2] = x[:, :, ::3, ::2].round()                  # ties and exact zeros after the ReLU are common in practice
    bn_a, bn_b, bn_r = nn.BatchNorm2d(C), nn.BatchNorm2d(C), nn.BatchNorm2d(C).double()
    with torch.no_grad():
        for bn in (bn_a, bn_b, bn_r):
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.4, 0.4, C))
            bn.weight[:3] = torch.tensor([1e-3, 0.0, -0.7])         # tiny / zero / negative gamma: the backward's slow and sign paths
            bn.running_mean.copy_(torch.linspace(-0.2, 0.5, C)); bn.running_var.copy_(torch.linspace(0.8, 2.5, C))
    bn_a, bn_b = bn_a.to(DEV).train(training), bn_b.to(DEV).train(training)
    bn_r.train(training)
    pool = nn.MaxPool2d(3, 2, 1)
    ident = type('Ident', (), {'__call__': lambda self, t, **kw: t})()

    xa = x.to(DEV).requires_grad_(True)
    ya = ops.conv_bn_relu_maxpool(ident, bn_a, pool, xa)            # fused (the "conv" is the identity here)
    xb = x.to(DEV).requires_grad_(True)
    yb = ops.maxpool2d(ops.bn_act(bn_b, xb, relu=True), 3, 2, 1)    # three stages
    assert torch.equal(ya, yb)
    dy = torch.randn(ya.shape, generator=g)
    ya.backward(dy.to(DEV)); yb.backward(dy.to(DEV))
    assert_close(xa.grad, xb.grad, 1e-4, 1e-5 * xb.grad.abs().max().item(), 'dx fused vs staged')
    assert_close(bn_a.weight.grad, bn_b.weight.grad, 1e-4, 1e-4, 'dgamma fused vs staged')
    assert_close(bn_a.bias.grad, bn_b.bias.grad, 1e-4, 1e-4, 'dbeta fused vs staged')
    assert_close(bn_a.running_mean, bn_b.running_mean, 1e-6, 1e-6, 'running_mean')
    assert_close(bn_a.running_var, bn_b.running_var, 1e-6, 1e-6, 'running_var')
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked)

    xr = x.double().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(torch.relu(bn_r(xr)), 3, 2, 1)
    yr.backward(dy.double())
    assert_close(ya, yr.float(), 1e-4, 1e-5, 'y vs torch fp64')
    assert_close(bn_a.running_var, bn_r.running_var.float(), 1e-5, 1e-6, 'running_var vs torch')
    assert_close(bn_a.weight.grad, bn_r.weight.grad.float(), 2e-3, 2e-3, 'dgamma vs torch')
    bad = ((xa.grad.cpu() - xr.grad.float()).abs() > 1e-3 + 1e-3 * xr.grad.float().abs()).float().mean().item()
    assert bad < 1e-3, 'dx vs torch fp64: %.2e of the elements differ (ties between equal maxima may legitimately move)' % bad
