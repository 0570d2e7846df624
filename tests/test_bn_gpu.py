"""GPU parity of the fused BatchNorm(+residual)(+ReLU) kernels against torch CPU BatchNorm."""
import pytest
import torch
import torch.nn as nn

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'

CASES = [  # shape, residual, relu, training, mean offset
    ((4, 64, 14, 14), True, True, True, 0.0),
    ((2, 128, 7, 9), False, True, True, 3.0),
    ((3, 512, 2, 2), True, False, True, 0.0),
    ((64, 14), False, False, True, 50.0),          # BatchNorm1d, C % 4 != 0, |mean| >> std (cancellation check)
    ((1024, 1000), False, False, True, 0.0),       # tabular hidden layer
    ((8, 1024), False, False, True, 0.0),
    ((4, 64, 14, 14), True, True, False, 0.0),     # eval mode (frozen BN)
    ((5, 20, 3, 3), False, True, False, 1.0),
]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_bn_act(case):
    from neuralnetworklibrary_amd import ops
    shape, has_res, relu, training, offset = case
    g = torch.Generator().manual_seed(len(shape) * 100 + shape[1])
    C = shape[1]
    x = torch.randn(shape, generator=g) * 1.7 + offset
    res = torch.randn(shape, generator=g) if has_res else None
    dy = torch.randn(shape, generator=g)
    mk = nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d
    bn_c = mk(C)
    with torch.no_grad():
        bn_c.weight.copy_(torch.randn(C, generator=g) * 0.3 + 1)
        bn_c.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn_c.running_mean.copy_(torch.randn(C, generator=g) * 0.1 + offset)
        bn_c.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn_g = mk(C)
    bn_g.load_state_dict(bn_c.state_dict())
    bn_g = bn_g.to(DEV)
    bn_c.train(training); bn_g.train(training)

    xc = x.clone().requires_grad_(True)
    rc = res.clone().requires_grad_(True) if has_res else None
    yc = bn_c(xc)
    if has_res:
        yc = yc + rc
    if relu:
        yc = torch.relu(yc)
    yc.backward(dy)

    xg = x.to(DEV).requires_grad_(True)
    rg = res.to(DEV).requires_grad_(True) if has_res else None
    yg = ops.bn_act(bn_g, xg, residual=rg, relu=relu)
    yg.backward(dy.to(DEV))

    assert_close(yg, yc, 1e-4, 1e-5, 'y')
    assert_close(xg.grad, xc.grad, 1e-3, 1e-5 * xc.grad.abs().max().item() + 1e-7, 'dx')
    if has_res:
        assert_close(rg.grad, rc.grad, 1e-5, 1e-6, 'dres')
    assert_close(bn_g.weight.grad, bn_c.weight.grad, 1e-3, 1e-4 * bn_c.weight.grad.abs().max().item(), 'dgamma')
    assert_close(bn_g.bias.grad, bn_c.bias.grad, 1e-3, 1e-4 * bn_c.bias.grad.abs().max().item(), 'dbeta')
    assert_close(bn_g.running_mean, bn_c.running_mean, 1e-5, 1e-5, 'running_mean')
    assert_close(bn_g.running_var, bn_c.running_var, 1e-4, 1e-6, 'running_var')
    assert int(bn_g.num_batches_tracked) == int(bn_c.num_batches_tracked)
