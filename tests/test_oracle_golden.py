"""Pins the oracle (oracle/reference_math.py, CPU torch restatement) to golden vectors produced by running the
reference itself (oracle/gen_golden.py).  CPU-only; this is the "parity pin" of the oracle."""
import numpy as np
import torch

from conftest import T, assert_close, load_golden
from oracle import reference_math as RM


def _collab_params(g, prefix='init.'):
    names = ['user_emb.weight', 'item_emb.weight', 'user_bias.weight', 'item_bias.weight']
    return [T(g[prefix + n]).clone().requires_grad_(True) for n in names]


def test_g1_collab_forward_loss_grads():
    g = load_golden('g1_collab')
    U, M, bu, bi = _collab_params(g)
    x, y = T(g['x0']), T(g['y0'])
    pred = RM.embdotbias(x, U, M, bu, bi, [float(g['lo']), float(g['hi'])])
    assert_close(pred, g['pred0'], rtol=1e-6, atol=1e-6, msg='pred')
    assert_close(RM.embdotbias(x, U, M, bu, bi, None), g['pred0_norange'], rtol=1e-6, atol=1e-6, msg='pred_norange')
    loss = RM.mse_loss(pred, y)
    assert_close(loss, g['loss0'], rtol=1e-6, msg='loss')
    loss.backward()
    for p, n in zip([U, M, bu, bi], ['user_emb', 'item_emb', 'user_bias', 'item_bias']):
        assert_close(p.grad, g['grad0.%s.weight' % n], rtol=1e-5, atol=1e-7, msg=n)


def test_g1_collab_three_adam_steps():
    """Restated train1minibatch: decoupled wd -> Adam, lr 1e-2, wd 1e-4 (one layer group)."""
    g = load_golden('g1_collab')
    params = _collab_params(g)
    state = RM.OptimState(params)
    losses = []
    for i in range(3):
        x, y = T(g['x%d' % i]), T(g['y%d' % i])
        for p in params:
            p.grad = None
        loss = RM.mse_loss(RM.embdotbias(x, *params, [float(g['lo']), float(g['hi'])]), y)
        loss.backward()
        losses.append(loss.item())
        RM.optimizer_step(params, [p.grad for p in params], state, [1e-2] * 4, [1e-4] * 4, 'adam')
    assert_close(np.array(losses), g['step_losses'], rtol=1e-6, msg='losses')
    for p, n in zip(params, ['user_emb', 'item_emb', 'user_bias', 'item_bias']):
        assert_close(p, g['after3.%s.weight' % n], rtol=1e-5, atol=1e-7, msg=n)
