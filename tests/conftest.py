import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) when no GPU is visible, e.g. `pytest tests/` on CPU."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def T(a, device='cpu'):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device)


def assert_close(actual, expected, rtol=1e-5, atol=1e-6, msg=''):
    a = actual.detach().cpu().double().numpy() if torch.is_tensor(actual) else np.asarray(actual, dtype=np.float64)
    e = expected.detach().cpu().double().numpy() if torch.is_tensor(expected) else np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f'{msg} shape {a.shape} vs {e.shape}'
    err = np.abs(a - e)
    tol = atol + rtol * np.abs(e)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape) if err.ndim else ()
        raise AssertionError(f'{msg} max abs err {err.max():.3e} (at {i}: got {a[i]!r}, want {e[i]!r}); rtol={rtol} atol={atol}')


@pytest.fixture(autouse=True)
def _nnl_env_switches_fresh():
    """the library caches its NNL_* switches per call site: re-read them around every test (tests change them with monkeypatch)"""
    try:
        from neuralnetworklibrary_amd._lib import lib
    except Exception:
        lib = None
    if lib is not None:
        lib.nnl_reload_env()
    yield
    if lib is not None:
        lib.nnl_reload_env()
    try:                                                  # Learner.use_keyed_dropout() flips a process-wide switch
        from neuralnetworklibrary_amd.dist import drop_ctx
        drop_ctx.enabled = False
    except Exception:
        pass
