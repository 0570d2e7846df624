"""The C-ABI library loads and exports every symbol include/nnl.h declares (no compute: CPU-safe)."""
import os
import re
import subprocess

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'nnl.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nnl_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    from neuralnetworklibrary_amd import _lib
    names = _declared()
    assert len(names) >= 6
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (nnl_[a-z0-9_]+)', out))
    missing = [n for n in names if n not in exported]
    assert not missing, f'declared in nnl.h but not exported: {missing}'
    unbound = [n for n in names if n not in _lib.SIGNATURES]
    assert not unbound, f'declared in nnl.h but not bound in _lib.SIGNATURES: {unbound}'
    extra = [n for n in exported if n not in names]
    assert not extra, f'exported but not declared in nnl.h: {extra}'


def test_loaded_library_is_built_from_this_tree():
    """The loaded libnnl_hip.so carries the stamp of the sources it was built from (csrc/Makefile: sha256 over csrc/*.hip, *.h, the
    Makefile and include/nnl.h); it must equal the hash of the sources in THIS tree — a stale prebuilt library fails here, on the
    build box and on the GPU box alike (the sources travel with the snapshot)."""
    from neuralnetworklibrary_amd import _lib
    built, tree = _lib.source_stamp(), _lib.source_stamp_of_tree()
    assert len(built) == 16 and built == tree, 'libnnl_hip.so built from other sources (stamp %s, tree %s): run __graft_entry__.build()' % (built, tree)


def test_version_and_error_string():
    from neuralnetworklibrary_amd import _lib
    assert _lib.lib.nnl_version() >= 100
    assert isinstance(_lib.lib.nnl_last_error(), bytes)


def test_invalid_argument_is_reported_not_raised_in_c():
    """Argument validation happens before any HIP call, so it is testable without a GPU."""
    from neuralnetworklibrary_amd import _lib
    st = _lib.lib.nnl_embdotbias_fwd(None, None, None, None, None, None, None, 4, 0, 0, 0, 0, 0., 0., None, None)
    assert st == -1
    assert b'embdotbias' in _lib.lib.nnl_last_error()


def test_product_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd._lib import NnlError
    x = torch.zeros(4, 2, dtype=torch.long)
    w = torch.zeros(3, 2)
    b = torch.zeros(3, 1)
    with pytest.raises(NnlError):
        ops.embdotbias(x, w, w, b, b, [0., 1.])
