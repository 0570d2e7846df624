"""The MFMA : VALU mix of the hot k loops, read from the compiler's gfx950 assembly (no GPU needed: hipcc cross-compiles).

On gfx950 VALU instructions are not hidden behind a dependent MFMA chain of the same SIMD (tools/coissue_probe.hip,
profiles/r5_coissue_probe.log), so every address / mask / transform instruction of a k loop is paid in matrix time.  Round 5 cut the two
weight-gradient kernels from 110 -> 83 and 56 -> 33 static VALU instructions per 16 MFMAs (headline -0.3 ms, RetinaNet -0.85 ms) and removed
32 accumulator copies per pair of k tiles from the two-tiles-in-flight tap kernel; this test keeps those loops from growing back unnoticed
(a compiler update or an innocent-looking edit of the staging code is enough).  Counts are STATIC instructions between the loop's labels
(conditional blocks included), as tools/loop_valu_survey.py prints them."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'neuralnetworklibrary_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.fixture(scope='module')
def conv_loops(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    sys.path.insert(0, ROOT)
    from tools.loop_valu_survey import survey
    out = str(tmp_path_factory.mktemp('asm') / 'conv2d.s')
    subprocess.run([HIPCC, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-Wno-unused-function', '-S',
                    '--cuda-device-only', 'conv2d.hip', '-o', out], cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return survey(out)


def _loops(rows, needle):
    hit = [r for r in rows if needle in r['kernel']]
    assert hit, 'no MFMA loop found in a kernel matching %r' % needle
    return hit


# (kernel name fragment of the mangled instantiation, MFMAs per iteration, VALU bound, bound on register moves or None)
CASES = [
    ('igemm_wgrad2d_kernelILi64ELi32ELb1ELi4E', 16, 90, None),           # Winograd-domain weight gradient, 64 x 64 tile, four wave groups (was 110+)
    ('igemm_wgrad2d_kernelILi128ELi16ELb1ELi1E', 32, 106, None),         # ... 128 x 128 tile (was 125)
    ('igemm_wgrad_kernelILi64ELi64ELi32ELi2ELi2ELb1ELi1ELb0ELb1E', 16, 40, None),     # direct weight gradient, PAIR staging (one-chunk staging: 56)
    ('igemm_wgrad_kernelILi128ELi128ELi16ELi2ELi2ELb1ELi1ELb0ELb1E', 32, 56, None),   # (one-chunk staging: 70)
    ('igemm_taps_kernelILi64ELi64ELi32ELi2ELi2ELb1ELi0ELb0ELi1ELb0E', 16, 20, 0),     # forward / dgrad tap kernel: address state is scalar
    ('igemm_taps_kernelILi64ELi64ELi16ELi2ELi2ELb1ELi0ELb0ELi2ELb0E', 16, 12, 0),     # two tiles in flight: NO accumulator copies in the loop (was 16 x v_mov_b64)
]


@pytest.mark.parametrize('needle,mfma,valu_max,mov_max', CASES)
def test_hot_loops_keep_their_valu_budget(conv_loops, needle, mfma, valu_max, mov_max):
    for r in _loops(conv_loops, needle):
        assert r['mfma'] == mfma, r
        assert r['valu'] <= valu_max, 'VALU instructions in the k loop of %s: %d > %d' % (r['kernel'], r['valu'], valu_max)
        if mov_max is not None:
            assert r['mov'] <= mov_max, r
