"""Host-only checks of the partition planner of the 2-D persistent recurrence kernel (csrc/lstm_bptt2.hip;
reference call site: WeightDropLSTM1.forward -> nn.LSTM, Applications/Text.py:495-513): whatever shape they accept, the partition
must cover the problem, fit the launch limits the kernels assume (<= 256 co-resident workgroups, k slices in 16-wide groups,
<= 6 column tiles) — no GPU needed: the debug entries only run the planner."""
import ctypes

import pytest

from neuralnetworklibrary_amd._lib import lib

SHAPES = [(64, 1150), (64, 400), (64, 1152), (20, 100), (64, 32), (33, 256), (7, 37), (1, 4), (64, 2048), (64, 3), (48, 999), (64, 1500)]


@pytest.mark.parametrize('B,H', SHAPES)
def test_bptt2_partition_covers_the_problem(B, H):
    out = (ctypes.c_int32 * 5)()
    ok = lib.nnl_debug_lstm_bptt2_plan(B, H, out)
    Gp = int(lib.nnl_lstm_padded_gates(H))
    if not ok:
        pytest.skip('shape not taken by the 2-D BPTT kernel (the per-timestep path runs)')
    KG, NG, Ks, Ns, NT = list(out)
    assert 1 <= KG * NG <= 256
    assert KG * Ks == Gp and Ks % 16 == 0                       # k slices tile the padded gate dimension in 16-wide groups
    assert NG * Ns >= H and (NG - 1) * Ns < H                   # column slices cover the hidden units, none is empty
    assert 1 <= NT <= 6 and 16 * NT >= Ns
    assert (16 * Ns + KG - 1) // KG <= 128                      # a stream's elements per workgroup fit its two waves
    assert Ks * 16 * NT * 4 + 4 * 4 * NT * 64 * 4 <= 156 * 1024   # W block + hand-over buffers in LDS


def test_headline_shapes_are_taken_and_batches_over_64_are_not():
    out = (ctypes.c_int32 * 5)()
    for H in (1150, 400):                                       # the AWD-LSTM layers of BASELINE configs[3]
        assert lib.nnl_debug_lstm_bptt2_plan(64, H, out) == 1
    assert lib.nnl_debug_lstm_bptt2_plan(65, 1150, out) == 0     # four 16-row streams = 64 batch rows at most
