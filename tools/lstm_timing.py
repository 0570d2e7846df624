"""Phases of one timestep of the persistent LSTM forward kernel (timing build, tools/build_timing_lib.sh): for four workgroups
(first, 1/3, 2/3, last) the median over the 70 steps of: k loop + partial-sum exchange, cell + publishing h, drain + arrive,
issuing the tape stores, waiting at the grid barrier; and the step period.
Usage (GPU box): NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so python tools/lstm_timing.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import ops_text  # noqa: E402
from neuralnetworklibrary_amd._lib import lib  # noqa: E402

dev = torch.device('cuda', 0)
T, B = 70, 64
for H in (1150, 400):
    g = torch.Generator(device=dev).manual_seed(0)
    gx = torch.randn(T, B, 4 * H, device=dev, generator=g) * 0.5
    w = torch.randn(4 * H, H, device=dev, generator=g) / H ** 0.5
    h0, c0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    for _ in range(3):
        ops_text._LSTMRecurrence.apply(gx, w, h0, c0)
    torch.cuda.synchronize()
    host = np.zeros(4 * 128 * 6, dtype=np.uint64)
    assert lib.nnl_debug_lstm_stamps(host.ctypes.data_as(ctypes.c_void_p), host.size) == 0
    s = host.reshape(4, 128, 6)[:, :T - 1].astype(np.float64) / 100.0          # us; the last step has no barrier
    names = ['k loop + LDS exchange', 'cell + publish h', 'drain + arrive', 'tape stores issued', 'barrier wait']
    print('H = %d   step period (median over steps, per sampled workgroup): %s us' % (H, np.round(np.median(np.diff(s[:, :, 0], axis=1), axis=1), 2)))
    for i, n in enumerate(names):
        print('   %-24s %s' % (n, np.round(np.median(s[:, :, i + 1] - s[:, :, i], axis=1), 2)))
