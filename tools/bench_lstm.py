"""One LSTM layer (recurrence only) at the AWD-LSTM shapes: forward and backward time of ops_text._LSTMRecurrence.
NNL_LSTM_PERSIST=0 selects the per-timestep path (the switch is read once per process)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import ops_text  # noqa: E402

dev = torch.device('cuda', 0)
T, B = 70, int(sys.argv[1]) if len(sys.argv) > 1 else 64
for H in (1150, 400):
    g = torch.Generator(device=dev).manual_seed(0)
    gx = (torch.randn(T, B, 4 * H, device=dev, generator=g) * 0.5).requires_grad_(True)
    w = (torch.randn(4 * H, H, device=dev, generator=g) / H ** 0.5).requires_grad_(True)
    h0, c0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    n = 6
    for i in range(n + 2):
        gx.grad = w.grad = None
        ev[0].record()
        y, hT, cT = ops_text._LSTMRecurrence.apply(gx, w, h0, c0)
        ev[1].record()
        (y.sum() + cT.sum()).backward()
        ev[2].record()
        torch.cuda.synchronize()
        if i >= 2:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print(json.dumps({'H': H, 'B': B, 'persist': os.environ.get('NNL_LSTM_PERSIST', '5'), 'fwd_ms': round(tf / n, 3),
                      'fwd_us_per_step': round(tf / n / T * 1e3, 1), 'bwd_ms_incl_dW_gemm': round(tb / n, 3)}))
