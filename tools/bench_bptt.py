"""BPTT of one LSTM layer alone (nnl_lstm_bwd on random tapes): us per timestep on the path NNL_LSTM_PERSIST selects.
Usage: python tools/bench_bptt.py [H=1150] [B=64] [T=70]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd._lib import lib, ptr, check, stream  # noqa: E402
from neuralnetworklibrary_amd.ops import lstm_timeout_flag  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 70
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
Gp = int(lib.nnl_lstm_padded_gates(H))
r = lambda *s: torch.rand(*s, device=dev, generator=g)
dy, gates, cy, c0 = r(T, B, H) - 0.5, r(T, B, 4 * H), r(T, B, H) - 0.5, r(B, H) - 0.5
w_t = torch.zeros(H, Gp, device=dev)
w_t[:, :4 * H] = (r(H, 4 * H) - 0.5) / H ** 0.5
dh0, dc0 = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
wsb = int(lib.nnl_lstm_workspace_bytes(T, B, H))
ws = torch.empty(wsb // 4, device=dev)
flag = lstm_timeout_flag(dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
best = 1e9
for i in range(8):
    dgates = torch.zeros(T, B, Gp, device=dev)
    ev[0].record()
    check(lib.nnl_lstm_bwd(ptr(dy), None, None, ptr(gates), ptr(cy), ptr(c0), ptr(w_t), ptr(dgates), ptr(dh0), ptr(dc0), T, B, H, ptr(ws), wsb, ptr(flag), stream()))
    ev[1].record()
    torch.cuda.synchronize()
    if i >= 2:
        best = min(best, ev[0].elapsed_time(ev[1]))
print('H %d B %d T %d persist %s dbg %s kg/ng %s/%s: %.3f ms = %.1f us per step  (flag %d, ws %.0f MB)' % (
    H, B, T, os.environ.get('NNL_LSTM_PERSIST', '5'), os.environ.get('NNL_LSTM_BPTT2_DBG', '0'), os.environ.get('NNL_LSTM_BPTT2_KG', '-'),
    os.environ.get('NNL_LSTM_BPTT2_NG', '-'), best, best / T * 1e3, int(flag.item()), wsb / 1e6))
