"""Timing experiments on csrc/wino2s.hip (NNL_W2S_DBG: bit 0 no raw traffic, bit 1 no U traffic, bit 2 no MFMA; results invalid)."""
import json, os, sys
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check
from tools.bench_wino2s import timed

dev = torch.device('cuda:0')
counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
shapes = [('l1', 64, 64, 64, 56), ('l3', 64, 256, 256, 14), ('p3', 16, 256, 256, 64)]
for name, N, Cc, K, H in shapes:
    x = torch.randn(N, H, H, Cc, device=dev); w = torch.randn(K, 3, 3, Cc, device=dev) * 0.05; y = torch.empty(N, H, H, K, device=dev)
    row = {'layer': name}
    for dbg in (0, 3, 8):
        os.environ['NNL_W2S_DBG'] = str(dbg); lib.nnl_reload_env()
        wsb = lib.nnl_debug_conv_wino2s_workspace_bytes(N, H, H, Cc, K)
        ws = torch.empty(wsb // 4, device=dev)
        f = lambda: check(lib.nnl_debug_conv_wino2s_fwd(ptr(x), ptr(w), None, None, ptr(y), ptr(ws), wsb, ptr(counters), counters.numel(), None, None, N, H, H, Cc, K, 0, 0, stream()))
        row['dbg%d_us' % dbg] = round(timed(f), 1)
    print(json.dumps(row), flush=True)
