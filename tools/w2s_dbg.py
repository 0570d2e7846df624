"""Timing experiments on csrc/wino2s.hip (NNL_W2S_DBG: bit 0 no raw traffic, bit 1 no U traffic, bit 2 no MFMA, 8 prologue + epilogue only, 32
relaxed ring wait; results invalid).  `lone` shapes run one workgroup per CU (plain grid, <= 256 tiles)."""
import json, os, sys
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check
from tools.bench_wino2s import timed

dev = torch.device('cuda:0')
counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
a = torch.randn(4096, 4096, device=dev)
for _ in range(60):
    a = torch.tanh(a @ a * 1e-3)
shapes = [('l1', 64, 64, 64, 56, None), ('p3', 16, 256, 256, 64, None), ('lone_l3b32', 32, 256, 256, 14, (1, 1)), ('lone_p4b4', 4, 256, 256, 32, (1, 1)), ('pair_p4b8', 8, 256, 256, 32, (1, 1))]
for name, N, Cc, K, H, forced in shapes:
    x = torch.randn(N, H, H, Cc, device=dev); w = torch.randn(K, 3, 3, Cc, device=dev) * 0.05; y = torch.empty(N, H, H, K, device=dev)
    row = {'layer': name, 'tiles': ((N * ((H + 1) // 2) ** 2 + 63) // 64) * ((K + 63) // 64), 'quarters': Cc // 2}
    if forced:
        os.environ['NNL_WINO_PLAN_KS'], os.environ['NNL_WINO_PLAN_S'] = str(forced[0]), str(forced[1])
    for dbg in (0, 32, 1, 2, 3, 4, 7, 8):
        os.environ['NNL_W2S_DBG'] = str(dbg); lib.nnl_reload_env()
        wsb = lib.nnl_debug_conv_wino2s_workspace_bytes(N, H, H, Cc, K)
        ws = torch.empty(wsb // 4, device=dev)
        f = lambda: check(lib.nnl_debug_conv_wino2s_fwd(ptr(x), ptr(w), None, None, ptr(y), ptr(ws), wsb, ptr(counters), counters.numel(), None, None, N, H, H, Cc, K, 0, 0, stream()))
        row['dbg%d_us' % dbg] = round(timed(f), 1)
    os.environ.pop('NNL_WINO_PLAN_KS', None); os.environ.pop('NNL_WINO_PLAN_S', None)
    print(json.dumps(row), flush=True)
