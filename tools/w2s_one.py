"""One layer shape through the debug entries of csrc/wino2s.hip and csrc/wino2.hip, a few launches each: the target of rocprofv3 --pmc passes.
python3 tools/w2s_one.py l1 [reps]"""
import sys
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check

SH = {'l1': (64, 64, 64, 56), 'l2': (64, 128, 128, 28), 'l3': (64, 256, 256, 14), 'l4': (64, 512, 512, 7), 'p3': (16, 256, 256, 64)}
N, Cc, K, H = SH[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
x = torch.randn(N, H, H, Cc, device=dev); w = torch.randn(K, 3, 3, Cc, device=dev) * 0.05; y = torch.empty(N, H, H, K, device=dev)
for f, fw in ((lib.nnl_debug_conv_wino2s_fwd, lib.nnl_debug_conv_wino2s_workspace_bytes), (lib.nnl_debug_conv_wino2_fwd, lib.nnl_debug_conv_wino2_workspace_bytes)):
    wsb = fw(N, H, H, Cc, K)
    ws = torch.empty(wsb // 4, device=dev)
    for _ in range(reps):
        check(f(ptr(x), ptr(w), None, None, ptr(y), ptr(ws), wsb, ptr(counters), counters.numel(), None, None, N, H, H, Cc, K, 0, 0, stream()))
    torch.cuda.synchronize()
