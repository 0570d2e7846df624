"""Where does a conv launch spend its time?  Runs ONE plain (unbalanced) launch of the tap-table kernel per shape with the
timing build of the library (tools/build_timing_lib.sh; four 100 MHz timestamps per workgroup: entry, loop start, loop end, exit)
and prints, in microseconds: the dispatch skew of the workgroups, prologue / k loop / epilogue per workgroup, the launch span
seen from inside (first entry -> last exit) and the HIP-event time of the same launch.
Usage (GPU box): NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_timing.so NNL_IGEMM_BALANCE=0 python tools/conv_timing.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import ops  # noqa: E402
from neuralnetworklibrary_amd._lib import check, lib, ptr, stream  # noqa: E402

SHAPES = [('l1 bs64', 64, 64, 56, 64, 3, 1, 1), ('l2 bs64', 64, 128, 28, 128, 3, 1, 1), ('l3 bs64', 64, 256, 14, 256, 3, 1, 1), ('l4 bs64', 64, 512, 7, 512, 3, 1, 1), ('l1 bs8', 8, 64, 56, 64, 3, 1, 1),
          ('l2 bs8', 8, 128, 28, 128, 3, 1, 1), ('l3 bs8', 8, 256, 14, 256, 3, 1, 1), ('l4 bs8', 8, 512, 7, 512, 3, 1, 1),
          ('1x1 bs8', 8, 64, 56, 128, 1, 2, 0)]
dev = 'cuda'
cnt = ops._tile_counters(torch.device(dev))
stamps = cnt[32768:].view(torch.int64)[:3276 * 5].view(-1, 5)
print('%-9s %6s | %8s %8s | %8s %8s %8s | %8s %8s' % ('shape', 'wgs', 'skew50', 'skewmax', 'prolog', 'loop', 'epilog', 'span', 'event'))
for name, N, C, H, K, R, stride, pad in (SHAPES if 'wgrad' not in sys.argv[1:] else []):
    g = ops._geom(N, H, H, C, K, R, R, stride, pad)
    x = torch.randn(N, H, H, C, device=dev); w = torch.randn(K, R, R, C, device=dev) * 0.05
    y = torch.empty(N, g.P, g.Q, K, device=dev)
    fn = lambda: check(lib.nnl_conv2d_fwd(ptr(x), ptr(w), None, ptr(y), g, 0, None, 0, ptr(cnt), None, None, None, stream()))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stamps.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    raw = raw[raw[:, 0] > 0]
    if os.environ.get('NNL_TIMING_DUMP'):
        np.save(os.path.join(os.environ['NNL_TIMING_DUMP'], 'stamps_%s.npy' % name.replace(' ', '_')), raw)
    t = raw[:, :4].astype(np.float64) / 100.0                    # us (100 MHz counter)
    t0 = t[:, 0].min()
    print('%-9s %6d | %8.2f %8.2f | %8.2f %8.2f %8.2f | %8.2f %8.2f' % (
        name, len(t), np.median(t[:, 0] - t0), (t[:, 0] - t0).max(), np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]),
        np.median(t[:, 3] - t[:, 2]), t[:, 3].max() - t0, a.elapsed_time(b) * 1e3))


def wgrad_timelines():
    """the same for igemm_wgrad_kernel: prologue / loop / epilogue per workgroup, plus the steady-state k-loop rate of a CU"""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    lib.nnl_debug_wgrad_stamps.restype = ctypes.c_void_p
    print('%-9s %6s %-9s | %8s %8s %8s | %8s %8s | %s' % ('wgrad', 'wgs', 'plan', 'prolog', 'loop', 'epilog', 'span', 'event', 'CU k-loop rate (share of fp32-MFMA peak)'))
    for name, N, C, H, K, R, stride, pad in SHAPES:
        g = ops._geom(N, H, H, C, K, R, R, stride, pad)
        x = torch.randn(N, H, H, C, device=dev); dy = torch.randn(N, g.P, g.Q, K, device=dev); dw = torch.empty(K, R, R, C, device=dev)
        wsb = int(lib.nnl_conv2d_wgrad_workspace_bytes(g)); ws = torch.empty(max(wsb // 4, 1), device=dev)
        fn = lambda: check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(ws), wsb, stream()))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dptr = lib.nnl_debug_wgrad_stamps()
        hip.hipMemset(ctypes.c_void_p(dptr), 0, 3276 * 40)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        host = np.zeros(3276 * 5, dtype=np.int64)
        hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dptr), 3276 * 40, 2)
        raw = host.reshape(-1, 5)
        raw = raw[raw[:, 0] > 0]
        if os.environ.get('NNL_TIMING_DUMP'):
            np.save(os.path.join(os.environ['NNL_TIMING_DUMP'], 'wgrad_%s.npy' % name.replace(' ', '_')), raw)
        t = raw[:, :4].astype(np.float64) / 100.0
        t0 = t[:, 0].min()
        # steady-state rate: FLOPs of the launch / (sum over CUs of the time at least one workgroup of the CU is inside its k loop)
        hw = raw[:, 4]
        key = ((hw >> 32) & 0xf) * 100000 + (hw & 0xffff)
        busy = 0.0
        for k in np.unique(key):
            iv = sorted((s, e) for s, e in t[key == k][:, 1:3])
            cs, ce = iv[0]
            for s, e in iv[1:]:
                if s <= ce:
                    ce = max(ce, e)
                else:
                    busy += ce - cs; cs, ce = s, e
            busy += ce - cs
        flop = 2.0 * N * g.P * g.Q * K * R * R * C
        rate = flop / (busy * 1e-6) / 157.3e12 * (256.0 / len(np.unique(key))) / 256.0 * len(np.unique(key))
        print('%-9s %6d %-9s | %8.2f %8.2f %8.2f | %8.2f %8.2f | %.3f' % (
            name, len(t), '', np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]),
            t[:, 3].max() - t0, a.elapsed_time(b) * 1e3, flop / (busy * 1e-6) / (157.3e12 / 256)))


if len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
    wgrad_timelines()
