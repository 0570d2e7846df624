"""Fused 1-D Winograd F(2,3) 3x3 convolution (csrc/wino.hip) against the direct implicit-GEMM kernel: correctness (bias / ReLU /
addend / BatchNorm partial statistics / flipped dgrad filter / odd width; float64 reference on one image) and kernel time per
ResNet-34 stage shape, through the debug entry nnl_debug_conv_wino_fwd.   python tools/bench_wino.py [--bs 64]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, '.')
from neuralnetworklibrary_amd import ops
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--two-d', action='store_true', help='the 2-D F(2x2,3x3) kernel (csrc/wino2.hip)')
    args = ap.parse_args()
    f = lib.nnl_debug_conv_wino2_fwd if args.two_d else lib.nnl_debug_conv_wino_fwd      # (signatures: neuralnetworklibrary_amd/_lib.py)
    fw = lib.nnl_debug_conv_wino2_workspace_bytes if args.two_d else lib.nnl_debug_conv_wino_workspace_bytes
    dev = torch.device('cuda:0')
    counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
    for name, Cc, K, H in [('l1', 64, 64, 56), ('l2', 128, 128, 28), ('l3', 256, 256, 14), ('l4', 512, 512, 7), ('odd', 64, 128, 9), ('fpn', 256, 256, 32)]:
        N = args.bs
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(N, Cc, H, H, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, Cc, 3, 3, device=dev, generator=g) / (Cc * 9) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(K, device=dev, generator=g)
        addt = torch.randn(N, H, H, K, device=dev, generator=g)
        piv = torch.randn(K, device=dev, generator=g) * 0.1
        xn, wn = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1)          # NHWC / KRSC views of the same memory
        assert xn.is_contiguous() and wn.is_contiguous()
        y = torch.empty(N, H, H, K, device=dev)
        wsb = fw(N, H, H, Cc, K)
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        nrows = (N * ((H + 1) // 2 if args.two_d else H) * ((H + 1) // 2) + 63) // 64
        part = torch.zeros(nrows, K, 2, device=dev)

        def run(relu=1, add=None, bn=False, flip=0, filt=wn, bias=b, xin=xn, out=y, cc=Cc, kk=K):
            check(f(ptr(xin), ptr(filt), ptr(bias), ptr(add), ptr(out), ptr(ws), wsb, ptr(counters), counters.numel(),
                    ptr(part) if bn else None, ptr(piv) if bn else None, N, H, H, cc, kk, relu, flip, stream()))

        def direct(fn):                      # the references: the DIRECT implicit-GEMM kernel (NNL_CONV_WINO=0 around the call)
            os.environ['NNL_CONV_WINO'] = '0'; lib.nnl_reload_env()
            try:
                with torch.no_grad():
                    return fn()
            finally:
                os.environ['NNL_CONV_WINO'] = '1'; lib.nnl_reload_env()

        ref = direct(lambda: ops.conv2d(x, w, b, 1, 1, relu=True).permute(0, 2, 3, 1).contiguous())
        flops = 2.0 * N * H * H * K * 9 * Cc
        out = {'layer': name, 'N': N, 'C': Cc, 'K': K, 'H': H, 'ws_MB': round(wsb / 2 ** 20, 1)}
        run()
        torch.cuda.synchronize()
        out['diff_vs_direct'] = (y - ref).abs().max().item()
        # addend + BatchNorm partials, no ReLU
        run(relu=0, add=addt, bn=True)
        ref2 = direct(lambda: ops.conv2d(x, w, b, 1, 1, relu=False).permute(0, 2, 3, 1) + addt)
        out['diff_add'] = (y - ref2).abs().max().item()
        d = (ref2 - piv).reshape(-1, K).double()
        s1, s2 = part[:, :, 0].double().sum(0), part[:, :, 1].double().sum(0)
        out['bn_s1_rel'] = ((s1 - d.sum(0)).abs().max() / d.abs().sum(0).max()).item()
        out['bn_s2_rel'] = ((s2 - (d * d).sum(0)).abs().max() / (d * d).sum(0).max()).item()
        assert int(counters.abs().sum()) == 0, 'tile counters not back to zero'
        # the dgrad filter: dx = conv(dy, flip(W^T)); dy has K channels, dx has C
        dy = torch.randn(N, H, H, K, device=dev, generator=g)
        wt = w.permute(1, 2, 3, 0).contiguous()                      # [C][R][S][K]
        dx = torch.empty(N, H, H, Cc, device=dev)
        wsb2 = fw(N, H, H, K, Cc)
        if wsb2 > wsb:
            ws = torch.empty(wsb2 // 4, device=dev); wsb = wsb2
        run(relu=0, flip=1, filt=wt, bias=None, xin=dy, out=dx, cc=K, kk=Cc)
        refdx = torch.nn.grad.conv2d_input((N, Cc, H, H), w, dy.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        out['diff_dgrad_vs_torch'] = (dx - refdx).abs().max().item()
        out['dgrad_scale'] = refdx.abs().max().item()
        t = timed(lambda: run())
        out['wino_us'] = round(t, 1)
        out['wino_tflops_algorithmic'] = round(flops / t / 1e6, 1)
        td = direct(lambda: timed(lambda: ops.conv2d(x, w, b, 1, 1, relu=True)))
        out['direct_us_incl_host'] = round(td, 1)
        r64 = torch.relu(torch.nn.functional.conv2d(x[:1].double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)).permute(0, 2, 3, 1)
        run()
        out['wino_err_vs_f64'] = (y[:1].double().cpu() - r64).abs().max().item()
        out['direct_err_vs_f64'] = (ref[:1].double().cpu() - r64).abs().max().item()
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
