"""The spatially staged 2-D Winograd kernel (csrc/wino2s.hip) against the 2-D kernel it succeeds (csrc/wino2.hip) and float64, through
the debug entries: bias / ReLU / addend / BatchNorm partials / flipped dgrad filter / odd sizes / ragged channel counts / forced k
slicing, bitwise run-to-run repeatability, and kernel time per layer shape.   python tools/bench_wino2s.py [--bs 64] [--net r34|r50|checks]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check


def timed(fn, n=30, warm=8):
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


SHAPES = {
    'r34': [('l1', 64, 64, 56, 56), ('l2', 128, 128, 28, 28), ('l3', 256, 256, 14, 14), ('l4', 512, 512, 7, 7)],
    'r50': [('b1', 64, 64, 128, 128), ('b2', 128, 128, 64, 64), ('b3', 256, 256, 32, 32), ('b4', 512, 512, 16, 16),
            ('p3', 256, 256, 64, 64), ('p4', 256, 256, 32, 32), ('p5', 256, 256, 16, 16)],
    'checks': [('odd', 64, 128, 9, 9), ('rect', 16, 80, 11, 14), ('k80', 48, 80, 8, 13), ('wide', 16, 64, 2, 70), ('l4', 512, 512, 7, 7),
               ('fpn', 256, 256, 32, 32), ('c16', 16, 32, 10, 7)],
}


def one(name, N, Cc, K, H, W, dev, counters, timing=True):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, H, W, Cc, device=dev, generator=g)
    w = torch.randn(K, 3, 3, Cc, device=dev, generator=g) / (Cc * 9) ** 0.5
    b = torch.randn(K, device=dev, generator=g)
    addt = torch.randn(N, H, W, K, device=dev, generator=g)
    piv = torch.randn(K, device=dev, generator=g) * 0.1
    out = {'layer': name, 'N': N, 'C': Cc, 'K': K, 'H': H, 'W': W}
    res = {}
    for tag, f, fw in (('w2', lib.nnl_debug_conv_wino2_fwd, lib.nnl_debug_conv_wino2_workspace_bytes),
                       ('w2s', lib.nnl_debug_conv_wino2s_fwd, lib.nnl_debug_conv_wino2s_workspace_bytes)):
        wsb = max(fw(N, H, W, Cc, K), fw(N, H, W, K, Cc))
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        nrows = (N * ((H + 1) // 2) * ((W + 1) // 2) + 63) // 64
        part = torch.zeros(nrows, K, 2, device=dev)
        y = torch.empty(N, H, W, K, device=dev)

        def run(relu=1, add=None, bn=False, flip=0, filt=w, bias=b, xin=x, o=y, cc=Cc, kk=K):
            check(f(ptr(xin), ptr(filt), ptr(bias), ptr(add), ptr(o), ptr(ws), wsb, ptr(counters), counters.numel(),
                    ptr(part) if bn else None, ptr(piv) if bn else None, N, H, W, cc, kk, relu, flip, stream()))

        run()
        y1 = y.clone()
        run()
        torch.cuda.synchronize()
        rep = bool((y == y1).all())
        run(relu=0, add=addt, bn=True)
        y2 = y.clone()
        p2 = part.clone()
        dy = torch.randn(N, H, W, K, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        wt = w.permute(3, 1, 2, 0).contiguous()                        # [C][R][S][K]
        dx = torch.empty(N, H, W, Cc, device=dev)
        run(relu=0, flip=1, filt=wt, bias=None, xin=dy, o=dx, cc=K, kk=Cc)
        torch.cuda.synchronize()
        assert int(counters.abs().sum()) == 0, 'tile counters not back to zero'
        res[tag] = (y1, y2, p2, dx.clone(), rep)
        if timing:
            t = timed(lambda: run())
            out[tag + '_us'] = round(t, 1)
            out[tag + '_tf'] = round(2.0 * N * H * W * K * 9 * Cc / t / 1e6, 1)
    a, bb = res['w2'], res['w2s']
    out['rep'] = bb[4]
    out['d_relu'] = (a[0] - bb[0]).abs().max().item()
    out['d_add'] = (a[1] - bb[1]).abs().max().item()
    out['d_dgrad'] = (a[3] - bb[3]).abs().max().item()
    out['dgrad_scale'] = a[3].abs().max().item()
    s_a, s_b = a[2].double().sum(0), bb[2].double().sum(0)
    out['d_bn'] = ((s_a - s_b).abs().max() / s_a.abs().max()).item()
    xs, ws_ = x[:1].permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu()
    r64 = torch.relu(torch.nn.functional.conv2d(xs, ws_, b.double().cpu(), padding=1)).permute(0, 2, 3, 1)
    out['w2s_err64'] = (bb[0][:1].double().cpu() - r64).abs().max().item()
    out['w2_err64'] = (a[0][:1].double().cpu() - r64).abs().max().item()
    if timing and 'w2_us' in out:
        out['speedup'] = round(out['w2_us'] / out['w2s_us'], 3)
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--net', default='r34')
    ap.add_argument('--no-time', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
    if not args.no_time:                           # clocks / power state: the first second of work on an idle GPU runs 5-15 % slow
        a = torch.randn(4096, 4096, device=dev)
        for _ in range(60):
            a = torch.tanh(a @ a * 1e-3)
        torch.cuda.synchronize()
    for net in args.net.split(','):
        for name, Cc, K, H, W in SHAPES[net]:
            N = args.bs if net != 'checks' else 3
            one(name, N, Cc, K, H, W, dev, counters, timing=not args.no_time)
    if 'checks' in args.net:                       # forced k slicing on a small problem: every split path of the kernel
        for ks, S in ((2, 1), (1, 3), (2, 4), (4, 8)):
            os.environ['NNL_WINO_PLAN_KS'] = str(ks); os.environ['NNL_WINO_PLAN_S'] = str(S); lib.nnl_reload_env()
            print('forced', ks, S)
            one('l2', 5, 128, 128, 28, 28, dev, counters, timing=False)
            one('bal', 70, 64, 64, 14, 14, dev, counters, timing=False)
        os.environ.pop('NNL_WINO_PLAN_KS'); os.environ.pop('NNL_WINO_PLAN_S'); lib.nnl_reload_env()


if __name__ == '__main__':
    main()
