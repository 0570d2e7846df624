"""Per-layer microbenchmark of the implicit-GEMM conv kernels at ResNet-34 bs=64 224^2 geometries.
Usage (GPU box): python tools/bench_conv.py [--bs 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralnetworklibrary_amd import _lib, ops  # noqa: E402
from neuralnetworklibrary_amd._lib import check, lib, ptr, stream  # noqa: E402

LAYERS = [  # (name, C, H, K, R, stride, pad, count in ResNet-34)
    ('stem7x7', 4, 224, 64, 7, 2, 3, 1),
    ('l1_3x3', 64, 56, 64, 3, 1, 1, 6),
    ('l2_3x3s2', 64, 56, 128, 3, 2, 1, 1),
    ('l2_1x1s2', 64, 56, 128, 1, 2, 0, 1),
    ('l2_3x3', 128, 28, 128, 3, 1, 1, 7),
    ('l3_3x3s2', 128, 28, 256, 3, 2, 1, 1),
    ('l3_1x1s2', 128, 28, 256, 1, 2, 0, 1),
    ('l3_3x3', 256, 14, 256, 3, 1, 1, 11),
    ('l4_3x3s2', 256, 14, 512, 3, 2, 1, 1),
    ('l4_1x1s2', 256, 14, 512, 1, 2, 0, 1),
    ('l4_3x3', 512, 7, 512, 3, 1, 1, 5),
]


# RetinaNet R50-FPN at 512x512 (BASELINE configs[4], bs 16): (name, C, H, K, R, stride, pad, launches per step and direction)
LAYERS_R50 = [
    ('s1_1x1_64_64', 64, 128, 64, 1, 1, 0, 1), ('s1_1x1_256_64', 256, 128, 64, 1, 1, 0, 2), ('s1_3x3_64', 64, 128, 64, 3, 1, 1, 3),
    ('s1_1x1_64_256', 64, 128, 256, 1, 1, 0, 4),
    ('s2_1x1_256_128', 256, 128, 128, 1, 1, 0, 1), ('s2_3x3s2_128', 128, 128, 128, 3, 2, 1, 1), ('s2_1x1_512_128', 512, 64, 128, 1, 1, 0, 3),
    ('s2_3x3_128', 128, 64, 128, 3, 1, 1, 3), ('s2_1x1_128_512', 128, 64, 512, 1, 1, 0, 4), ('s2_ds_256_512', 256, 128, 512, 1, 2, 0, 1),
    ('s3_1x1_512_256', 512, 64, 256, 1, 1, 0, 1), ('s3_3x3s2_256', 256, 64, 256, 3, 2, 1, 1), ('s3_1x1_1024_256', 1024, 32, 256, 1, 1, 0, 5),
    ('s3_3x3_256', 256, 32, 256, 3, 1, 1, 5), ('s3_1x1_256_1024', 256, 32, 1024, 1, 1, 0, 6), ('s3_ds_512_1024', 512, 64, 1024, 1, 2, 0, 1),
    ('s4_1x1_1024_512', 1024, 32, 512, 1, 1, 0, 1), ('s4_3x3s2_512', 512, 32, 512, 3, 2, 1, 1), ('s4_1x1_2048_512', 2048, 16, 512, 1, 1, 0, 2),
    ('s4_3x3_512', 512, 16, 512, 3, 1, 1, 2), ('s4_1x1_512_2048', 512, 16, 2048, 1, 1, 0, 3), ('s4_ds_1024_2048', 1024, 32, 2048, 1, 2, 0, 1),
    ('fpn_lat_2048', 2048, 16, 256, 1, 1, 0, 1), ('fpn_lat_1024', 1024, 32, 256, 1, 1, 0, 1), ('fpn_lat_512', 512, 64, 256, 1, 1, 0, 1),
    ('fpn_3x3_64', 256, 64, 256, 3, 1, 1, 1), ('fpn_3x3_32', 256, 32, 256, 3, 1, 1, 1), ('fpn_3x3_16', 256, 16, 256, 3, 1, 1, 1),
    ('fpn_p6', 2048, 16, 256, 3, 2, 1, 1),
    ('head_64', 256, 64, 256, 3, 1, 1, 8), ('head_32', 256, 32, 256, 3, 1, 1, 8), ('head_16', 256, 16, 256, 3, 1, 1, 8),
    ('head_8', 256, 8, 256, 3, 1, 1, 8), ('head_4', 256, 4, 256, 3, 1, 1, 8), ('out_clas_64', 256, 64, 180, 3, 1, 1, 1),
]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def ab(args):
    """Interleaved A/B of a tuning env var (the library re-reads it per call): per layer and pass, median of 5 rounds."""
    import statistics
    var, vals = args.ab.split('=')
    vals = vals.split(',')
    names = var.split('+')                          # several variables per setting: "A+B=0+1,3+2" sets A=0 B=1, then A=3 B=2
    dev, N = 'cuda', args.bs
    print('%-10s %-6s ' % ('layer', 'pass') + ' '.join('%s=%-4s ms   TF/s |' % (var[-6:], v) for v in vals))
    totals = {v: 0.0 for v in vals}
    for name, C, H, K, R, stride, pad, count in LAYERS:
        g = ops._geom(N, H, H, C, K, R, R, stride, pad)
        x = torch.randn(N, H, H, C, device=dev); w = torch.randn(K, R, R, C, device=dev) * 0.05
        y = torch.empty(N, g.P, g.Q, K, device=dev); dy = torch.randn(N, g.P, g.Q, K, device=dev)
        wt = torch.empty(C, R, R, K, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
        big = torch.empty(64 << 20, device=dev)             # 256 MB: the workspace queries are re-run per env value (plans change)
        check(lib.nnl_conv2d_weight_transpose(ptr(w), ptr(wt), K, R, R, C, stream()))
        flop = 2.0 * N * g.P * g.Q * K * R * R * C
        fns = {'fwd': lambda: check(lib.nnl_conv2d_fwd(ptr(x), ptr(w), None, ptr(y), g, 0, ptr(big), int(lib.nnl_conv2d_fwd_workspace_bytes(g)), ptr(ops._tile_counters(x.device)), None, None, None, stream())),
               'dgrad': lambda: check(lib.nnl_conv2d_dgrad(ptr(dy), ptr(wt), ptr(dx), g, None, ptr(big), int(lib.nnl_conv2d_dgrad_workspace_bytes(g)), ptr(ops._tile_counters(x.device)), stream())),
               'wgrad': lambda: check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(big), int(lib.nnl_conv2d_wgrad_workspace_bytes(g)), stream()))}
        for pname, fn in fns.items():
            if name == 'stem7x7' and pname == 'dgrad':
                continue
            res = {v: [] for v in vals}
            for _ in range(5):
                for v in vals:
                    for nm, vv in zip(names, v.split('+')):
                        os.environ[nm] = vv
                    _lib.lib.nnl_reload_env()
                    res[v].append(timeit(fn, iters=5))
            med = {v: statistics.median(res[v]) for v in vals}
            for v in vals:
                totals[v] += med[v] * count
            print('%-10s %-6s ' % (name, pname) + ' '.join('%10.3f %7.1f |' % (med[v], flop / med[v] / 1e9) for v in vals))
    print('ResNet-34 conv total ms/step: ' + '  '.join('%s=%s: %.2f' % (var, v, totals[v]) for v in vals))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--net', default='r34', help="'r34' (ResNet-34 at 224, default) or 'r50' (RetinaNet R50-FPN at 512)")
    ap.add_argument('--ab', default=None, help='A/B in ONE process, interleaved: ENVVAR=v0,v1[,v2] (e.g. NNL_IGEMM_BK32=0,1); several variables per setting: A+B=0+1,3+2')
    ap.add_argument('--only', default=None, help='comma-separated layer-name substrings to keep (e.g. 3x3)')
    args = ap.parse_args()
    global LAYERS
    if args.net == 'r50':
        LAYERS = LAYERS_R50
    if args.only:
        LAYERS = [l for l in LAYERS if any(k in l[0] for k in args.only.split(','))]
    if args.ab:
        return ab(args)
    dev = 'cuda'
    tot = {'fwd': 0., 'dgrad': 0., 'wgrad': 0.}
    totf = 0.
    print('%-10s %9s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GFLOP', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
    for name, C, H, K, R, stride, pad, count in LAYERS:
        N = args.bs
        g = ops._geom(N, H, H, C, K, R, R, stride, pad)
        x = torch.randn(N, H, H, C, device=dev)
        w = torch.randn(K, R, R, C, device=dev) * 0.05
        y = torch.empty(N, g.P, g.Q, K, device=dev)
        dy = torch.randn(N, g.P, g.Q, K, device=dev)
        wt = torch.empty(C, R, R, K, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        fwsb = int(lib.nnl_conv2d_fwd_workspace_bytes(g)); fws = torch.empty(max(fwsb // 4, 1), device=dev)
        dwsb = int(lib.nnl_conv2d_dgrad_workspace_bytes(g)); dws = torch.empty(max(dwsb // 4, 1), device=dev)
        wsb = int(lib.nnl_conv2d_wgrad_workspace_bytes(g))
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        flop = 2.0 * N * g.P * g.Q * K * R * R * C
        t_f = timeit(lambda: check(lib.nnl_conv2d_fwd(ptr(x), ptr(w), None, ptr(y), g, 0, ptr(fws), fwsb, ptr(ops._tile_counters(x.device)), None, None, None, stream())))
        check(lib.nnl_conv2d_weight_transpose(ptr(w), ptr(wt), K, R, R, C, stream()))
        t_d = timeit(lambda: check(lib.nnl_conv2d_dgrad(ptr(dy), ptr(wt), ptr(dx), g, None, ptr(dws), dwsb, ptr(ops._tile_counters(x.device)), stream())))
        t_w = timeit(lambda: check(lib.nnl_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), g, ptr(ws), wsb, stream())))
        print('%-10s %9.2f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % (
            name, flop / 1e9, t_f, flop / t_f / 1e9, t_d, flop / t_d / 1e9, t_w, flop / t_w / 1e9))
        tot['fwd'] += t_f * count
        tot['wgrad'] += t_w * count
        if name != 'stem7x7':
            tot['dgrad'] += t_d * count
        totf += flop * count
    print('ResNet-34 conv totals @bs=%d: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms; sum %.2f ms; %.1f GFLOP fwd' % (
        args.bs, tot['fwd'], tot['dgrad'], tot['wgrad'], sum(tot.values()), totf / 1e9))
    print('aggregate: %.1f TFLOP/s (3 passes)' % (3 * totf / sum(tot.values()) / 1e9))


if __name__ == '__main__':
    main()
