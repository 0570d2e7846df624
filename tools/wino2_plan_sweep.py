"""2-D Winograd kernel (csrc/wino2.hip): launch time per forced schedule (NNL_WINO_PLAN_KS main k slices x NNL_WINO_PLAN_S tail slices,
NNL_WINO2_BK) against the planner's own choice, per ResNet-34 stage shape.   python tools/wino2_plan_sweep.py [--bs 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check


def timed(fn, n=20, warm=3):
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
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
    N = args.bs
    for name, Cc, K, H in [('l1', 64, 64, 56), ('l2', 128, 128, 28), ('l3', 256, 256, 14), ('l4', 512, 512, 7), ('fpn', 256, 256, 32)]:
        x = torch.randn(N, H, H, Cc, device=dev)
        w = torch.randn(K, 3, 3, Cc, device=dev) / (Cc * 9) ** 0.5
        y = torch.empty(N, H, H, K, device=dev)

        def measure():
            lib.nnl_reload_env()
            wsb = lib.nnl_debug_conv_wino2_workspace_bytes(N, H, H, Cc, K)
            ws = torch.empty(wsb // 4 + 4, device=dev)
            return timed(lambda: check(lib.nnl_debug_conv_wino2_fwd(ptr(x), ptr(w), None, None, ptr(y), ptr(ws), wsb, ptr(counters), counters.numel(),
                                                                    None, None, N, H, H, Cc, K, 0, 0, stream())))
        for bk in ('16', '32'):
            os.environ['NNL_WINO2_BK'] = bk
            os.environ.pop('NNL_WINO_PLAN_KS', None); os.environ.pop('NNL_WINO_PLAN_S', None)
            res = ['auto %.1f' % measure()]
            for ks in (1, 2, 4):
                for S in (1, 2, 3, 4, 6, 8, 12):
                    os.environ['NNL_WINO_PLAN_KS'] = str(ks); os.environ['NNL_WINO_PLAN_S'] = str(S)
                    try:
                        res.append('%dx%d %.1f' % (ks, S, measure()))
                    except Exception as e:
                        res.append('%dx%d err' % (ks, S))
            print(name, 'bs', N, 'bk', bk, ' | '.join(res), flush=True)
    for k in ('NNL_WINO2_BK', 'NNL_WINO_PLAN_KS', 'NNL_WINO_PLAN_S'):
        os.environ.pop(k, None)


if __name__ == '__main__':
    main()
