"""Spatially staged 2-D Winograd kernel (csrc/wino2s.hip): launch time per forced schedule (NNL_W2S_UNIT main-round unit, NNL_WINO_PLAN_KS
main k slices, NNL_WINO_PLAN_S tail slices) against the planner's own choice.   python tools/wino2s_plan_sweep.py [--bs 64]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd._lib import lib, ptr, stream, check
from tools.bench_wino2s import timed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--shapes', default='l1,l2,l3,l4,fpn')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    counters = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
    N = args.bs
    allsh = {'l1': (64, 64, 56), 'l2': (128, 128, 28), 'l3': (256, 256, 14), 'l4': (512, 512, 7), 'fpn': (256, 256, 32), 'p3': (256, 256, 64), 'b1': (64, 64, 128), 'b2': (128, 128, 64)}
    for name in args.shapes.split(','):
        Cc, K, H = allsh[name]
        x = torch.randn(N, H, H, Cc, device=dev)
        w = torch.randn(K, 3, 3, Cc, device=dev) / (Cc * 9) ** 0.5
        y = torch.empty(N, H, H, K, device=dev)

        def measure():
            lib.nnl_reload_env()
            wsb = lib.nnl_debug_conv_wino2s_workspace_bytes(N, H, H, Cc, K)
            ws = torch.empty(wsb // 4 + 4, device=dev)
            return timed(lambda: check(lib.nnl_debug_conv_wino2s_fwd(ptr(x), ptr(w), None, None, ptr(y), ptr(ws), wsb, ptr(counters), counters.numel(),
                                                                     None, None, N, H, H, Cc, K, 0, 0, stream())), n=25, warm=8)
        for _ in range(40):                              # clocks / power state: the first second of work on an idle GPU runs 5-15 % slow
            measure()
        for k in ('NNL_W2S_UNIT', 'NNL_WINO_PLAN_KS', 'NNL_WINO_PLAN_S'):
            os.environ.pop(k, None)
        row = {'shape': name, 'N': N, 'C': Cc, 'K': K, 'H': H, 'auto': round(measure(), 1)}
        for unit in (256, 512):
            os.environ['NNL_W2S_UNIT'] = str(unit)
            for ks in (1, 2, 4):
                for S in (1, 2, 3, 4, 6, 8, 12, 16):
                    os.environ['NNL_WINO_PLAN_KS'] = str(ks); os.environ['NNL_WINO_PLAN_S'] = str(S)
                    try:
                        row['%d:%dx%d' % (unit, ks, S)] = round(measure(), 1)
                    except Exception:
                        pass
        print(json.dumps(row), flush=True)
    for k in ('NNL_W2S_UNIT', 'NNL_WINO_PLAN_KS', 'NNL_WINO_PLAN_S'):
        os.environ.pop(k, None)


if __name__ == '__main__':
    main()
