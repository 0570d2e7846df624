"""Time ops.linear forward + backward for the structured-data MLP's GEMM shapes as they are (C = 203 / 1000 / 500: not multiples
of 16) and zero-padded to the next multiple of 16 / 32 — how much does the first-generation kernel cost these layers?
python tools/bench_linear_shapes.py"""
import json
import sys
import torch
sys.path.insert(0, '.')
from neuralnetworklibrary_amd import ops


def timed(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


def main():
    dev = torch.device('cuda:0')
    out = []
    for M, C, K in [(1024, 203, 1000), (1024, 204, 1000), (1024, 208, 1000), (1024, 224, 1000), (1024, 1000, 500), (1024, 1008, 500), (1024, 1024, 500),
                    (1024, 500, 1), (1024, 512, 1), (1024, 512, 4)]:
        x = torch.randn(M, C, device=dev, requires_grad=True)
        w = torch.randn(K, C, device=dev, requires_grad=True)
        b = torch.randn(K, device=dev, requires_grad=True)
        dy = torch.randn(M, K, device=dev)

        def fwd():
            return ops.linear(x, w, b, relu=True)

        def fwd_bwd():
            y = ops.linear(x, w, b, relu=True)
            y.backward(dy)
            x.grad = w.grad = b.grad = None

        with torch.no_grad():
            tf = timed(fwd)
        tfb = timed(fwd_bwd)
        # the same under a captured graph (host launch cost excluded)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            fwd_bwd()
        tg = timed(g.replay)
        out.append({'M': M, 'C': C, 'K': K, 'fwd_us': round(tf, 1), 'fwd_bwd_us': round(tfb, 1), 'fwd_bwd_graph_us': round(tg, 1)})
        print(json.dumps(out[-1]), flush=True)


if __name__ == '__main__':
    main()
