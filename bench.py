"""Headline benchmark: Learner.train1minibatch on ResNet-34 (+ default head), 224x224, bs=64 per GPU, fp32,
synthetic data, SGD-momentum — BASELINE.json configs[1] ("DogsCats ResNet-34 classifier, 224x224 bs=64").

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank/GPU)

Prints ONE JSON line (rank 0): whole-job images/s, ms/step, plus
  roofline     — the dominant kernel family (fp32-MFMA implicit-GEMM conv fwd+dgrad+wgrad): algorithmic FLOPs per
                 launch / average launch duration, measured with HIP events recorded by libnnl_hip.so on the launch
                 stream during K extra steps of the same command, against the dense fp32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline — the CPU oracle (oracle/reference_nets.py, a torch-CPU restatement pinned to reference goldens; the
                 reference's own Python cannot travel to the GPU box) running the SAME step on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak
CONV_GFLOP_PER_IMAGE = 21.98           # 3 x 2 x 3.6638 GMAC (fwd + dgrad + wgrad)


class SynthData:
    """Device-resident synthetic batches honouring the Learner's data protocol."""
    target_type = 'single_label'
    categories = {0: 'cat', 1: 'dog'}

    def __init__(self, bs, sz, n_batches, device, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        self.bs, self.sz = bs, (sz, sz)
        self.batches = [(torch.randn(bs, 3, sz, sz, device=device, generator=g),
                         torch.randint(0, 2, (bs,), device=device, generator=g)) for _ in range(n_batches)]
        self.train_dl = self.batches
        self.val_dl = self.batches[:1]


def build_learner(device, bs, sz, seed):
    from neuralnetworklibrary_amd.Applications import Vision as V
    from neuralnetworklibrary_amd.General.Core import set_default_device
    set_default_device(device)
    torch.manual_seed(seed)
    data = SynthData(bs, sz, 4, device, seed)
    net = V.ImageClassificationNet(data, V.models.resnet34())
    learner = V.ImageLearner('/tmp/nnl_bench', data, net, optimizer='SGD_Mom')
    learner.init_optimizer(wd=1e-4)
    return learner, data


def cpu_baseline(bs, sz, steps=2):
    """The same train step on the host cores with the CPU oracle (plain torch fp32 eager + restated Optimizer.step)."""
    from oracle import reference_math as RM
    from oracle import reference_nets as RN
    # the GPU box shares its host: 16 cores is the CPU share of a 1-GPU slot (more threads thrash: 256 threads
    # measured 116 s/step vs ~4 s/step on 8 cores)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    torch.manual_seed(0)
    net = RN.ImageClassificationNet(RN.resnet34(), 2, 512).train()
    params = [p for p in net.parameters()]
    state = RM.OptimState(params)
    x, y = torch.randn(bs, 3, sz, sz), torch.randint(0, 2, (bs,))
    times = []
    for i in range(steps + 1):
        t0 = time.time()
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        RM.optimizer_step(params, [p.grad for p in params], state, [1e-2] * len(params), [1e-4] * len(params), 'sgd')
        loss.item()
        if i > 0:
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {'value': bs / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d steps of the same bs=%d %dx%d ResNet-34 train step (after 1 warm-up), torch-CPU oracle' % (steps, bs, sz, sz),
            'ms_per_step': dt * 1e3}


FLOP_KINDS = ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'gemm', 'lstm')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=64, help='per-GPU batch (weak scaling) or global batch (--scaling strong)')
    ap.add_argument('--sz', type=int, default=224)
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    force_dist = os.environ.get('NNL_BENCH_FORCE_DIST') == '1'       # exercise the RCCL path on a 1-GPU box (world_size 1)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    per_gpu_bs = args.bs if args.scaling == 'weak' else max(args.bs // world, 1)
    learner, data = build_learner(device, per_gpu_bs, args.sz, 1234 + 1 + 1000 * rank)
    if world > 1 or force_dist:
        learner.distribute()
    learner.model.train()
    lr = [1e-3, 3e-3, 1e-2]

    def step(i):
        x, y = data.batches[i % len(data.batches)]
        return learner.train1minibatch(x, y, lr)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline leg: same steps again with per-launch HIP events on the launch stream ----
    from neuralnetworklibrary_amd import _lib
    _lib.prof_enable(True)
    n_prof = min(args.steps, 10)
    for i in range(n_prof):
        step(i)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    prof = _lib.prof_collect()
    conv_ms = sum(prof[k]['ms'] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad'))
    conv_flop = sum(prof[k]['work'] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad'))
    conv_launches = sum(prof[k]['launches'] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad'))
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # HBM-side bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes of THIS command
    # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); the committed summary is profiles/r1_traffic.json
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r1_traffic.json')) as f:
            traffic = round(json.load(f)['traffic_bytes_per_launch'])
    except Exception:
        pass
    roofline = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                'kernel': 'igemm_taps_kernel / igemm_wgrad_kernel (fp32 MFMA implicit-GEMM conv2d + linear: fwd, dgrad, wgrad; incl. their slab reduces)',
                'launches_per_step': conv_launches / max(n_prof, 1),
                'avg_launch_ms': conv_ms / max(conv_launches, 1),
                'flop_per_launch': conv_flop / max(conv_launches, 1),
                'conv_ms_per_step': conv_ms / max(n_prof, 1),
                # per C-entry-point family: algorithmic FLOPs (conv / gemm / lstm) or algorithmic BYTES (the HBM-bound kinds)
                'by_kind': {k: {'ms_per_step': round(v['ms'] / max(n_prof, 1), 3),
                                ('tflops' if k in FLOP_KINDS else 'tbytes_per_s'):
                                    round(v['work'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else None}
                            for k, v in prof.items() if v['launches']}}

    if rank == 0:
        global_bs = per_gpu_bs * world
        ms = dt / args.steps * 1e3
        out = {
            'metric': 'ResNet-34 224x224 training throughput (Learner.train1minibatch, fwd+loss+bwd+optimizer)',
            'value': round(global_bs * args.steps / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'DogsCats ResNet-34 classifier, 224x224, bs=%d per GPU, SGD momentum 0.9, fp32' % per_gpu_bs,
                       'global_batch': global_bs, 'image_size': args.sz, 'parallelism': 'dp%d' % world,
                       'last_loss': loss},
            'roofline': roofline,
        }
        if learner.grad_sync is not None:                # data parallel: how the gradients reached the all-reduce buckets
            gs = learner.grad_sync
            out['config']['grad_buckets'] = len(gs.buckets)
            out['config']['grads_written_in_place'] = '%d of %d tensors per step' % (
                gs.direct_writes // max(gs.steps, 1), sum(len(b.params) for b in gs.buckets))
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(per_gpu_bs, args.sz)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
