"""Benchmark of the Learner.fit() hot path (BASELINE.json): Learner.train1minibatch — forward + loss + backward +
Optimizer.step + the per-step loss read-back — on synthetic device-resident batches of the five BASELINE configs.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts N worker processes itself (one rank per GPU, RCCL) BEFORE it
touches the GPU and relays rank 0's line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
launcher's environment is used instead.  Rank 0 prints ONE JSON line:

  value / ms_per_step  headline, BASELINE configs[1]: ResNet-34 + default head, 224x224, 64 images PER GPU (weak scaling), fp32
  roofline             dominant kernel family of the headline (fp32-MFMA implicit-GEMM conv fwd + dgrad + wgrad): algorithmic
                       FLOPs per launch / average launch duration from HIP events that libnnl_hip.so records on the launch
                       stream during extra steps of the same command, against the dense fp32 MFMA peak (157.3 TFLOP/s)
  cpu_baseline         (N = 1) the CPU oracle (oracle/: torch-CPU restatement pinned to reference goldens — the reference's own
                       Python cannot travel to the GPU box) running the SAME step on the host cores
  strong               (N > 1) the same model at GLOBAL batch 64 (64/N images per GPU): the north-star's strong-scaling case
  dp                   (N > 1) ranks_seen (an RCCL all-reduce of ones), gradient buckets, stand-alone all-reduce time per step,
                       exposed communication and overlap fraction
  strong_scaling_proxy (N = 1) step time at bs 8 / 16 / 32 on one GPU: t(64) / t(bs) is the compute-side ceiling of strong
                       scaling at 8 / 4 / 2 GPUs
  configs              the other four BASELINE configs (collab bs 64, Rossmann-shape MLP bs 1024, AWD-LSTM bs 64 bptt 70
                       V 47 343, RetinaNet R50-FPN 512x512 bs 16), each with ms/step, whole-job samples/s, roofline, cpu_baseline
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
CONV_GFLOP_PER_IMAGE = 21.98           # 3 x 2 x 3.6638 GMAC (fwd + dgrad + wgrad)
LM_MFLOP_PER_TOKEN = 234.8             # 3 x 2 x 39.13 MMAC
RETINA_GFLOP_PER_IMAGE = 325.4         # 3 x 2 x 54.235 GMAC
RETINA_LOSS_BYTES_PER_ANCHOR = 224
FLOP_KINDS = ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'gemm', 'lstm')
ROSSMANN_CARDS = [1116, 5, 4, 13, 53, 13, 4, 8, 32, 23, 27, 24, 28, 9, 5, 5] + [10] * 16


# =====================================================================================================================
# self-launch: N ranks from a plain `python bench.py --gpus N`
# =====================================================================================================================
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """Start n copies of this command as child processes, rank r on GPU r, and relay rank 0's stdout.  The parent never
    initialises the GPU (no HIP call, no device query), and nothing is exec'ed over a process that has."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                         # one rank failed: the others would wait in a collective for ever
                    q.terminate()
    return rc


# =====================================================================================================================
# workloads
# =====================================================================================================================
class Data:
    "minimal data object of the Learner protocol (train_dl / val_dl / bs / target_type) over device-resident batches"

    def __init__(self, batches, bs, target_type, **kw):
        self.train_dl, self.val_dl, self.bs, self.target_type = batches, batches[:1], bs, target_type
        self.batches = batches
        self.__dict__.update(kw)


class Workload:
    def __init__(self, name, learner, batches, lr, unit, units_per_step, **step_kw):
        self.name, self.learner, self.batches, self.lr, self.unit = name, learner, batches, lr, unit
        self.units_per_step, self.step_kw = units_per_step, step_kw          # units per step on THIS rank
        self.loss = None

    def step(self, i):
        x, y = self.batches[i % len(self.batches)]
        self.loss = self.learner.train1minibatch(x, y, self.lr, **self.step_kw)
        return self.loss

    def step_as_in_fit(self, i):
        """The step as Learner.fit()'s inner loop makes it (General/Learner.py train_gen_sched): a replayed step hands back a handle,
        and its float is read after the NEXT step has been launched; `flush()` reads the last one — every loss of the timed
        region is read inside it."""
        x, y = self.batches[i % len(self.batches)]
        r = self.learner.train1minibatch(x, y, self.lr, _defer=True, **self.step_kw)
        if self._pending is not None:
            self.loss = self._pending.result()
            self._pending = None
        if hasattr(r, 'result'):
            self._pending = r
        else:
            self.loss = r
        return self.loss

    _pending = None

    def flush(self, _=None):
        if self._pending is not None:
            self.loss = self._pending.result()
            self._pending = None


def _learner_cls():
    from neuralnetworklibrary_amd.General.Learner import Learner
    Learner.verbose = False
    return Learner


def _finish(learner, world):
    if world > 1 or os.environ.get('NNL_BENCH_FORCE_DIST') == '1':
        learner.distribute(equal_shards=True)          # synthetic batches: every rank always holds a full shard
        if world == 1:
            # the 1-GPU rehearsal of the N > 1 legs also runs the tabular renorm sync (a no-op at world size 1 otherwise): its
            # all-gather really goes through RCCL, and the replayed step gathers before the graph (StructuredDataNet.nnl_dp_prepare)
            from neuralnetworklibrary_amd import dist as nnl_dist
            from neuralnetworklibrary_amd.ops import DistComm
            nnl_dist.enable_sync_renorm(learner.model, capacity=learner.data.bs, comm=DistComm)
    learner.model.train()
    return learner


def resnet34_workload(device, bs, seed, world, sz=224, sync_bn=False):
    "BASELINE configs[1]: DogsCats ResNet-34 classifier (Vision.py:1244-1337), SGD momentum, CE loss"
    from neuralnetworklibrary_amd.Applications import Vision as V
    torch.manual_seed(seed % 1000)                   # model init: the same on every rank (seed = base + 1000 * rank)
    g = torch.Generator(device=device).manual_seed(seed)
    batches = [(torch.randn(bs, 3, sz, sz, device=device, generator=g), torch.randint(0, 2, (bs,), device=device, generator=g))
               for _ in range(4)]
    data = Data(batches, bs, 'single_label', sz=(sz, sz), categories={0: 'cat', 1: 'dog'})
    net = V.ImageClassificationNet(data, V.models.resnet34())
    learner = V.ImageLearner('/tmp/nnl_bench', data, net, optimizer='SGD_Mom')
    learner.init_optimizer(wd=1e-4)
    _finish(learner, world)
    if sync_bn and world > 1:
        from neuralnetworklibrary_amd import dist as nd
        nd.enable_sync_bn(learner.model)
    return Workload('resnet34', learner, batches, [1e-3, 3e-3, 1e-2], 'images/s', bs)


def collab_workload(device, bs, seed, world):
    "BASELINE configs[0] shape on the GPU: ML-100K CollabFilterNet (CollabFiltering.py:168-213), D=30, Adam, wd 1e-4"
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    torch.manual_seed(seed % 1000)                   # model init: the same on every rank (seed = base + 1000 * rank)
    g = torch.Generator(device=device).manual_seed(seed)
    batches = [(torch.stack([torch.randint(0, 943, (bs,), device=device, generator=g),
                             torch.randint(0, 1682, (bs,), device=device, generator=g)], 1),
                torch.randint(1, 6, (bs,), device=device, generator=g).float()) for _ in range(4)]
    net = CollabFilterNet(943, 1682, 30, [0.8, 5.2])
    learner = _learner_cls()('/tmp/nnl_bench', Data(batches, bs, 'cont'), net, optimizer='Adam')
    learner.init_optimizer(wd=1e-4)
    return Workload('collab', _finish(learner, world), batches, 1e-2, 'samples/s', bs)


def tabular_workload(device, bs, seed, world):
    "BASELINE configs[2]: Rossmann-shape StructuredDataNet (StructuredData.py:979-1096), fc [1000,500,1], Adam, wd 1e-3"
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    torch.manual_seed(seed % 1000)                   # model init: the same on every rank (seed = base + 1000 * rank)
    rs = np.random.RandomState(seed)
    n_cont, batches = 14, []
    for _ in range(4):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=bs) for c in ROSSMANN_CARDS], 1).astype(np.int64)).to(device)
        xcont = torch.from_numpy(rs.standard_normal((bs, n_cont)).astype(np.float32)).to(device)
        y = torch.from_numpy((5 + 7 * rs.rand(bs)).astype(np.float32)).to(device)
        batches.append(([xcat, xcont], y))
    net = StructuredDataNet('cont', len(ROSSMANN_CARDS), n_cont, [{i: i for i in range(c)} for c in ROSSMANN_CARDS],
                            [1000, 500, 1], output_range=[5, 12], dropout_levels=(0.04, 0.04, [0, 0.5, 0.25]))
    learner = _learner_cls()('/tmp/nnl_bench', Data(batches, bs, 'cont'), net, optimizer='Adam')
    learner.init_optimizer(wd=1e-3)
    return Workload('tabular', _finish(learner, world), batches, [1e-3, 1e-3], 'samples/s', bs)


LM_V, LM_BPTT = 47343, 70


def lm_workload(device, bs, seed, world):
    "BASELINE configs[3]: AWD-LSTM LanguageModelNet 400/1150/3 (Text.py:611-702), V=47 343, bptt 70, Adam(0.8, 0.99), RegSeqCE(2,1)"
    from neuralnetworklibrary_amd.Applications.Text import LanguageModelNet, RegSeqCrossEntropyLoss, _Vocab
    torch.manual_seed(seed % 1000)                   # model init: the same on every rank (seed = base + 1000 * rank)
    stoi = {i: i for i in range(LM_V)}
    stoi['_pad_'] = 1
    del stoi[1]
    g = torch.Generator(device=device).manual_seed(seed)
    stream_ = torch.randint(4, LM_V, (bs, LM_BPTT * 4 + 1), device=device, generator=g)
    batches = [(stream_[:, i * LM_BPTT:(i + 1) * LM_BPTT].contiguous(), stream_[:, i * LM_BPTT + 1:(i + 1) * LM_BPTT + 1].contiguous())
               for i in range(4)]
    net = LanguageModelNet(_Vocab(stoi, bs))
    learner = _learner_cls()('/tmp/nnl_bench', Data(batches, bs, 'lang_model'), net, optimizer='Adam',
                             loss_func=RegSeqCrossEntropyLoss(2.0, 1.0))
    learner.init_optimizer(wd=1e-6, clip=0.4)
    return Workload('lm', _finish(learner, world), batches, [1e-3, 1e-3], 'tokens/s', bs * LM_BPTT, betas_batch=(0.8, 0.99))


def _retina_targets(rs, bs, M=8):
    boxes = -np.ones((bs, M, 4), np.float32)
    cats = -np.ones((bs, M), np.int64)
    for i in range(bs):
        m = rs.randint(1, M + 1)
        xy, wh = rs.uniform(0, 300, (m, 2)), rs.uniform(30, 210, (m, 2))
        boxes[i, :m] = np.concatenate([xy, xy + wh], 1)
        cats[i, :m] = rs.randint(0, 20, m)
    return boxes, cats


def retina_workload(device, bs, seed, world, sz=512):
    "BASELINE configs[4]: ObjectDetectionNet(20) = ResNet-50 + FPN + heads (Vision.py:1382-1471), SSD_loss(.5,.25,2), SGD momentum"
    from neuralnetworklibrary_amd.Applications.Vision import ObjectDetectionNet, SSD_loss
    torch.manual_seed(seed % 1000)                   # model init: the same on every rank (seed = base + 1000 * rank)
    rs = np.random.RandomState(seed)
    g = torch.Generator(device=device).manual_seed(seed)
    batches = []
    for _ in range(2):
        boxes, cats = _retina_targets(rs, bs)
        batches.append((torch.randn(bs, 3, sz, sz, device=device, generator=g),
                        [torch.from_numpy(boxes).to(device), torch.from_numpy(cats).to(device)]))
    net = ObjectDetectionNet(20)
    learner = _learner_cls()('/tmp/nnl_bench', Data(batches, bs, 'bbox'), net, optimizer='SGD_Mom', loss_func=SSD_loss(0.5, 0.25, 2.0))
    learner.init_optimizer(wd=1e-4)
    return Workload('retinanet', _finish(learner, world), batches, [1e-4, 1e-3, 1e-3], 'images/s', bs)


# =====================================================================================================================
# measurement
# =====================================================================================================================
class Clock:
    "barrier + device synchronise on both sides of a timed region; the time is the MAX over ranks"

    def __init__(self, dist, device):
        self.dist, self.device = dist, device

    def sync(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.device.type == 'cuda':
            torch.cuda.synchronize()

    def timed(self, fn, warmup, steps):
        for i in range(warmup):
            fn(i)
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        self.sync()
        dt = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([dt], device=self.device, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_median(self, fn, warmup, steps):
        """timed() plus the MEDIAN per-step time (SURVEY.md §8d protocol: median of >= 50 steps after 10 warm-ups): one event per
        step boundary recorded on torch's current stream — the stream every launch of the step is on — read after the closing
        synchronise, so the host never waits inside the loop.  Returns (whole-job seconds, MAX over ranks; median step ms, MAX)."""
        if self.device.type != 'cuda':
            return self.timed(fn, warmup, steps), None
        for i in range(warmup):
            fn(i)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        self.sync()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            fn(i)
            ev[i + 1].record()
        self.sync()
        dt = time.perf_counter() - t0
        med = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]))
        if self.dist is not None:
            t = torch.tensor([dt, med], device=self.device, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt, med = float(t[0].item()), float(t[1].item())
        return dt, med


def profile_kinds(wl, n):
    "n extra steps with per-launch HIP events on the launch stream -> {kind: {launches, ms, work}} (nnl_prof_*)"
    from neuralnetworklibrary_amd import _lib
    _lib.prof_enable(True)
    for i in range(n):
        wl.step(i)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    return {k: v for k, v in _lib.prof_collect().items() if v['launches']}


def by_kind(prof, n):
    return {k: {'ms_per_step': round(v['ms'] / max(n, 1), 3),
                ('tflops' if k in FLOP_KINDS else 'tbytes_per_s'): round(v['work'] / (v['ms'] * 1e-3) / 1e12, 3) if v['ms'] > 0 else None}
            for k, v in prof.items()}


def mfma_roofline(prof, n, kinds=('conv_fwd', 'conv_dgrad', 'conv_wgrad'), kernel=''):
    ms = sum(prof[k]['ms'] for k in kinds if k in prof)
    flop = sum(prof[k]['work'] for k in kinds if k in prof)
    launches = sum(prof[k]['launches'] for k in kinds if k in prof)
    achieved = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    executed = sum(prof[k].get('exec', prof[k]['work']) for k in kinds if k in prof) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None, 'kernel': kernel,
            # multiplies actually ISSUED per second / peak: the Winograd kernels issue 1.5x / 2.25x fewer than the algorithmic count `frac` is
            # quoted on (nnl_prof_collect2); this is what the matrix pipe sees, to be read beside the counter figure `mfma_busy`
            'executed_tflops': round(executed, 2), 'executed_frac': round(executed / FP32_MFMA_PEAK_TFLOPS, 4), 'mfma_busy': None,
            'launches_per_step': launches / max(n, 1), 'avg_launch_ms': ms / max(launches, 1),
            'flop_per_launch': flop / max(launches, 1), 'kernel_ms_per_step': ms / max(n, 1), 'by_kind': by_kind(prof, n)}


def committed_counters(bs, sz, world):
    """HBM-side bytes per launch and MFMA-pipe busy fraction of the dominant kernel family: NOT measured by this run — they come from
    separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` / `SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` passes of this same command
    (tools/pmc_traffic.py; gfx950 corrections applied), committed under profiles/ TOGETHER WITH the source stamp of the library they
    profiled.  Reported only for the configuration those passes profiled AND only while the loaded library still carries that stamp
    (nnl_source_stamp): after any kernel change the figures read null until the passes are re-run — they cannot go stale silently."""
    from neuralnetworklibrary_amd import _lib
    name = next((n for n in ('r5_traffic.json', 'r4_traffic.json') if os.path.exists(os.path.join(ROOT, 'profiles', n))), None)
    try:
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            t = json.load(f)
    except Exception:
        return None, None, 'no committed counter passes'
    if (bs, sz, world) != (t.get('bs', 64), t.get('sz', 224), t.get('gpus', 1)):
        return None, None, 'counter passes exist for another configuration only'
    if t.get('source_stamp') != _lib.source_stamp():
        return None, None, ('profiles/%s was measured on source stamp %s, this library is %s: re-run tools/gpu/r5_counters.sh (or bench.py --counters)'
                            % (name, t.get('source_stamp'), _lib.source_stamp()))
    return round(t['traffic_bytes_per_launch']), t.get('mfma_busy'), ('profiles/%s (rocprofv3 --pmc passes of this command on this very build, source stamp %s; '
                                                                       'committed figures, not measured in this run)' % (name, t['source_stamp']))


def measured_counters(args):
    """--counters (opt-in, N = 1): the three PMC passes the committed figures come from, taken NOW on the loaded library — each pass a FRESH child
    process `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    --no-sweep --configs none` (subprocess: the GPU-holding parent is not replaced; `python3` directly after `--`, no env / shell hop;
    counters in their own runs, never with trace domains).  ~1 min per pass.  Summarised by tools/pmc_traffic.py (gfx950 corrections)."""
    import glob
    import shutil
    import tempfile
    if shutil.which('rocprofv3') is None:
        return {'error': 'rocprofv3 not on PATH'}
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_traffic
    base = tempfile.mkdtemp(prefix='nnl_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    csvs = {}
    try:
        for tag, counters in (('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']), ('mfma', ['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'])):
            d = os.path.join(base, tag)
            cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', tag, '--', 'python3', os.path.abspath(__file__),
                                                       '--steps', '3', '--warmup', '1', '--bs', str(args.bs), '--sz', str(args.sz), '--no-cpu-baseline', '--no-sweep',
                                                       '--no-counters', '--configs', 'none']
            r = subprocess.run(cmd, env=env, cwd='/tmp', capture_output=True, text=True, timeout=900)
            found = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not found:
                return {'error': 'rocprofv3 --pmc %s failed (rc %s): %s' % (' '.join(counters), r.returncode, (r.stderr or '')[-300:])}
            csvs[tag] = found[0]
        js = pmc_traffic.main(csvs['fetch'], csvs['write'], 'live', csvs['mfma'], None, write_files=False)
        return {'traffic': round(js['traffic_bytes_per_launch']), 'mfma_busy': js.get('mfma_busy'),
                'fetch_size_bytes_per_launch_raw': round(js['fetch_size_bytes_per_launch_raw']), 'write_size_bytes_per_launch': round(js['write_size_bytes_per_launch']),
                'source': 'measured by this run: three rocprofv3 --pmc child passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --configs none` '
                          'on the loaded library (FETCH_SIZE x2 per the gfx950 note, WRITE_SIZE exact; per conv / linear GEMM launch)'}
    except Exception as e:                               # noqa: BLE001 — diagnostics must not take the JSON line down
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    finally:
        shutil.rmtree(base, ignore_errors=True)


def host_info():
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'host_cores': os.cpu_count(), 'cpu_model': model}


# ---- CPU baselines: the oracle's restatement of the same step on the host cores (rank 0, N = 1 only) ------------------
def cpu_share():
    """The host share this process really has: the GPU boxes run under a cgroup CPU QUOTA (cpu.max 1600000 / 100000 = 16 CPUs' worth of
    time on a 256-core host, tools/cpu_threads_probe.py) — more OpenMP threads than that are throttled by the scheduler, which is why
    32 / 64 threads measured 9 - 18x SLOWER in round 4.  -> {'threads': min(affinity, quota), 'affinity': ..., 'quota_cpus': ... or None}"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                      # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != 'max':
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {'threads': n, 'affinity': aff, 'quota_cpus': quota}


def _cpu_threads():
    # inside the baseline child the thread count comes from the parent (NNL_BENCH_CPU_THREADS); otherwise the quota / affinity share, at most 64
    n = int(os.environ.get('NNL_BENCH_CPU_THREADS', 0)) or min(cpu_share()['threads'], 64)
    torch.set_num_threads(n)
    return n


def cpu_baseline_child(name, bs, sz):
    """Run one CPU baseline in a FRESH child process: its OpenMP pool is created with OMP_NUM_THREADS = the host share and bound to cores
    (OMP_PROC_BIND=close, OMP_PLACES=cores: +19 % on the ResNet-34 step at 16 threads, tools/cpu_threads_probe.py) — a binding has to be
    in the environment before the runtime starts, hence the child.  The child never touches the GPU.  (subprocess = fork + exec of a
    CHILD; the GPU-holding parent is not replaced.)"""
    share = cpu_share()
    n = min(share['threads'], 64)
    env = dict(os.environ, OMP_NUM_THREADS=str(n), MKL_NUM_THREADS=str(n), OMP_PROC_BIND='close', OMP_PLACES='cores', NNL_BENCH_CPU_THREADS=str(n),
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', name, '--bs', str(bs), '--sz', str(sz)], env=env,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'error': 'cpu baseline child failed (rc %s): %s' % (r.returncode, (r.stderr or r.stdout)[-300:])}
    out = json.loads(lines[-1])
    out['cpu_share'] = share
    out['binding'] = 'OMP_PROC_BIND=close OMP_PLACES=cores, fresh process'
    return out


def cpu_baseline_child_main(name, bs, sz):
    fn = {'resnet': lambda: cpu_baseline_resnet(bs, sz), 'collab': lambda: cpu_baseline_collab(64), 'tabular': lambda: cpu_baseline_tabular(1024),
          'lm': cpu_baseline_lm, 'retinanet': cpu_baseline_retina}[name]
    print(json.dumps(fn()), flush=True)


def _cpu_time(step, steps):
    step()                                                  # warm-up
    t0 = time.time()
    for _ in range(steps):
        step()
    return (time.time() - t0) / steps


def _sgd_adam_step(RM, params, state, kind, lr, wd, **kw):
    RM.optimizer_step(params, [p.grad for p in params], state, [lr] * len(params), [wd] * len(params), kind, **kw)


def cpu_baseline_resnet(bs, sz, steps=5):
    """north_star: "next to the reference run on the node's own host cores (core count stated)".  The oracle's ResNet-34 step at the
    thread count the parent chose (cpu_baseline_child: the cgroup quota / affinity share of this box, threads bound to cores), `steps`
    timed steps after a warm-up.  Round 4 swept 16 / 32 / 64 threads and found the larger counts 9 - 18x slower: the boxes run under a
    16-CPU quota (cpu_share), more threads are throttled — 16 IS this box's host share."""
    from oracle import reference_math as RM, reference_nets as RN
    n = _cpu_threads()
    torch.manual_seed(0)
    net = RN.ImageClassificationNet(RN.resnet34(), 2, 512).train()
    params = list(net.parameters())
    state = RM.OptimState(params)
    x, y = torch.randn(bs, 3, sz, sz), torch.randint(0, 2, (bs,))

    def step():
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        _sgd_adam_step(RM, params, state, 'sgd', 1e-2, 1e-4)
        loss.item()
    dt = _cpu_time(step, steps)
    return dict({'value': bs / dt, 'unit': 'images/s', 'cores': n, 'kind': 'port',
                 'sample': '%d timed steps (after 1 warm-up) of the same bs=%d %dx%d ResNet-34 train step, torch-CPU oracle, %d threads' % (steps, bs, sz, sz, n),
                 'ms_per_step': dt * 1e3}, **host_info())


def cpu_baseline_collab(bs, steps=200):
    from oracle import reference_math as RM, reference_nets as RN
    threads = _cpu_threads()
    torch.manual_seed(0)
    net = RN.CollabFilterNet(943, 1682, 30, [0.8, 5.2])
    params = list(net.parameters())
    state = RM.OptimState(params)
    x = torch.stack([torch.randint(0, 943, (bs,)), torch.randint(0, 1682, (bs,))], 1)
    y = torch.randint(1, 6, (bs,)).float()

    def step():
        for p in params:
            p.grad = None
        loss = RM.mse_loss(net(x), y)
        loss.backward()
        _sgd_adam_step(RM, params, state, 'adam', 1e-2, 1e-4)
        loss.item()
    dt = _cpu_time(step, steps)
    return {'value': bs / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port', 'ms_per_step': dt * 1e3,
            'sample': '%d steps of the same bs=%d ML-100K-shape step, torch-CPU oracle' % (steps, bs)}


def cpu_baseline_tabular(bs, steps=10):
    from oracle import reference_math as RM, reference_nets as RN
    threads = _cpu_threads()
    torch.manual_seed(0)
    rs = np.random.RandomState(0)
    net = RN.StructuredDataNet('cont', [(c, RN.embedding_dim(c)) for c in ROSSMANN_CARDS], 14, [1000, 500, 1], [5, 12],
                               (0.04, 0.04, [0, 0.5, 0.25])).train()
    params = list(net.parameters())
    state = RM.OptimState(params)
    xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=bs) for c in ROSSMANN_CARDS], 1).astype(np.int64))
    xcont, y = torch.randn(bs, 14), 5 + 7 * torch.rand(bs)

    def step():
        for p in params:
            p.grad = None
        loss = RM.mse_loss(net(xcat, xcont), y)
        loss.backward()
        _sgd_adam_step(RM, params, state, 'adam', 1e-3, 1e-3)
        loss.item()
    dt = _cpu_time(step, steps)
    return {'value': bs / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port', 'ms_per_step': dt * 1e3,
            'sample': '%d steps of the same bs=%d Rossmann-shape step, torch-CPU oracle' % (steps, bs)}


def cpu_baseline_lm(streams=16, steps=1):
    from oracle import reference_math as RM, reference_text as RT
    threads = _cpu_threads()
    torch.manual_seed(0)
    net = RT.LanguageModelNet(LM_V, 1, streams)
    params = list(net.parameters())
    state = RM.OptimState(params)
    x = torch.randint(4, LM_V, (streams, LM_BPTT))
    y = torch.randint(4, LM_V, (streams, LM_BPTT))
    keep = lambda shape, p: torch.bernoulli(torch.full(shape, 1 - p)) / (1 - p)
    sizes = [400, 1150, 1150, 400]

    def step():
        for p in params:
            p.grad = None
        masks = {'emb_rows': keep((LM_V, 1), 0.035), 'emb_locked': keep((1, streams, 400), 0.175),
                 'weights': [keep((4 * sizes[i + 1], sizes[i + 1]), 0.14) for i in range(3)],
                 'hidden': [keep((1, streams, sizes[i + 1]), 0.105) for i in range(3)]}
        loss = RT.reg_seq_cross_entropy(net(x, masks, keep((1, streams, 400), 0.07)), y, 2.0, 1.0)[0]
        loss.backward()
        _sgd_adam_step(RM, params, state, 'adam', 1e-3, 1e-6, betas=(0.8, 0.99), clip=0.4)
        loss.item()
    dt = _cpu_time(step, steps)
    return {'value': streams * LM_BPTT / dt, 'unit': 'tokens/s', 'cores': threads, 'kind': 'port', 'ms_per_step': dt * 1e3,
            'sample': '%d step(s) of the full-size model (400/1150/3, V=%d, bptt %d) on %d of the 64 streams (after 1 warm-up), '
                      'torch-CPU oracle' % (steps, LM_V, LM_BPTT, streams)}


def cpu_baseline_retina(bs=1, sz=512, steps=1):
    from oracle import reference_math as RM, reference_nets as RN
    threads = _cpu_threads()
    torch.manual_seed(0)
    net = RN.ObjectDetectionNet(20).train()
    with torch.no_grad():                                    # the reference's head initialisation (Vision.py:1425-1428)
        net.classifier.output.weight.zero_(); net.classifier.output.bias.fill_(-float(np.log(99.0)))
        net.regressor.output.weight.zero_(); net.regressor.output.bias.zero_()
    params = list(net.parameters())
    state = RM.OptimState(params)
    x = torch.randn(bs, 3, sz, sz)
    boxes, cats = _retina_targets(np.random.RandomState(0), bs)
    B, Cc = torch.from_numpy(boxes), torch.from_numpy(cats)

    def step():
        for p in params:
            p.grad = None
        anchors, reg, clas = net(x)
        loss = RM.ssd_loss(anchors, reg, clas, B, Cc, 0.5, 0.25, 2.0)[0]
        loss.backward()
        _sgd_adam_step(RM, params, state, 'sgd', 1e-3, 1e-4)
        loss.item()
    dt = _cpu_time(step, steps)
    return {'value': bs / dt, 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'ms_per_step': dt * 1e3,
            'sample': '%d step(s) of the same RetinaNet R50-FPN %dx%d train step on %d of the 16 images (after 1 warm-up), '
                      'torch-CPU oracle' % (steps, sz, sz, bs)}


# ---- the other four configs -----------------------------------------------------------------------------------------
def _n_params(model):
    seen, n = set(), 0
    for p in model.parameters():
        if id(p) not in seen and p.requires_grad:
            seen.add(id(p)); n += p.numel()
    return n


def run_config(name, device, world, rank, clock, steps, warmup, cpu):
    """One of the non-headline configs: weak scaling (the named batch PER GPU), K timed steps, a HIP-event profile of 3 more,
    the roofline SURVEY.md §8(d) prescribes for it and (N = 1) a bounded CPU-oracle baseline."""
    seed = 1234 + {'collab': 0, 'tabular': 2, 'lm': 3, 'retinanet': 4}[name] + 1000 * rank
    if name == 'collab':
        wl, bs = collab_workload(device, 64, seed, world), 64
    elif name == 'tabular':
        wl, bs = tabular_workload(device, 1024, seed, world), 1024
    elif name == 'lm':
        wl, bs = lm_workload(device, 64, seed, world), 64
    else:
        wl, bs = retina_workload(device, 16, seed, world), 16
    out = {'workload': {'collab': 'MovieLens-100K CollabFiltering EmbeddingDotBias D=30, bs=64 per GPU, Adam',
                        'tabular': 'Rossmann-shape StructuredData MLP (32 embeddings + 14 continuous -> 1000 -> 500 -> 1), bs=1024 per GPU, Adam',
                        'lm': 'IMDB AWD-LSTM language model 400/1150/3, V=47343, bptt=70, bs=64 per GPU, Adam, RegSeqCrossEntropyLoss(2,1)',
                        'retinanet': 'Pascal RetinaNet (ResNet-50 FPN + FocalLoss / smooth-L1), 512x512, bs=16 per GPU, SGD momentum'}[name],
           'unit': wl.unit, 'steps': steps, 'dtype': 'f32', 'scaling': 'weak'}
    if name in ('collab', 'tabular'):
        wl.learner.use_graphs(False)                        # these heads default to whole-step replay: measure the eager path first
    dt, med = clock.timed_median(wl.step, warmup, steps)
    ms = dt / steps * 1e3
    out.update(ms_per_step=round(ms, 3), median_ms_per_step=None if med is None else round(med, 3),
               value=round(wl.units_per_step * world * steps / dt, 1), last_loss=wl.loss,
               protocol='%d timed steps after %d warm-ups: value / ms_per_step = whole-job wall clock (barrier + synchronise on both sides, '
                        'MAX over ranks); median_ms_per_step = median of the per-step device times (SURVEY 8d)' % (steps, warmup))
    if name in ('collab', 'tabular') and world == 1:        # launch-bound heads: the captured whole-step hipGraph (Learner.use_graphs)
        wl.learner.use_graphs(True)
        dtg, medg = clock.timed_median(wl.step, warmup + 3, steps)
        wl.learner.use_graphs(False)
        msg = dtg / steps * 1e3
        out['eager_step'] = {'ms_per_step': round(ms, 3), 'median_ms_per_step': out['median_ms_per_step'], 'value': out['value']}
        out['hipgraph_step'] = {'ms_per_step': round(msg, 3), 'median_ms_per_step': None if medg is None else round(medg, 3),
                                'value': round(wl.units_per_step * steps / dtg, 1)}
        # unambiguous keys for the two modes (`ms_per_step` / `median_ms_per_step` below describe the FASTER mode only)
        out['eager_mean_ms'], out['eager_median_ms'] = round(ms, 3), out['median_ms_per_step']
        out['replay_mean_ms'], out['replay_median_ms'] = round(msg, 3), None if medg is None else round(medg, 3)
        out['eager_over_replay'] = round(ms / msg, 2)
        # ... and as Learner.fit() drives it since round 4: the loss of a replayed step is read one step late (same values, same order),
        # so the GPU does not idle while the host stages the next minibatch
        wl.learner.use_graphs(True)
        for i in range(warmup + 3):
            wl.step_as_in_fit(i)
        wl.flush()

        def fit_steps(i):
            wl.step_as_in_fit(i)
            if i == steps - 1:
                wl.flush()
        dtf = clock.timed(fit_steps, 0, steps)
        wl.learner.use_graphs(False)
        msf = dtf / steps * 1e3
        out['replay_in_fit_loop_mean_ms'] = round(msf, 3)
        out['fit_loop_step'] = {'ms_per_step': round(msf, 3), 'value': round(wl.units_per_step * steps / dtf, 1),
                                'note': "the replayed step inside Learner.fit()'s loop: every loss is read (inside the timed region), one step "
                                        'after its launch; train1minibatch called on its own still returns its own float (replay_mean_ms)'}
        if msf < msg and msf < ms:                          # headline of this config = the product's fastest mode, named
            out.update(ms_per_step=round(msf, 3), value=out['fit_loop_step']['value'], median_ms_per_step=None,
                       mode="whole-step hipGraph replay as Learner.fit() drives it (the default of these launch-bound heads): forward + loss + "
                            'backward + fused optimizer in one graph, the loss read back one step late; hipgraph_step = the same replay with '
                            'the per-step loss.item() of a bare train1minibatch call; eager_step = learner.use_graphs(False)')
        elif msg < ms:                                      # headline of this config = the product's faster mode, named
            out.update(ms_per_step=round(msg, 3), value=out['hipgraph_step']['value'], median_ms_per_step=out['replay_median_ms'],
                       mode='whole-step hipGraph replay (forward + loss + backward + fused optimizer in one graph): the DEFAULT of these '
                            'launch-bound heads since round 3 (Learner enables it for models marked nnl_default_graphs); '
                            'eager_step = learner.use_graphs(False)')
        ms_best = min(ms, msg, msf)
    else:
        ms_best = ms
    n_prof = 3
    prof = profile_kinds(wl, n_prof)
    n_par = _n_params(wl.learner.model)
    if name == 'lm':
        rf = mfma_roofline(prof, n_prof, FLOP_KINDS, 'igemm_taps / igemm_wgrad (input, decoder and weight-gradient GEMMs) + the LSTM recurrence kernels')
        rf['algorithmic_tflop_per_step'] = LM_MFLOP_PER_TOKEN * 1e6 * wl.units_per_step / 1e12
        rf['whole_step_tflops'] = round(rf['algorithmic_tflop_per_step'] / (ms * 1e-3), 2)
    elif name == 'retinanet':
        rf = mfma_roofline(prof, n_prof, FLOP_KINDS, 'igemm_taps / wino / igemm_wgrad (ResNet-50 + FPN + head convolutions: fwd, dgrad, wgrad)')
        rf['algorithmic_tflop_per_step'] = RETINA_GFLOP_PER_IMAGE * 1e9 * bs / 1e12
        rf['whole_step_tflops'] = round(rf['algorithmic_tflop_per_step'] / (ms * 1e-3), 2)
        if 'retina_loss' in prof:
            lb = RETINA_LOSS_BYTES_PER_ANCHOR * 49104.0 * bs
            lms = prof['retina_loss']['ms'] / n_prof
            rf['loss_kernels'] = {'bound': 'hbm', 'achieved': round(lb / (lms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': round(lb / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'ms_per_step': round(lms, 4),
                                  'kernel': 'retina_fwd + retina_finalize + retina_bwd (224 B/anchor algorithmic)'}
    else:
        # launch / HBM-latency bound — judged by algorithmic bytes per step over the WHOLE step time
        if name == 'collab':
            alg = 516.0 * bs + n_par * 28.0
            kern = 'embdotbias_fwd/bwd + fused Adam over the dense tables (516 B/sample + 28 B/parameter)'
        else:
            alg = n_par * 28.0 + bs * (203 + 3 * 1000 + 3 * 500) * 4.0 * 2 + bs * 189 * 4.0 * 2
            kern = 'tab_gather/scatter + igemm linears + BN1d + fused Adam (weights+Adam state 28 B/param, activations, gathers)'
        gbs = alg / (ms_best * 1e-3) / 1e9
        rf = {'bound': 'hbm', 'achieved': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 5),
              'traffic': None, 'kernel': kern, 'algorithmic_bytes_per_step': alg,
              'note': 'latency-bound at this batch size: achieved = algorithmic bytes / whole step time (best of eager and hipGraph)',
              'kernel_ms_per_step': round(sum(v['ms'] for v in prof.values()) / n_prof, 4), 'by_kind': by_kind(prof, n_prof)}
    out['roofline'] = rf
    del wl
    torch.cuda.empty_cache()
    if cpu:
        out['cpu_baseline'] = cpu_baseline_child(name, 0, 0)
    return out


# =====================================================================================================================
# data-parallel diagnostics (N > 1)
# =====================================================================================================================
def dp_diagnostics(wl, clock, dist, steps, ms_dp):
    """Stand-alone all-reduce time of one step's gradient buckets, the step time with the collectives removed, and from the
    two the communication that stays exposed: overlap = 1 - exposed / standalone."""
    gs = wl.learner.grad_sync
    reps = 5

    def allreduce_only(_):
        hs = [dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, async_op=True) for b in gs.buckets]
        for h in hs:
            h.wait()
    ar_ms = clock.timed(allreduce_only, 2, reps) / reps * 1e3
    # each bucket alone (so that the first real N = 8 run can be read bucket by bucket: size -> ms -> bus GB/s)
    per_bucket = []
    world = dist.get_world_size()
    for b in gs.buckets:
        one = lambda _i, b=b: dist.all_reduce(b.flat, op=dist.ReduceOp.AVG)
        ms_b = clock.timed(one, 1, 3) / 3 * 1e3
        mb = b.numel * 4 / 2 ** 20
        per_bucket.append({'mbytes': round(mb, 2), 'ms': round(ms_b, 3),
                           'bus_gbs': round(2.0 * (world - 1) / max(world, 1) * b.numel * 4 / (ms_b * 1e-3) / 1e9, 1) if ms_b > 0 else None})
    learner = wl.learner
    learner.grad_sync = None
    learner.optimizer.attach_grad_sync(None)
    ms_nosync = clock.timed(wl.step, 2, steps) / steps * 1e3
    learner.grad_sync = gs
    learner.optimizer.attach_grad_sync(gs)
    exposed = max(ms_dp - ms_nosync, 0.0)
    return {'grad_buckets': len(gs.buckets), 'grad_mbytes': round(sum(b.numel for b in gs.buckets) * 4 / 2 ** 20, 1),
            'grads_written_in_place': '%d of %d tensors per step' % (gs.direct_writes // max(gs.steps, 1), sum(len(b.params) for b in gs.buckets)),
            'allreduce_ms_per_step_standalone': round(ar_ms, 3), 'allreduce_per_bucket': per_bucket,
            'ms_per_step_without_allreduce': round(ms_nosync, 3),
            'exposed_comm_ms_per_step': round(exposed, 3),
            'overlap_frac': round(1.0 - exposed / ar_ms, 3) if ar_ms > 0 else None}


# =====================================================================================================================
# dry run (CPU, gloo): the launch / rendezvous / timing / JSON protocol without the HIP path
# =====================================================================================================================
def dry_run_worker(args, world, rank):
    import torch.distributed as dist
    import torch.nn as nn
    from neuralnetworklibrary_amd.General.Core import make_model_basic, set_default_device
    set_default_device('cpu')
    torch.set_num_threads(1)
    device = torch.device('cpu')
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)
    clock = Clock(dist if world > 1 else None, device)
    out = {}
    for tag, bs in (('weak', args.bs), ('strong', max(args.bs // world, 1))):
        torch.manual_seed(0)
        net = make_model_basic(nn.Sequential(nn.Linear(16, 32), nn.ReLU(), nn.Linear(32, 1), nn.Flatten(0)))
        g = torch.Generator().manual_seed(rank)
        batches = [(torch.randn(bs, 16, generator=g), torch.randn(bs, generator=g)) for _ in range(4)]
        learner = _learner_cls()('/tmp/nnl_bench_dry', Data(batches, bs, 'cont'), net, optimizer='SGD_Mom')
        learner.init_optimizer(wd=1e-4)
        if world > 1:
            learner.distribute(equal_shards=True)
        wl = Workload('dry', learner, batches, 1e-2, 'samples/s', bs)
        dt = clock.timed(wl.step, args.warmup, args.steps)
        out[tag] = {'global_batch': bs * world, 'per_gpu_batch': bs, 'ms_per_step': round(dt / args.steps * 1e3, 3),
                    'value': round(bs * world * args.steps / dt, 2)}
    if rank == 0:
        print(json.dumps({'metric': 'DRY RUN (CPU, gloo): launch protocol only, not a measurement', 'dry_run': True,
                          'value': out['weak']['value'], 'unit': 'samples/s', 'n_gpus': world, 'ranks_seen': int(ones.item()),
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': out['weak']['ms_per_step'],
                          'higher_is_better': True, 'scaling': 'weak', 'strong': out['strong']}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# =====================================================================================================================
def worker(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    if args.dry_run:
        return dry_run_worker(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    from neuralnetworklibrary_amd.General.Core import set_default_device
    set_default_device(device)
    dist = None
    force_dist = os.environ.get('NNL_BENCH_FORCE_DIST') == '1'       # exercise the RCCL path on a 1-GPU box (world_size 1)
    ranks_seen = 1
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL prints a version banner on STDOUT when the communicator is created: keep rank 0's stdout to the ONE JSON line by
        # pointing fd 1 at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
            ones = torch.ones(1, device=device)
            dist.all_reduce(ones)
            ranks_seen = int(ones.item())
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    clock = Clock(dist, device)
    seed = 1234 + 1 + 1000 * rank

    # ---- headline: weak scaling, args.bs images per GPU ----
    wl = resnet34_workload(device, args.bs, seed, world, args.sz)
    dt = clock.timed(wl.step, args.warmup, args.steps)
    ms = dt / args.steps * 1e3
    last_loss = wl.loss
    n_prof = min(args.steps, 10)
    prof = profile_kinds(wl, n_prof)
    roofline = mfma_roofline(prof, n_prof, kernel='igemm_taps_kernel / wino_kernel / wino2_kernel / igemm_wgrad_kernel (fp32 MFMA implicit-GEMM conv2d + linear: '
                                                  'fwd, dgrad, wgrad; incl. their slab reduces and filter transforms)')
    roofline['note'] = ('achieved = ALGORITHMIC convolution flop (2 N P Q K R S C per pass, SURVEY.md 8d) / measured time: the 3x3 stride-1 '
                        'forward / dgrad launches run a fused Winograd F(2,3) kernel that issues 1.5x fewer MFMA multiplies than that count')
    roofline['conv_ms_per_step'] = roofline['kernel_ms_per_step']
    roofline['traffic'], roofline['mfma_busy'], roofline['counters_source'] = committed_counters(args.bs, args.sz, world)
    roofline['note'] += ('; executed_frac counts the multiplies actually issued (1.5x / 2.25x fewer on the Winograd launches); mfma_busy = '
                         'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) over the same kernels, from the committed counter passes')
    from neuralnetworklibrary_amd import _lib as _nl
    roofline['library_source_stamp'] = _nl.source_stamp()
    out = {
        'metric': 'ResNet-34 224x224 training throughput (Learner.train1minibatch, fwd+loss+bwd+optimizer)',
        'value': round(args.bs * world * args.steps / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'ranks_seen': ranks_seen,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'DogsCats ResNet-34 classifier, 224x224, bs=%d per GPU, SGD momentum 0.9, fp32' % args.bs,
                   'global_batch': args.bs * world, 'image_size': args.sz, 'parallelism': 'dp%d' % world, 'last_loss': last_loss},
        'roofline': roofline,
    }
    if wl.learner.grad_sync is not None and dist is not None:
        out['dp'] = dp_diagnostics(wl, clock, dist, args.steps, ms)
    del wl
    torch.cuda.empty_cache()

    # ---- strong scaling: GLOBAL batch args.bs (north-star: >= 6.5x at 8 GPUs) ----
    if world > 1 or force_dist:                      # (force_dist: rehearse the N > 1 legs on a 1-GPU box, world size 1)
        per = max(args.bs // max(world, 1) // (8 if force_dist and world == 1 else 1), 1)
        strong = {'global_batch': per * world, 'per_gpu_batch': per}
        # the leg that has never met a real multi-GPU communicator (per-bucket wait kernels + RCCL on a side stream under the replay,
        # round 4) runs LAST, and no leg can take the JSON line down with it: a failure is reported in its place
        for tag, sync_bn, graphs in (('local_bn', False, False), ('local_bn_hipgraph_no_overlap', False, True), ('sync_bn', True, False),
                                     ('local_bn_hipgraph', False, True)):
            os.environ['NNL_DIST_REPLAY_OVERLAP'] = '0' if tag.endswith('no_overlap') else '1'
            try:
                w2 = resnet34_workload(device, per, seed, world, args.sz, sync_bn=sync_bn)
                if graphs:
                    w2.learner.use_graphs(True)
                d2 = clock.timed(w2.step, args.warmup + (3 if graphs else 0), args.steps)
                strong[tag] = {'ms_per_step': round(d2 / args.steps * 1e3, 3), 'value': round(per * world * args.steps / d2, 2)}
                gs2 = w2.learner.grad_sync
                if graphs and gs2 is not None:
                    if gs2.overlap is not None:
                        gs2.raise_if_overlap_error()
                        strong[tag]['buckets_signalled_in_the_captured_backward'] = '%d of %d' % (gs2.last_signalled or 0, len(gs2.buckets))
                        strong[tag]['collectives_behind_wait_kernels'] = gs2.overlap_launches
                    if tag == 'local_bn_hipgraph':           # stand-alone all-reduce time of this step's buckets: the yardstick of the overlap
                        def ar_only(_):
                            hs = [dist.all_reduce(bk.flat, op=dist.ReduceOp.AVG, async_op=True) for bk in gs2.buckets]
                            for hh in hs:
                                hh.wait()
                        strong['allreduce_ms_standalone'] = round(clock.timed(ar_only, 2, 5) / 5 * 1e3, 3)
                del w2
                ok = 1.0
            except Exception as e:                       # noqa: BLE001 — reported, not raised: the headline above is already measured
                strong[tag] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
                ok = 0.0
            torch.cuda.empty_cache()
            if world > 1:
                # the ranks agree on the leg's outcome before the next one starts: a failure that every rank meets at the same
                # point (an overlap time-out, a deterministic Python error) leaves all of them here together; rank 0's report then says so
                # even when its own copy of the leg went through
                okt = torch.tensor([ok], device=device)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                if float(okt.item()) < 1.0 and 'error' not in strong[tag]:
                    strong[tag] = {'error': 'the leg failed on another rank', 'this_rank': strong[tag]}
        os.environ.pop('NNL_DIST_REPLAY_OVERLAP', None)
        have = all('ms_per_step' in strong.get(k, {}) for k in ('local_bn_hipgraph_no_overlap', 'local_bn_hipgraph'))
        saved = strong['local_bn_hipgraph_no_overlap']['ms_per_step'] - strong['local_bn_hipgraph']['ms_per_step'] if have else 0.0
        strong['replay_overlap'] = {'ms_saved_per_step': round(saved, 3) if have else None,
                                    'overlap_fraction': round(max(saved, 0.0) / strong['allreduce_ms_standalone'], 3) if (have and strong.get('allreduce_ms_standalone')) else None,
                                    'note': 'local_bn_hipgraph: the captured backward signals each bucket, its all-reduce starts behind a wait kernel on a side '
                                            'stream in the middle of the replay (dist.GradSync.reduce_overlapped); _no_overlap: all buckets after the replay '
                                            '(NNL_DIST_REPLAY_OVERLAP=0); overlap_fraction = time saved / stand-alone all-reduce time of the same buckets'}
        strong['note'] = ('local_bn: per-replica BatchNorm statistics (standard DDP); local_bn_hipgraph: the same with forward + backward '
                          'replayed as one hipGraph per step (Learner.use_graphs; the all-reduces and the optimizer launch follow it); sync_bn: '
                          "global-batch statistics = the single-GPU reference's numerics on the same global minibatch (SURVEY.md §8e)")
        out['strong'] = strong
    elif not args.no_sweep:
        # one GPU: the compute-side ceiling of strong scaling — the per-GPU step at 64/N images
        proxy = {'t64_ms': round(ms, 3)}
        for bs in (32, 16, 8):
            w2 = resnet34_workload(device, bs, seed, 1, args.sz)
            d2 = clock.timed(w2.step, max(args.warmup, 3), args.steps)
            m2 = d2 / args.steps * 1e3
            w2.learner.use_graphs(True)                 # the whole step replayed as one hipGraph: no per-launch host cost
            d3 = clock.timed(w2.step, max(args.warmup, 3) + 3, args.steps)
            m3 = d3 / args.steps * 1e3
            proxy['bs%d' % bs] = {'ms_per_step': round(m2, 3), 'images_per_s': round(bs * args.steps / d2, 1),
                                  't64_over_t': round(ms / m2, 2), 'hipgraph_ms_per_step': round(m3, 3),
                                  'hipgraph_t64_over_t': round(ms / m3, 2), 'ideal': 64 // bs}
            del w2
            torch.cuda.empty_cache()
        proxy['note'] = 't64_over_t at bs=64/N bounds the strong-scaling speed-up at N GPUs before any communication (target >= 6.5 at N=8)'
        out['strong_scaling_proxy'] = proxy

    # ---- the other four BASELINE configs ----
    cpu = (not args.no_cpu_baseline) and world == 1 and rank == 0
    if args.configs != 'none':
        names = ['collab', 'tabular', 'lm', 'retinanet'] if args.configs == 'all' else args.configs.split(',')
        out['configs'] = {}
        for name in names:
            # >= 50 timed steps after 10 warm-ups for every side config (RetinaNet: 50 x ~57 ms = 3 s), unless the
            # caller asks for a quick run (--steps < 10)
            quick = args.steps < 10
            k = {'collab': 200, 'tabular': 100, 'lm': 50, 'retinanet': 50}[name] if not quick else max(args.steps, 3)
            try:
                out['configs'][name] = run_config(name, device, world, rank, clock, k, 10 if not quick else 3, cpu)
            except Exception as e:                       # noqa: BLE001 — a side config must not take the headline's JSON line down
                out['configs'][name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    if cpu:
        out['cpu_baseline'] = cpu_baseline_child('resnet', args.bs, args.sz)
    under_profiler = any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprofiler' in os.environ.get('LD_PRELOAD', '')
    if args.counters and world == 1 and rank == 0 and not under_profiler:       # (never from inside a profiler run: its children would nest)
        # the committed, stamp-gated figures stay the fallback; a measured pass of THIS build is reported beside them and fills the
        # two roofline fields when the committed ones are null (any source change since the passes were committed)
        mc = measured_counters(args)
        out['roofline']['measured_counters'] = mc
        if 'error' not in mc and out['roofline'].get('traffic') is None:
            out['roofline']['traffic'], out['roofline']['mfma_busy'] = mc['traffic'], mc['mfma_busy']
            out['roofline']['counters_source'] = mc['source']
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=64, help='images per GPU of the headline (weak scaling); also the GLOBAL batch of the strong-scaling leg')
    ap.add_argument('--sz', type=int, default=224)
    ap.add_argument('--configs', default='all', help="'all', 'none' or a comma list of collab,tabular,lm,retinanet")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true', help='skip the 1-GPU bs 8/16/32 strong-scaling proxy')
    ap.add_argument('--dry-run', action='store_true', help='CPU + gloo rehearsal of the launch / timing / JSON protocol (no HIP path)')
    ap.add_argument('--no-counters', dest='counters', action='store_false', help='N = 1: skip the three rocprofv3 --pmc child passes (HBM-side bytes, MFMA busy of this build, ~40 s)')
    ap.add_argument('--cpu-baseline-child', default=None, help='(internal) run ONE CPU baseline in this fresh process and print its JSON')
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child_main(args.cpu_baseline_child, args.bs, args.sz)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))                # BEFORE any GPU call in this process
    worker(args)


if __name__ == '__main__':
    main()
