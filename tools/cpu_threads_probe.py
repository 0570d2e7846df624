"""Why did the oracle's ResNet-34 step get 9 - 18x SLOWER with 32 / 64 threads on the GPU box?  Prints the process's CPU
environment (affinity, cgroup quota and throttling counters, load) and times the step in a FRESH subprocess per (threads, binding) leg.
Usage: python tools/cpu_threads_probe.py [--legs 16,32,64] [--bs 16]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def env_report():
    rep = {'os_cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0)), 'loadavg': read('/proc/loadavg')}
    for name in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpuset.cpus.effective', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us',
                 '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
        v = read(name)
        if v is not None:
            rep[name] = v.replace('\n', ' | ')
    return rep


def leg(threads, bs):
    import torch
    sys.path.insert(0, ROOT)
    from oracle import reference_nets as RN
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = RN.ImageClassificationNet(RN.resnet34(), 2, 512).train()
    x, y = torch.randn(bs, 3, 224, 224), torch.randint(0, 2, (bs,))

    def step():
        for p in net.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(net(x), y).backward()
    t0 = time.time(); step(); warm = time.time() - t0
    t0 = time.time(); step(); step(); dt = (time.time() - t0) / 2
    print(json.dumps({'threads': threads, 'torch_threads': torch.get_num_threads(), 'interop': torch.get_num_interop_threads(), 'warm_s': round(warm, 2),
                      'step_s': round(dt, 3), 'img_per_s': round(bs / dt, 1), 'throttle': read('/sys/fs/cgroup/cpu.stat')}))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--legs', default='16,32,64')
    ap.add_argument('--bs', type=int, default=16)
    ap.add_argument('--leg', type=int, default=0)
    a = ap.parse_args()
    if a.leg:
        leg(a.leg, a.bs)
        sys.exit(0)
    print(json.dumps(env_report()))
    for n in [int(v) for v in a.legs.split(',')]:
        for bind in (None, 'close'):
            env = dict(os.environ, OMP_NUM_THREADS=str(n), MKL_NUM_THREADS=str(n))
            if bind:
                env.update(OMP_PROC_BIND=bind, OMP_PLACES='cores')
            t0 = time.time()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--leg', str(n), '--bs', str(a.bs)], env=env, capture_output=True, text=True, timeout=600)
            print('threads=%d bind=%s wall=%.1fs -> %s' % (n, bind, time.time() - t0, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
