"""Two REAL ranks through the replay-overlap hand-over (dist.GradSync.reduce_overlapped): two processes on cuda:0 over gloo
(the pool has one GPU per box), Learner.use_graphs(True) under distribute(): the captured backward signals each bucket, every rank's side
stream runs wait kernel -> all-reduce per bucket.  One rank is delayed every step, so the flags of the two ranks fire at different times;
the run must equal eager data parallelism and the no-overlap replay bitwise.  Then the time-out path: a rank whose replay starts far
too late lets its wait kernels expire — it must not train on silently: its error word reaches every rank in the next step's first
bucket and BOTH ranks raise at the same step, after which training goes on with collectives behind the replay."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'
STEPS = 10


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _net():
    from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import HipConv2d
    from neuralnetworklibrary_amd.General.Core import make_model_basic
    from neuralnetworklibrary_amd import ops

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.b1 = HipConv2d(4, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16)
            self.c2, self.b2 = HipConv2d(16, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16)
            self.c3, self.b3 = HipConv2d(16, 32, 3, padding=1, bias=False), nn.BatchNorm2d(32)
            self.fc = nn.Linear(32, 1)

        def forward(self, x):
            h = ops.bn_act(self.b1, self.c1(x), relu=True)
            h = ops.bn_act(self.b2, self.c2(h), residual=h, relu=True)
            h = ops.bn_act(self.b3, self.c3(h), relu=True)
            return self.fc(h.mean(dim=(2, 3))).flatten()

    torch.manual_seed(0)
    return make_model_basic(Net())


def _batches():
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(8, 4, 6, 6, generator=g), torch.randn(8, generator=g)) for _ in range(STEPS)]


class _Data:
    target_type = 'cont'

    def __init__(self, batches, bs):
        self.train_dl, self.val_dl, self.bs = batches, batches, bs


def _gpu_sleep(seconds):
    "torch.cuda._sleep counts device clock ticks of an unspecified rate: calibrate once, then sleep on the current stream"
    if not hasattr(_gpu_sleep, 'ticks_per_s'):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); torch.cuda._sleep(2000000); b.record()
        torch.cuda.synchronize()
        _gpu_sleep.ticks_per_s = 2000000 / max(a.elapsed_time(b) * 1e-3, 1e-6)
    torch.cuda._sleep(int(seconds * _gpu_sleep.ticks_per_s))


def _fit(rank, world, port, q, mode):
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    os.environ['NNL_DIST_REPLAY_OVERLAP'] = '0' if mode == 'graph_no_overlap' else '1'
    if mode == 'timeout':
        os.environ['NNL_DIST_WAIT_SECONDS'] = '0.05'
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    batches = [(x.to(DEV), y.to(DEV)) for x, y in _batches()]
    data = _Data(nd.ShardedBatches(batches, rank, world), 8 // world)
    learner = Learner('/tmp/nnl_dp2_%s_%d' % (mode, rank), data, _net(), optimizer='SGD_Mom')
    learner.distribute(bucket_mb=0.008)                              # 2048 floats per bucket: four buckets for this net
    assert len(learner.grad_sync.buckets) >= 3
    if mode != 'eager':
        learner.use_graphs(True, warmup=2)
    learner.model.train()
    losses, raised = [], []
    for i, (x, y) in enumerate(data.train_dl):
        if mode == 'graph_delayed' and i % 2 == rank:
            time.sleep(0.03)                                          # this rank's replay (and its flags) come 30 ms after the other's
        if mode == 'timeout' and i == 5 and rank == 1:
            _gpu_sleep(0.4)                                           # 0.4 s on the main stream ahead of the replay: the 50 ms waits expire
        try:
            losses.append(learner.train1minibatch(x, y, 5e-2, mom_batch=0.9))
        except RuntimeError as e:
            assert 'timed out' in str(e), e
            raised.append(i)
            losses.append(float('nan'))
    gs = learner.grad_sync
    info = dict(replays=gs.overlap.replays if gs.overlap is not None else 0, launches=gs.overlap_launches, ok=gs.overlap_ok,
                graphs=sum(g.graph is not None for g in learner._graphs.values()), raised=raised)
    if mode != 'timeout':
        gs.raise_if_overlap_error()                                   # the epoch-end form: one MAX all-reduce, nothing to raise
    sd = {k: v.detach().cpu().numpy() for k, v in learner.model.state_dict().items()}
    q.put((rank, losses, sd, info))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode):
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_fit, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_ranks_replay_overlap_equals_eager_data_parallelism():
    eager, delayed, plain = _run('eager'), _run('graph_delayed'), _run('graph_no_overlap')
    for r in range(2):
        info = delayed[r][3]
        assert info['graphs'] == 1 and info['ok'] and info['raised'] == []
        assert info['replays'] == STEPS - 2 and info['launches'] >= 3 * (STEPS - 2)       # every replay's collectives went out behind wait kernels
        assert plain[r][3]['launches'] == 0
    for k in eager[0][2]:
        if 'running_' not in k and 'num_batches' not in k:            # (BatchNorm buffers are per-replica statistics: local BN)
            assert np.array_equal(delayed[0][2][k], delayed[1][2][k]), 'replicas agree bitwise: ' + k
        for r in range(2):
            assert np.array_equal(delayed[r][2][k], plain[r][2][k]), 'overlapped and non-overlapped replays are bitwise identical: ' + k
            assert_close(delayed[r][2][k], eager[r][2][k], 1e-4, 1e-6, 'replayed vs eager data parallelism: ' + k)
    assert_close(np.array(delayed[0][1]), np.array(eager[0][1]), 1e-5, 1e-6, 'loss curve, rank 0')
    assert_close(np.array(delayed[1][1]), np.array(eager[1][1]), 1e-5, 1e-6, 'loss curve, rank 1')


def test_two_ranks_wait_time_out_is_raised_on_both_ranks_at_the_same_step():
    res = _run('timeout')
    r0, r1 = res[0][3], res[1][3]
    # rank 1's waits expired at step 5 (its replay started ~0.4 s late); the word travelled in step 6's first bucket: both raise there
    assert r0['raised'] == [6] and r1['raised'] == [6], (r0, r1)
    assert r1['ok'] is False                                          # rank 1 reduces behind the replay from then on; the run went on to the end
    assert all(np.isfinite(v) for i, v in enumerate(res[0][1]) if i != 6)
