"""Data-parallel logic on CPU with the gloo backend, world_size 2 (the RCCL path is the same code with backend 'nccl'):
bucketed gradient averaging == single-process gradients on the concatenated batch; a 2-rank Learner.fit reproduces the
1-rank run on the global batches (loss curve and final weights); ragged last batch lr scaling uses the global size."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _model(seed=0, dropout=0.0):
    from neuralnetworklibrary_amd.General.Core import make_model_basic
    from neuralnetworklibrary_amd.dist import KeyedDropout
    torch.manual_seed(seed)
    drop = [KeyedDropout(dropout)] if dropout else []
    net = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), *drop, nn.Linear(16, 16), nn.Tanh(), *([KeyedDropout(dropout)] if dropout else []),
                        nn.Linear(16, 1), nn.Flatten(0))
    return make_model_basic(net)


def _batches(n_batches=5, bs=8, last=6):
    g = torch.Generator().manual_seed(3)
    out = []
    for i in range(n_batches):
        b = last if i == n_batches - 1 else bs
        out.append((torch.randn(b, 6, generator=g), torch.randn(b, generator=g)))
    return out


class _Data:
    target_type = 'cont'

    def __init__(self, batches, bs):
        self.train_dl, self.val_dl, self.bs = batches, batches, bs


def _fit(rank, world, port, q, last=6, n_batches=5, hint=True, dropout=0.0):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device('cpu')
    torch.set_num_threads(1)
    Learner.verbose = False
    if world > 1:
        nd.init_from_env('gloo')
    batches = _batches(n_batches, 8, last)
    shard = nd.ShardedBatches(batches, rank, world)
    if not hint:
        shard = list(shard)          # plain pre-cut batches: no dp_info -> the Learner agrees on the batch size by all-reduce
    data = _Data(shard, 8 // world)
    net = _model(dropout=dropout)
    learner = Learner('/tmp/nnl_dist_test_%d_%d' % (world, rank), data, net, optimizer='Adam')
    if dropout:
        learner.use_keyed_dropout(seed=11)            # masks keyed by (seed, step, request, GLOBAL sample index)
    if world > 1:
        learner.distribute(bucket_mb=0.0005)          # tiny buckets: several collectives per step
        assert len(learner.grad_sync.buckets) > 1
    learner.fit(1e-2, 2, wd=1e-3, clip=1.0)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    if rank == 0:
        q.put((learner.loss_sched, flat.numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, **kw):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit, args=(r, world, port, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_fit_matches_single_rank():
    losses1, w1 = _run(1)
    losses2, w2 = _run(2)
    # rank 0's logged loss is the mean over ITS shard; the averaged gradients (hence the weights) must match exactly
    np.testing.assert_allclose(w2, w1, rtol=2e-5, atol=2e-6)
    assert len(losses1) == len(losses2) == 10


@pytest.mark.parametrize('world,last,hint', [(2, 7, True), (2, 7, False), (4, 6, True), (4, 3, True)])
def test_ragged_last_batch_unequal_and_ghost_shards(world, last, hint):
    """ADVICE r1 / SURVEY §8e: a ragged GLOBAL batch cut over the ranks gives unequal shards (7 over 2 = 4 + 3; 6 over 4 =
    2 + 2 + 1 + 1) or fewer rows than ranks (3 over 4: rank 3 runs a weight-0 ghost row).  The decision how to scale the lr is
    rank-uniform (no hang), gradients are weighted by local / global rows, and the result equals the single-process run."""
    _, w1 = _run(1, last=last, n_batches=3)
    _, wn = _run(world, last=last, n_batches=3, hint=hint)
    np.testing.assert_allclose(wn, w1, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('world,last', [(2, 8), (2, 7), (4, 6)])
def test_dropout_on_two_ranks_reproduce_one_rank_with_keyed_masks(world, last):
    """VERDICT r1 #8 / SURVEY §7 step 9: with `Learner.use_keyed_dropout` every dropout mask element is a function of (seed,
    step, request, global sample index), so an N-rank run with dropout ON equals the 1-rank run on the same global minibatches
    (equal, unequal and 1-row shards); with torch's per-process streams it could not."""
    l1, w1 = _run(1, last=last, n_batches=3, dropout=0.3)
    ln, wn = _run(world, last=last, n_batches=3, dropout=0.3)
    np.testing.assert_allclose(wn, w1, rtol=2e-5, atol=2e-6)
    l0, w0 = _run(1, last=last, n_batches=3, dropout=0.0)
    assert np.abs(w1 - w0).max() > 1e-4                     # (the dropout really was on)


def test_shard_bounds_cover_every_row_once():
    from neuralnetworklibrary_amd.dist import shard_bounds
    for world in (1, 2, 3, 4, 8):
        for n in range(1, 20):
            rows, sizes = [], []
            for r in range(world):
                a, z, ghost = shard_bounds(n, r, world)
                assert z > a                                    # never empty
                if not ghost:
                    rows += list(range(a, z)); sizes.append(z - a)
                else:
                    assert n < world and z - a == 1 and 0 <= a < n
            assert rows == list(range(n)) and max(sizes) - min(sizes) <= 1


def _gradless_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    torch.set_num_threads(1)
    nd.init_from_env('gloo')
    torch.manual_seed(0)
    a, b, c = nn.Linear(3, 1), nn.Linear(3, 1), nn.Linear(3, 1)       # a: used by all, b: by rank 1 only, c: by nobody
    net = nn.ModuleList([a, b, c])
    sync = nd.GradSync(net, bucket_mb=1e-6)                            # one bucket per tensor
    x = torch.ones(2, 3)
    sync.begin()
    out = a(x).sum() + (b(x).sum() if rank == 1 else 0.0)
    out.backward()
    sync.finish()
    if rank == 0:
        q.put([None if p.grad is None else p.grad.clone().numpy() for p in net.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_gradless_parameters_keep_none_and_collectives_stay_uniform():
    """ADVICE r1 (dist.finish): a parameter without a gradient on ANY rank keeps grad None (the optimizer skips it as the
    reference's does); one that only another rank reached gets the rank average; no rank skips a collective."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gradless_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ga_w, ga_b, gb_w, gb_b, gc_w, gc_b = got
    np.testing.assert_allclose(ga_w, np.full((1, 3), 2.0)); np.testing.assert_allclose(ga_b, [2.0])
    np.testing.assert_allclose(gb_w, np.full((1, 3), 1.0)); np.testing.assert_allclose(gb_b, [1.0])   # (0 + 2) / 2
    assert gc_w is None and gc_b is None


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    torch.set_num_threads(1)
    nd.init_from_env('gloo')
    net = _model(1)
    net[4].weight.requires_grad_(False)               # a frozen parameter is simply not synchronised
    sync = nd.GradSync(net, bucket_mb=0.0003)
    x, y = _batches(1, 8, 8)[0]
    xs, ys = nd.ShardedBatches([(x, y)], rank, world).__iter__().__next__()
    sync.begin()
    nn.MSELoss()(net(xs), ys).backward()
    sync.finish()
    if rank == 0:
        q.put([None if p.grad is None else p.grad.clone().numpy() for p in net.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_equals_global_batch_gradient():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _model(1)
    net[4].weight.requires_grad_(False)
    x, y = _batches(1, 8, 8)[0]
    nn.MSELoss()(net(x), y).backward()
    for g, p in zip(got, net.parameters()):
        if p.grad is None:
            assert g is None
        else:
            np.testing.assert_allclose(g, p.grad.numpy(), rtol=1e-5, atol=1e-7)
