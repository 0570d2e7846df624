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


def _fit(rank, world, port, q, last=6, n_batches=5, hint=True, dropout=0.0, bs64=False):
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
    gbs = 64 if bs64 else 8
    batches = _batches(n_batches, gbs, last)
    shard = nd.ShardedBatches(batches, rank, world)
    if not hint:
        shard = list(shard)          # plain pre-cut batches: no dp_info -> the Learner agrees on the batch size by all-reduce
    data = _Data(shard, gbs // world)
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


# ---- world size 8: the shape of the driver's N = 8 run -------------------------------------------------
def test_eight_ranks_global_batch_64_with_ragged_last_batch_61():
    """BASELINE's strong-scaling shape rehearsed on gloo: global minibatch 64 over 8 ranks (8 rows each), ragged LAST batch of 61 =
    8+8+8+8+8+7+7+7 (dist.shard_bounds), lr scaled by 61/64 on every rank, gradients weighted by local / global rows — final
    weights equal the single-process run on the same global minibatches."""
    _, w1 = _run(1, n_batches=3, last=61, bs64=True)
    _, w8 = _run(8, n_batches=3, last=61, bs64=True)
    np.testing.assert_allclose(w8, w1, rtol=2e-5, atol=2e-6)


class _TinyLM(nn.Module):
    """a language model with the product LSTM_Encoder's state semantics (Text.py:535-551): hidden state carried across
    minibatches per STREAM (dim 0 of the batch), detached after every forward, never reset by the Learner"""

    def __init__(self, V=23, E=6, H=10, bs=64):
        super().__init__()
        self.emb, self.lstm, self.dec = nn.Embedding(V, E), nn.LSTM(E, H), nn.Linear(H, V)
        self.h, self.c = torch.zeros(1, bs, H), torch.zeros(1, bs, H)

    def forward(self, x):                                   # x [bs, seq]
        out, (h, c) = self.lstm(self.emb(x.t()), (self.h, self.c))
        self.h, self.c = h.detach(), c.detach()
        return self.dec(out).permute(1, 2, 0)               # [bs, V, seq]


def _lm_fit(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.General.Core import make_model_basic, set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device('cpu')
    torch.set_num_threads(1)
    Learner.verbose = False
    if world > 1:
        nd.init_from_env('gloo')
    bs, seq, V = 64, 5, 23
    stream = np.random.RandomState(7).randint(0, V, size=(bs, 3 * seq + 1)).astype(np.int64)
    batches = [(torch.from_numpy(stream[:, i * seq:(i + 1) * seq].copy()), torch.from_numpy(stream[:, i * seq + 1:(i + 1) * seq + 1].copy()))
               for i in range(3)]                           # three CONSECUTIVE windows of the 64 streams
    torch.manual_seed(0)
    net = make_model_basic(_TinyLM(V, bs=bs // world))
    data = _Data(nd.ShardedBatches(batches, rank, world), bs // world)
    data.val_dl = []                                        # (a validation pass would advance the carried state: Learner never resets it)
    data.target_type = 'lang_model'
    learner = Learner('/tmp/nnl_dist_lm_%d_%d' % (world, rank), data, net, optimizer='Adam', loss_func=nn.CrossEntropyLoss())
    learner.evaluate = lambda *a, **k: [0.0]
    if world > 1:
        learner.distribute(bucket_mb=0.001, equal_shards=True)
    learner.fit(1e-2, 1, wd=1e-4)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    q.put((rank, flat.numpy(), net.h.numpy().copy(), net.c.numpy().copy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_lm_stream_sharding_64_over_8_keeps_the_carried_state_rank_local():
    """SURVEY §8e / Text.py:254-263,531-551: the language model's minibatch is split along the STREAM dimension (64 streams over 8
    ranks = 8 each), every rank keeps its own streams and their carried (h, c) from batch to batch.  After three consecutive
    minibatches the weights equal the 1-rank run and rank r's carried state equals rows [8r, 8r+8) of the 1-rank state."""
    def run(world):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_lm_fit, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return got
    (_, w1, h1, c1), = run(1)
    got8 = run(8)
    for r, w, h, c in got8:
        np.testing.assert_allclose(w, w1, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(h, h1[:, 8 * r:8 * r + 8], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(c, c1[:, 8 * r:8 * r + 8], rtol=2e-5, atol=2e-6)


# ---- the replay path of Learner.use_graphs() under data parallelism (round 4: per-bucket segments) -------------------------------------
def _replay_protocol(rank, world, port, q):
    """Each rank runs the SAME three data-parallel steps twice from the same weights: (a) eagerly — the hooks launch bucket k's all-reduce
    as soon as its gradients are complete, in the middle of backward — and (b) through the protocol of a captured step: hooks armed in
    `capturing` mode during backward (buckets are filled, bucket completions are recorded in order, NO collective is issued), then
    `reduce_overlapped()` issues the collectives bucket by bucket in bucket order (on the GPU each behind its wait kernel; on CPU tensors
    the device-side signals do not exist and the collectives follow the backward).  Both must leave bitwise the same averaged gradients
    and weights, and every rank must have issued the same number of collectives in the same order."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    torch.set_num_threads(1)
    nd.init_from_env('gloo')
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 6, generator=g) for _ in range(3)]
    ys = [torch.randn(8, generator=g) for _ in range(3)]
    a, z = rank * 4, rank * 4 + 4
    results = []
    for mode in ('eager', 'replay'):
        net = _model(seed=1)
        gs = nd.GradSync(net, bucket_mb=0.0001)
        assert len(gs.buckets) >= 3
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        order = []
        launch = gs._launch
        gs._launch = lambda b, _l=launch: (order.append(gs.buckets.index(b)), _l(b))[1]
        launch_f = gs._launch_filled
        gs._launch_filled = lambda b, _l=launch_f: (order.append(gs.buckets.index(b)), _l(b))[1]
        for x, y in zip(xs, ys):
            opt.zero_grad(set_to_none=True)
            loss = ((net(x[a:z]) - y[a:z]) ** 2).mean()
            if mode == 'eager':
                gs.begin(1.0)
                loss.backward()
                gs.finish()
            else:
                gs.begin(1.0)
                gs.capturing = True
                gs.prepare_overlap()                       # CPU tensors: no device-side signals (overlap is None)
                gs.capture_begin()
                loss.backward()                            # "capture": buckets filled, nothing sent
                gs.capture_end()
                gs.capturing, gs._active = False, False
                assert order == [] or len(order) % len(gs.buckets) == 0
                gs.reduce_overlapped(1.0)                  # "after the replay": collectives in bucket order
            opt.step()
        assert order == list(range(len(gs.buckets))) * 3, order
        results.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [p.grad.reshape(-1) for p in net.parameters()]))
    assert torch.equal(results[0], results[1]), 'replay protocol != eager data parallelism'
    # and both equal the single-process gradient step on the concatenated batch
    net = _model(seed=1)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for x, y in zip(xs, ys):
        opt.zero_grad(set_to_none=True)
        ((net(x) - y) ** 2).mean().backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(results[1][:ref.numel()], ref, rtol=1e-5, atol=1e-6)
    if rank == 0:
        q.put('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_replay_protocol_collectives_in_bucket_order_equal_eager_dp():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replay_protocol, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=180) == 'ok'
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
