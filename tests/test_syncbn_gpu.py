"""Cross-replica BatchNorm (nnl_bn_sync_*): a batch split over 2 "ranks" must give the results of the
single-replica kernels (and of torch's batch_norm in fp64) on the whole batch — forward, running statistics, dx, dgamma /
dbeta — first with an injected communicator in one process, then with 2 real processes (gloo, both on cuda:0) running a
data-parallel Learner with sync_bn=True against the 1-process run on the global batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


class FakeComm:
    """Plays the other rank: all_gather returns [mine, theirs] in rank order; all_reduce_sum adds the other rank's recorded
    sums (pass 1 records, pass 2 replays)."""

    def __init__(self, rank, other_stats):
        self.rank, self.other_stats, self.local_sums, self.other_sums = rank, other_stats, None, None

    def all_gather(self, t, group):
        pair = [t, self.other_stats] if self.rank == 0 else [self.other_stats, t]
        return torch.stack(pair).contiguous()

    def all_reduce_sum(self, t, group):
        self.local_sums = t.clone()
        return t + self.other_sums if self.other_sums is not None else t.clone()


def _local_stats(xm):
    from neuralnetworklibrary_amd._lib import check, lib, ptr, stream
    rows, C = xm.shape
    wsb = int(lib.nnl_bn_workspace_bytes(rows, C))
    ws = torch.empty(wsb // 4, device=DEV)
    st = torch.empty(2 * C + 2, device=DEV)
    check(lib.nnl_bn_sync_stats(ptr(xm), ptr(st), rows, C, ptr(ws), wsb, stream()))
    return st


@pytest.mark.parametrize('shape,split,relu,res', [((12, 16, 5, 7), 5, True, True), ((64, 10), 24, False, False),
                                                  ((6, 64, 9, 9), 3, True, False)])
def test_two_rank_syncbn_equals_full_batch(shape, split, relu, res):
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(11)
    cl = dict(memory_format=torch.channels_last) if len(shape) == 4 else {}
    x = (torch.randn(shape, generator=g) * 2 + 3).to(DEV).contiguous(**cl)
    r = torch.randn(shape, generator=g).to(DEV).contiguous(**cl) if res else None
    w = torch.randn(shape, generator=g).to(DEV).contiguous(**cl)
    C = shape[1]
    BN = nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d

    def make_bn():
        torch.manual_seed(1)
        bn = BN(C).to(DEV)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-1, 1)
        return bn.train()

    # whole batch, single-replica kernels
    bn_full = make_bn()
    xf = x.clone().requires_grad_(True)
    rf = None if r is None else r.clone().requires_grad_(True)
    yf = ops.bn_act(bn_full, xf, residual=rf, relu=relu)
    (yf * w).sum().backward()

    parts = [slice(0, split), slice(split, shape[0])]
    rows_view = lambda t: (t.permute(0, 2, 3, 1).reshape(-1, C) if t.dim() == 4 else t).contiguous()
    stats = [_local_stats(rows_view(x[p])) for p in parts]
    outs = {}
    recorded = [None, None]
    for pass_ in range(2):
        for rk in range(2):
            bn = make_bn()
            comm = FakeComm(rk, stats[1 - rk])
            comm.other_sums = recorded[1 - rk] if pass_ == 1 else None
            bn.nnl_sync = (None, comm)
            xr = x[parts[rk]].clone().contiguous(**cl).requires_grad_(True)
            rr = None if r is None else r[parts[rk]].clone().contiguous(**cl).requires_grad_(True)
            y = ops.bn_act(bn, xr, residual=rr, relu=relu)
            (y * w[parts[rk]]).sum().backward()
            if pass_ == 0:
                recorded[rk] = comm.local_sums
            outs[rk] = (y.detach(), xr.grad, None if rr is None else rr.grad, bn)
    y2 = torch.cat([outs[0][0], outs[1][0]])
    dx2 = torch.cat([outs[0][1], outs[1][1]])
    assert_close(y2, yf.detach(), 1e-5, 1e-5, 'y')
    assert_close(dx2, xf.grad, 2e-4, 2e-5, 'dx')
    if res:
        assert_close(torch.cat([outs[0][2], outs[1][2]]), rf.grad, 1e-6, 1e-6, 'dres')
    assert_close(outs[0][3].weight.grad + outs[1][3].weight.grad, bn_full.weight.grad, 2e-4, 2e-4, 'dgamma')
    assert_close(outs[0][3].bias.grad + outs[1][3].bias.grad, bn_full.bias.grad, 2e-4, 2e-4, 'dbeta')
    for rk in range(2):                                   # every rank holds the same (global) running statistics
        assert_close(outs[rk][3].running_mean, bn_full.running_mean, 1e-5, 1e-6, 'running_mean')
        assert_close(outs[rk][3].running_var, bn_full.running_var, 1e-5, 1e-6, 'running_var')
        assert int(outs[rk][3].num_batches_tracked) == 1
    assert torch.equal(outs[0][3].running_var, outs[1][3].running_var)       # bit-identical across ranks

    # and against torch's own batch_norm in fp64 on the whole batch
    xd = x.double().detach().requires_grad_(True)
    yd = F.batch_norm(xd, None, None, bn_full.weight.double().detach(), bn_full.bias.double().detach(), True, 0.1, bn_full.eps)
    if r is not None:
        yd = yd + r.double()
    if relu:
        yd = yd.relu()
    (yd * w.double()).sum().backward()
    assert_close(y2, yd.detach(), 1e-4, 1e-4, 'y vs fp64')
    assert_close(dx2, xd.grad, 1e-3, 1e-3, 'dx vs fp64')


# ---- 2 real processes ---------------------------------------------------------------------------------------------------
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
            self.fc = nn.Linear(16, 1)

        def forward(self, x):
            h = ops.bn_act(self.b1, self.c1(x), relu=True)
            h = ops.bn_act(self.b2, self.c2(h), residual=h, relu=True)
            return self.fc(h.mean(dim=(2, 3))).flatten()

    torch.manual_seed(0)
    return make_model_basic(Net())


def _batches():
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(8, 4, 6, 6, generator=g), torch.randn(8, generator=g)) for _ in range(3)]


class _Data:
    target_type = 'cont'

    def __init__(self, batches, bs):
        self.train_dl, self.val_dl, self.bs = batches, batches, bs


def _fit(rank, world, port, q):
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    batches = [(x.to(DEV), y.to(DEV)) for x, y in _batches()]
    data = _Data(nd.ShardedBatches(batches, rank, world), 8 // world)
    learner = Learner('/tmp/nnl_syncbn_%d_%d' % (world, rank), data, _net(), optimizer='SGD_Mom')
    if world > 1:
        learner.distribute(sync_bn=True)
    learner.model.train()
    losses = [learner.train1minibatch(x, y, 5e-2, mom_batch=0.9) for x, y in data.train_dl]
    sd = {k: v.detach().cpu().numpy() for k, v in learner.model.state_dict().items()}
    q.put((rank, losses, sd))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, target=None):
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=target or _fit, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_process_dp_with_syncbn_reproduces_single_process():
    one = _run(1)[0]
    two = _run(2)
    for k in one[2]:
        assert_close(two[0][2][k], two[1][2][k], 0, 1e-7, 'replicas agree: ' + k)
        assert_close(two[0][2][k], one[2][k], 2e-4, 2e-5, k)
    # the global loss is the mean of the equal-sized shards' losses
    assert_close(np.mean([two[0][1], two[1][1]], axis=0), np.array(one[1]), 1e-4, 1e-6, 'loss curve')


# ---- data-parallel max_norm renorm of the tabular embeddings -------------------------------------------------------------
def test_renorm_covers_all_ranks_lookups():
    """ops.tab_embed_concat(sync=...): rows looked up by the OTHER rank are renormalised too (and no others)."""
    from neuralnetworklibrary_amd import ops

    class Comm:
        def __init__(self, other):
            self.other = other

        def all_gather(self, t, group):
            return torch.stack([t, self.other]).contiguous()

    cards, dims, cap = [7, 5], [4, 3], 4
    torch.manual_seed(2)
    W = [nn.Parameter((torch.randn(c, d) * 3).to(DEV)) for c, d in zip(cards, dims)]        # most row norms > 1.5
    W0 = [w.detach().clone() for w in W]
    mine = torch.tensor([[0, 1], [2, 1]], device=DEV)                                        # 2 local rows (ragged: < cap)
    theirs = torch.tensor([[5, 4], [6, 0], [5, 3]], device=DEV)
    other = torch.zeros(cap + 1, 2, dtype=torch.int64, device=DEV)
    other[:3] = theirs
    other[cap, 0] = 3
    out, _ = ops.tab_embed_concat(mine, W, max_norm=1.5, sync=(None, Comm(other), cap))
    for j in range(2):
        touched = set(mine[:, j].tolist()) | set(theirs[:, j].tolist())
        for r in range(cards[j]):
            n0 = float(W0[j][r].norm())
            want = W0[j][r] * (1.5 / (n0 + 1e-7)) if (r in touched and n0 > 1.5) else W0[j][r]
            assert_close(W[j][r].detach(), want, 1e-6, 1e-6, f'table {j} row {r}')
    assert_close(out[:, :4], W[0].detach()[mine[:, 0]], 0, 0, 'gather')


def _tab_net():
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    cards = [9, 6, 4]
    torch.manual_seed(0)
    net = StructuredDataNet('cont', 3, 2, [{i: i for i in range(c)} for c in cards], [12, 1], output_range=[0, 1],
                            dropout_levels=(0.0, 0.0, [0, 0.0]))
    with torch.no_grad():
        for e in net.embeddings:
            e.emb.weight.mul_(3.0)                        # norms above max_norm = 1.5: the renorm matters
    return net


def _tab_batches():
    rs = np.random.RandomState(9)
    out = []
    for _ in range(4):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=8) for c in [9, 6, 4]], 1).astype(np.int64))
        out.append(([xcat, torch.from_numpy(rs.standard_normal((8, 2)).astype(np.float32))], torch.from_numpy(rs.rand(8).astype(np.float32))))
    return out


def _fit_tab(rank, world, port, q):
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    batches = [([x[0].to(DEV), x[1].to(DEV)], y.to(DEV)) for x, y in _tab_batches()]
    data = _Data(nd.ShardedBatches(batches, rank, world), 8 // world)
    learner = Learner('/tmp/nnl_tabdp_%d_%d' % (world, rank), data, _tab_net(), optimizer='SGD_Mom')
    if world > 1:
        learner.distribute(sync_bn=True)
    learner.model.train()
    losses = [learner.train1minibatch(x, y, [5e-2, 5e-2], mom_batch=0.9) for x, y in data.train_dl]
    sd = {k: v.detach().cpu().numpy() for k, v in learner.model.state_dict().items()}
    q.put((rank, losses, sd))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_process_tabular_dp_keeps_tables_in_sync_and_matches_single_process():
    one = _run(1, _fit_tab)[0]
    two = _run(2, _fit_tab)
    for k in one[2]:
        assert_close(two[0][2][k], two[1][2][k], 0, 1e-7, 'replicas agree: ' + k)
        assert_close(two[0][2][k], one[2][k], 2e-4, 2e-5, k)


# ---- dropout ON under data parallelism: keyed masks (Learner.use_keyed_dropout) on the PRODUCT tabular net ---------------
def _fit_tab_dropout(rank, world, port, q):
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    batches = [([x[0].to(DEV), x[1].to(DEV)], y.to(DEV)) for x, y in _tab_batches()]
    data = _Data(nd.ShardedBatches(batches, rank, world), 8 // world)
    cards = [9, 6, 4]
    torch.manual_seed(0)
    net = StructuredDataNet('cont', 3, 2, [{i: i for i in range(c)} for c in cards], [12, 1], output_range=[0, 1],
                            dropout_levels=(0.25, 0.2, [0.1, 0.3]))       # row dropout, continuous dropout, head dropouts: all ON
    learner = Learner('/tmp/nnl_tabdrop_%d_%d' % (world, rank), data, net, optimizer='SGD_Mom')
    learner.use_keyed_dropout(seed=7)
    if world > 1:
        learner.distribute(sync_bn=True)
    learner.model.train()
    losses = [learner.train1minibatch(x, y, [5e-2, 5e-2], mom_batch=0.9) for x, y in data.train_dl]
    sd = {k: v.detach().cpu().numpy() for k, v in learner.model.state_dict().items()}
    q.put((rank, losses, sd))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_process_tabular_dp_with_dropout_on_reproduces_single_process():
    """with masks keyed by (seed, step, request, GLOBAL sample index) the 2-rank run with EmbeddingDrop row
    dropout, continuous dropout and the head's nn.Dropouts all ON trains exactly like the 1-rank run on the same global
    minibatches (the product layers on the GPU; gloo between two processes on one GPU)."""
    one = _run(1, _fit_tab_dropout)[0]
    two = _run(2, _fit_tab_dropout)
    for k in one[2]:
        assert_close(two[0][2][k], two[1][2][k], 0, 1e-7, 'replicas agree: ' + k)
        assert_close(two[0][2][k], one[2][k], 2e-4, 2e-5, k)
    assert_close(np.mean([two[0][1], two[1][1]], axis=0), np.array(one[1]), 1e-4, 1e-6, 'loss curve')


# ---- round 5: the data-parallel tabular step REPLAYS (the renorm sync's all-gather runs before the hipGraph) -------------------
def _fit_tab_replay(rank, world, port, q, graphs):
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    batches = [([x[0].to(DEV), x[1].to(DEV)], y.to(DEV)) for x, y in _tab_batches()] * 3            # 12 steps
    data = _Data(nd.ShardedBatches(batches, rank, world), 8 // world)
    learner = Learner('/tmp/nnl_tabreplay_%d_%d_%d' % (world, rank, int(graphs)), data, _tab_net(), optimizer='SGD_Mom')
    learner.distribute()                                    # local BatchNorm (SyncBN's collectives sit inside both passes: that step stays eager)
    learner.use_graphs(bool(graphs), warmup=2)
    learner.model.train()
    assert getattr(learner.model, 'nnl_dp', None) is not None          # the renorm sync is active
    losses = [learner.train1minibatch(x, y, [5e-2, 5e-2], mom_batch=0.9) for x, y in data.train_dl]
    learner.grad_sync.raise_if_overlap_error()
    n_graphs = sum(g.graph is not None for g in learner._graphs.values())
    sd = {k: v.detach().cpu().numpy() for k, v in learner.model.state_dict().items()}
    q.put((rank, losses, sd, n_graphs))
    dist.barrier()
    dist.destroy_process_group()


def _fit_tab_replay_on(rank, world, port, q):
    _fit_tab_replay(rank, world, port, q, True)


def _fit_tab_replay_off(rank, world, port, q):
    _fit_tab_replay(rank, world, port, q, False)


def test_two_process_tabular_dp_replays_with_the_renorm_sync_active():
    """config 3 under distribute() used to fall back to the eager step because the forward contained the renorm sync's
    all-gather.  The gather only needs the step's INPUT indices, so it now runs before the graph into a static buffer
    (StructuredDataNet.nnl_dp_prepare) and the step is captured and replayed: same training as eager data parallelism, embedding
    tables identical on both ranks (the renorm hit every rank's rows), ten replays."""
    eager, replay = _run(2, _fit_tab_replay_off), _run(2, _fit_tab_replay_on)
    assert eager[0][3] == 0 and replay[0][3] == 1 and replay[1][3] == 1
    for k in replay[0][2]:
        if 'running_' not in k and 'num_batches' not in k:            # (local BatchNorm: the buffers are per-replica statistics)
            assert np.array_equal(replay[0][2][k], replay[1][2][k]), 'replicas agree bitwise: ' + k
        for r in range(2):
            assert_close(replay[r][2][k], eager[r][2][k], 2e-5, 2e-6, 'replayed vs eager data parallelism: ' + k)
    for r in range(2):
        assert_close(np.array(replay[r][1]), np.array(eager[r][1]), 1e-5, 1e-6, 'loss curve')
