"""Learner.use_graphs(): the captured-and-replayed training step must train exactly like the eager step (same kernels,
hyper-parameters read from device memory), including lr / momentum / betas schedules and the ragged last minibatch."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


class Data:
    def __init__(self, batches, bs, target_type):
        self.train_dl, self.val_dl, self.bs, self.target_type = batches, batches[:1], bs, target_type


def _train(make, batches, bs, optimizer, graphs, sched, wd, clip=None, **kw):
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    torch.manual_seed(0)
    net = make()
    learner = Learner('/tmp/nnl_graph_test', Data(batches, bs, 'cont'), net, optimizer=optimizer)
    learner.init_optimizer(wd=wd, clip=clip)
    learner.use_graphs(bool(graphs), warmup=2)        # (collab / tabular nets replay by default: the eager leg switches it off)
    learner.model.train()
    losses = []
    for i, lr in enumerate(sched):
        x, y = batches[i % len(batches)]
        extra = {k: v[i] for k, v in kw.items()}
        losses.append(learner.train1minibatch(x, y, lr, **extra))
    n_graphs = sum(g.graph is not None for g in learner._graphs.values())
    return losses, [p.detach().cpu().numpy().copy() for p in net.parameters()], n_graphs


def test_collab_adam_schedule_replay_matches_eager():
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    g = torch.Generator().manual_seed(3)
    def batch(n):
        return (torch.stack([torch.randint(0, 50, (n,), generator=g), torch.randint(0, 70, (n,), generator=g)], 1).to(DEV),
                torch.randint(1, 6, (n,), generator=g).float().to(DEV))
    batches = [batch(64), batch(64), batch(64), batch(17)]            # the last one is ragged -> eager (then its own graph)
    sched = [1e-2 * (1 + 0.3 * i) for i in range(12)]
    betas = [(0.9 - 0.01 * i, 0.99) for i in range(12)]
    make = lambda: CollabFilterNet(50, 70, 12, [0.8, 5.2])
    le, pe, ne = _train(make, batches, 64, 'Adam', False, sched, 1e-3, betas_batch=betas)
    lg, pg, ng = _train(make, batches, 64, 'Adam', True, sched, 1e-3, betas_batch=betas)
    assert ne == 0 and ng >= 1
    assert_close(np.array(lg), np.array(le), 1e-5, 1e-6, 'losses')
    for a, b in zip(pg, pe):
        assert_close(a, b, 1e-4, 1e-6, 'params')


def test_tabular_sgd_momentum_clip_replay_matches_eager():
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    cards = [11, 5, 7]
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(3):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=32) for c in cards], 1).astype(np.int64)).to(DEV)
        xcont = torch.from_numpy(rs.standard_normal((32, 4)).astype(np.float32)).to(DEV)
        batches.append(([xcat, xcont], torch.from_numpy(rs.rand(32).astype(np.float32)).to(DEV)))
    make = lambda: StructuredDataNet('cont', 8, 4, [{i: i for i in range(c)} for c in cards], [24, 1], output_range=[0, 1],
                                     dropout_levels=(0.0, 0.0, [0, 0.0]))
    sched = [[3e-2 * (1 + i % 3)] * 2 for i in range(9)]
    mom = [0.9 - 0.02 * i for i in range(9)]
    le, pe, _ = _train(make, batches, 32, 'SGD_Mom', False, sched, 1e-2, clip=0.5, mom_batch=mom)
    lg, pg, ng = _train(make, batches, 32, 'SGD_Mom', True, sched, 1e-2, clip=0.5, mom_batch=mom)
    assert ng == 1
    assert_close(np.array(lg), np.array(le), 1e-5, 1e-6, 'losses')
    for a, b in zip(pg, pe):
        assert_close(a, b, 1e-4, 1e-6, 'params')


def test_graph_is_dropped_on_freeze_and_refused_with_grad_sync():
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    from neuralnetworklibrary_amd.General.Learner import Learner
    batches = [(torch.stack([torch.randint(0, 9, (8,)), torch.randint(0, 9, (8,))], 1).to(DEV), torch.rand(8).to(DEV))]
    learner = Learner('/tmp/nnl_graph_test', Data(batches, 8, 'cont'), CollabFilterNet(9, 9, 4, None), optimizer='SGD')
    learner.init_optimizer()
    learner.use_graphs(True, warmup=1)
    learner.model.train()
    for _ in range(3):
        learner.train1minibatch(*batches[0], 1e-2)
    assert len(learner._graphs) == 1
    learner._new_optimizer()
    assert learner._graphs == {}


def test_resnet_classifier_graph_replay_trains_like_eager():
    """A conv net with BatchNorm, pooling, the balanced conv schedule's in-kernel fix-up, the shortcut-gradient fusion and the
    HIP cross-entropy: the replayed step must produce the SAME losses as the eager step (same kernels, same order)."""
    from neuralnetworklibrary_amd.Applications import Vision as V
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    N, S = 16, 64
    g = torch.Generator().manual_seed(4)
    batches = [(torch.randn(N, 3, S, S, generator=g).to(DEV), torch.randint(0, 2, (N,), generator=g).to(DEV)) for _ in range(3)]

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
        train_dl = val_dl = batches

    def run(graphs):
        torch.manual_seed(0)
        net = V.ImageClassificationNet(D, V.models.resnet18(), head=[[64], [0., 0.]])
        learner = Learner('/tmp/nnl_graph_test', D, net, optimizer='SGD_Mom')
        learner.init_optimizer(wd=1e-4)
        if graphs:
            learner.use_graphs(True, warmup=2)
        net.train()
        losses = [learner.train1minibatch(*batches[i % 3], [1e-3, 2e-3, 5e-3], mom_batch=0.9) for i in range(9)]
        return np.array(losses), sum(gs.graph is not None for gs in learner._graphs.values())
    le, _ = run(False)
    lg, ng = run(True)
    assert ng == 1
    assert_close(lg, le, 1e-5, 1e-6, 'losses: graph replay vs eager')


def test_resnet34_at_8_images_224_graph_replay_trains_like_eager():
    """the strong-scaling regime — ResNet-34 + default head at 224 x 224 with 8 images per GPU (global batch 64 over 8
    GPUs), where the dispatcher takes the small-grid plans (the position-split 2-D Winograd instantiation of round 5, k-sliced
    direct tiles, Winograd-domain weight gradients on tiny grids) and the replayed step captures ONE batched filter-transform launch:
    the replayed run must reproduce the eager run's losses (same kernels, same order), and the run with the position-split plan
    switched off must agree with it on the first steps."""
    from neuralnetworklibrary_amd.Applications import Vision as V
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    from neuralnetworklibrary_amd._lib import lib
    import os
    set_default_device(DEV)
    Learner.verbose = False
    N, S = 8, 224
    g = torch.Generator().manual_seed(6)
    batches = [(torch.randn(N, 3, S, S, generator=g).to(DEV), torch.randint(0, 2, (N,), generator=g).to(DEV)) for _ in range(3)]

    class D:
        sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
        train_dl = val_dl = batches

    def run(graphs):
        torch.manual_seed(0)
        net = V.ImageClassificationNet(D, V.models.resnet34(), head=[[512], [0., 0.]])
        learner = Learner('/tmp/nnl_graph_test', D, net, optimizer='SGD_Mom')
        learner.init_optimizer(wd=1e-4)
        if graphs:
            learner.use_graphs(True, warmup=2)
        net.train()
        losses = [learner.train1minibatch(*batches[i % 3], [1e-4, 2e-4, 1e-3], mom_batch=0.9) for i in range(8)]
        return np.array(losses), sum(gs.graph is not None for gs in learner._graphs.values())
    le, _ = run(False)
    lg, ng = run(True)
    assert ng == 1
    assert_close(lg, le, 1e-5, 1e-6, 'losses at 8 images: graph replay vs eager')
    os.environ['NNL_WINO2_POS'] = '0'; lib.nnl_reload_env()
    try:
        l0, _ = run(False)
    finally:
        os.environ.pop('NNL_WINO2_POS'); lib.nnl_reload_env()
    # (only the first two losses are comparable: with training-mode BatchNorm at 8 images and random labels the trajectory is chaotic —
    # two correct fp32 runs are 2x apart by step 5; the kernels themselves are pinned in test_conv_gpu.py::test_winograd_2d_position_split)
    assert_close(le[:1], l0[:1], 1e-4, 1e-6, 'first loss at 8 images: with / without the position-split 2-D Winograd plan')
    assert_close(le[:2], l0[:2], 2e-2, 1e-5, 'second loss (one update later; BatchNorm at 8 images: 4e-3 seen between two correct fp32 runs)')


def test_resnet_graph_replay_under_data_parallelism_matches_eager_dp():
    """Learner.use_graphs() with a GradSync attached: the captured forward + backward fills the all-reduce buckets (in-place
    wgrad writes and the hooks' copies are part of the graph), every replay is followed by the eager bucket all-reduces and the
    fused optimizer launch.  Run here on one GPU with a one-rank RCCL group and NNL_DIST_FORCE_ALLREDUCE semantics, so the RCCL
    calls really are issued: the replayed run must train like the eager data-parallel run and like the plain eager run."""
    import os
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.Applications import Vision as V
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    forced = nd._FORCE_ALLREDUCE
    nd._FORCE_ALLREDUCE = True
    try:
        N, S = 8, 64
        g = torch.Generator().manual_seed(4)
        batches = [(torch.randn(N, 3, S, S, generator=g).to(DEV), torch.randint(0, 2, (N,), generator=g).to(DEV)) for _ in range(3)]

        class D:
            sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, N, 'single_label'
            train_dl = val_dl = batches

        def run(dp, graphs, overlap=True):
            os.environ['NNL_DIST_REPLAY_OVERLAP'] = '1' if overlap else '0'
            torch.manual_seed(0)
            net = V.ImageClassificationNet(D, V.models.resnet18(), head=[[64], [0., 0.]])
            learner = Learner('/tmp/nnl_graph_test', D, net, optimizer='SGD_Mom')
            learner.init_optimizer(wd=1e-4)
            if dp:
                learner.distribute(bucket_mb=8.0, equal_shards=True)         # resnet18: 45 MB of gradients -> 6 buckets
                assert len(learner.grad_sync.buckets) >= 5
            if graphs:
                learner.use_graphs(True, warmup=2)
            net.train()
            losses = [learner.train1minibatch(*batches[i % 3], [1e-3, 2e-3, 5e-3], mom_batch=0.9) for i in range(9)]
            ng = sum(gs.graph is not None for gs in learner._graphs.values())
            gsync = learner.grad_sync
            if gsync is not None:
                gsync.raise_if_overlap_error()
            return np.array(losses), [p.detach().cpu().numpy().copy() for p in net.parameters()], ng, gsync
        l0, p0, _, _ = run(False, False)
        l1, p1, n1, _ = run(True, False)
        l2, p2, n2, g2 = run(True, True)
        l3, p3, n3, g3 = run(True, True, overlap=False)
        os.environ.pop('NNL_DIST_REPLAY_OVERLAP', None)
        assert n1 == 0 and n2 == 1 and n3 == 1
        # round 4: the captured backward carries one signal kernel per bucket; every replay's collectives were enqueued behind wait kernels
        # on the side stream (7 replays — the capture step's own and six more — x all buckets), none timed out; with the overlap switched off they follow the whole replay
        nb = len(g2.buckets)
        assert g2.overlap is not None and g2.last_signalled == nb and g2.overlap_launches == 7 * nb
        assert int(g2.overlap.flags[:nb].min().item()) == g2.overlap.replays == 7 and int(g2.overlap.step.item()) == 7
        assert g3.overlap is None and g3.overlap_launches == 0
        assert_close(l1, l0, 1e-5, 1e-6, 'eager DP vs eager')
        assert_close(l2, l1, 1e-5, 1e-6, 'graph DP (overlapped collectives) vs eager DP')
        assert_close(l3, l1, 1e-5, 1e-6, 'graph DP (collectives after the replay) vs eager DP')
        for a, b in zip(p2, p1):
            assert_close(a, b, 1e-4, 1e-6, 'params')
        for a, b in zip(p3, p2):
            assert np.array_equal(a, b), 'overlapped and non-overlapped replays must be bitwise identical'
    finally:
        nd._FORCE_ALLREDUCE = forced
        if created:
            dist.destroy_process_group()


def test_graph_replay_under_data_parallelism_with_two_input_signatures():
    """under data parallelism the ragged last minibatch gives a second per-rank input shape, so a second graph is
    captured next to the first.  Both graphs must share ONE set of protocol words (step counter, flags) — the first version replaced
    them at the second capture: the first graph's signal kernels then wrote into freed memory and every later wait spun to its bound.
    Here the two shapes alternate for 16 steps; the run must train like eager DP, every replay's collectives must have gone out
    behind wait kernels and no wait may have timed out."""
    import os
    import torch.distributed as dist
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.Applications import Vision as V
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    forced = nd._FORCE_ALLREDUCE
    nd._FORCE_ALLREDUCE = True
    try:
        S = 64
        g = torch.Generator().manual_seed(7)
        batches = [(torch.randn(n, 3, S, S, generator=g).to(DEV), torch.randint(0, 2, (n,), generator=g).to(DEV)) for n in (8, 6, 8, 6)]

        class D:
            sz, categories, bs, target_type = (S, S), {0: 'a', 1: 'b'}, 8, 'single_label'
            train_dl = val_dl = batches

        def run(graphs):
            torch.manual_seed(0)
            net = V.ImageClassificationNet(D, V.models.resnet18(), head=[[64], [0., 0.]])
            learner = Learner('/tmp/nnl_graph_test', D, net, optimizer='SGD_Mom')
            learner.init_optimizer(wd=1e-4)
            learner.distribute(bucket_mb=8.0, equal_shards=True)
            if graphs:
                learner.use_graphs(True, warmup=2)
            net.train()
            losses = [learner.train1minibatch(*batches[i % 4], [1e-3, 2e-3, 5e-3], mom_batch=0.9) for i in range(16)]
            learner.grad_sync.raise_if_overlap_error()
            return np.array(losses), [p.detach().cpu().numpy().copy() for p in net.parameters()], learner
        le, pe, _ = run(False)
        lg, pg, learner = run(True)
        gs = learner.grad_sync
        graphs = [v for v in learner._graphs.values() if v.graph is not None]
        assert len(graphs) == 2
        nb = len(gs.buckets)
        # 16 steps = 2 signatures x (2 eager warm-ups + 6 replays, the capture step's own included)
        assert gs.overlap.replays == 12 and int(gs.overlap.step.item()) == 12 and gs.overlap_launches == 12 * nb
        assert int(gs.overlap.flags[:nb].min().item()) == 12 and int(gs.overlap.err.item()) == 0
        assert all(v.signalled == nb for v in graphs)
        assert_close(lg, le, 1e-5, 1e-6, 'graph DP with two signatures vs eager DP')
        for a, b in zip(pg, pe):
            assert_close(a, b, 1e-4, 1e-6, 'params')
    finally:
        nd._FORCE_ALLREDUCE = forced
        if created:
            dist.destroy_process_group()


def test_replay_overlap_wait_times_out_loudly_and_falls_back():
    """a wait kernel whose signal never comes must not let training go on silently — its time bound is WALL time
    (device clock), the error is raised at the step's own loss read-back, the word is reset and later replays reduce after the
    whole replay.  Driven directly: a wait for a flag value that nobody publishes, with a 20 ms bound."""
    from neuralnetworklibrary_amd import dist as nd
    import time
    lin = torch.nn.Linear(8, 8).to(DEV)
    gs = nd.GradSync(lin, bucket_mb=1.0)
    gs.prepare_overlap()
    gs.capture_end()
    ov = gs.overlap
    assert ov is not None
    ov.timeout_us = 20000
    ov.replays = 5                                          # nobody bumped the device step word / set the flag to 5
    t0 = time.time()
    ov.wait(0)
    ov.side.synchronize()
    dt = time.time() - t0
    assert 0.015 < dt < 5.0, dt                             # the bound is time, not a poll count
    torch.cuda.current_stream().wait_stream(ov.side)
    ov.stage_err(gs.buckets[0].status)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='timed out'):
        gs.raise_if_overlap_error(synced=True)
    assert gs.overlap_ok is False and int(ov.err.item()) == 0
    gs.raise_if_overlap_error()                             # reset: nothing to raise any more


def test_keyed_dropout_keeps_the_step_eager():
    """keyed masks bake the Python step counter into the launch arguments, so a captured graph would replay the
    capture step's masks for ever — use_graphs() + use_keyed_dropout() must run eagerly (masks change from step to step)."""
    from neuralnetworklibrary_amd import dist as nnl_dist
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    cards = [11, 5, 7]
    rs = np.random.RandomState(5)
    xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=32) for c in cards], 1).astype(np.int64)).to(DEV)
    xcont = torch.from_numpy(rs.standard_normal((32, 4)).astype(np.float32)).to(DEV)
    batch = ([xcat, xcont], torch.from_numpy(rs.rand(32).astype(np.float32)).to(DEV))
    torch.manual_seed(0)
    net = StructuredDataNet('cont', 8, 4, [{i: i for i in range(c)} for c in cards], [24, 1], output_range=[0, 1],
                            dropout_levels=(0.3, 0.3, [0, 0.5]))
    learner = Learner('/tmp/nnl_graph_test', Data([batch], 32, 'cont'), net, optimizer='SGD')
    learner.init_optimizer(wd=0.0)
    learner.use_graphs(True, warmup=1).use_keyed_dropout(seed=7)
    try:
        learner.model.train()
        losses = [learner.train1minibatch(batch[0], batch[1], 0.0) for _ in range(6)]     # lr 0: only the masks change
        assert sum(g.graph is not None for g in learner._graphs.values()) == 0
        assert len(set(losses)) > 3, losses
    finally:
        nnl_dist.drop_ctx.enabled = False


def test_launch_bound_heads_replay_by_default(monkeypatch):
    """CollabFilterNet / StructuredDataNet are marked `nnl_default_graphs`: a Learner on the GPU replays their whole step as a
    hipGraph without the notebook asking for it (after the eager warm-up steps), `use_graphs(False)` and NNL_DEFAULT_GRAPHS=0
    keep the per-launch path, and other models are unaffected."""
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    from neuralnetworklibrary_amd.General.Core import make_model_basic, set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    g = torch.Generator().manual_seed(3)
    batch = (torch.stack([torch.randint(0, 50, (64,), generator=g), torch.randint(0, 70, (64,), generator=g)], 1).to(DEV),
             torch.randint(1, 6, (64,), generator=g).float().to(DEV))
    learner = Learner('/tmp/nnl_graph_test', Data([batch], 64, 'cont'), CollabFilterNet(50, 70, 12, [0.8, 5.2]), optimizer='Adam')
    assert learner._graph_warmup is not None
    learner.init_optimizer(wd=1e-4)
    learner.model.train()
    for _ in range(5):
        learner.train1minibatch(batch[0], batch[1], 1e-2)
    assert sum(gs.graph is not None for gs in learner._graphs.values()) == 1
    learner.use_graphs(False)
    assert learner._graph_warmup is None
    monkeypatch.setenv('NNL_DEFAULT_GRAPHS', '0')
    assert Learner('/tmp/nnl_graph_test', Data([batch], 64, 'cont'), CollabFilterNet(50, 70, 12, [0.8, 5.2]))._graph_warmup is None
    monkeypatch.delenv('NNL_DEFAULT_GRAPHS')
    plain = make_model_basic(nn.Sequential(nn.Linear(4, 1), nn.Flatten(0)))
    assert Learner('/tmp/nnl_graph_test', Data([batch], 64, 'cont'), plain)._graph_warmup is None


def test_fit_reads_replayed_losses_one_step_late_and_records_the_same_schedule():
    """Learner.fit()'s inner loop launches step i + 1 before it reads the loss of the replayed step i (round 4: the per-step
    `loss.item()` left the GPU idle while the host staged the next minibatch).  The recorded loss schedule, the moving average
    and the trained parameters must equal those of the same steps made one by one with train1minibatch (which still returns
    the float of its own step); ragged last minibatch (eager, then its own graph) and two epochs included."""
    from neuralnetworklibrary_amd.Applications.CollabFiltering import CollabFilterNet
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    g = torch.Generator().manual_seed(31)
    def batch(n):
        return (torch.stack([torch.randint(0, 50, (n,), generator=g), torch.randint(0, 70, (n,), generator=g)], 1).to(DEV),
                torch.randint(1, 6, (n,), generator=g).float().to(DEV))
    batches = [batch(64) for _ in range(7)] + [batch(23)]
    runs = []
    for mode in ('fit', 'steps'):
        torch.manual_seed(0)
        net = CollabFilterNet(50, 70, 12, [0.8, 5.2])
        learner = Learner('/tmp/nnl_graph_fit', Data(batches, 64, 'cont'), net, optimizer='Adam')
        learner.use_graphs(True, warmup=2)
        if mode == 'fit':
            learner.fit(1e-2, 2, wd=1e-3)
            losses, avg = list(learner.loss_sched), learner.moving_avg_loss
        else:
            learner.init_optimizer(wd=1e-3)
            learner.model.train()
            losses, avg = [], 0.0
            for ep in range(2):
                for x, y in batches:
                    losses.append(learner.train1minibatch(x, y, 1e-2))
                    avg = avg * 0.98 + losses[-1] * 0.02
        assert sum(gr.graph is not None for gr in learner._graphs.values()) >= 1
        runs.append((losses, avg, [p.detach().cpu().numpy().copy() for p in net.parameters()]))
    (lf, af, pf), (ls, as_, ps) = runs
    assert len(lf) == len(ls) == 16
    assert_close(np.array(lf), np.array(ls), 1e-6, 1e-7, 'loss schedule')
    assert abs(af - as_) <= 1e-6 * max(abs(as_), 1e-12)
    for a, b in zip(pf, ps):
        assert_close(a, b, 1e-6, 1e-8, 'params')


def test_fit_loop_runs_ahead_of_the_gpu_with_per_step_schedules():
    """The pipelined fit loop stages step i + 1 (inputs, lr / betas / Adam bias corrections) while step i may still be running:
    the replayed optimizer must read EACH step's own values (they travel through a ring of pinned buffers into a device buffer that a
    captured kernel patches into the descriptor table — a single pinned image was overwritten one step early, which the G10
    tabular curve caught).  64 steps with a different lr and betas on every step against the same steps made one by one."""
    from neuralnetworklibrary_amd.Applications.StructuredData import StructuredDataNet
    from neuralnetworklibrary_amd.General.Core import set_default_device
    from neuralnetworklibrary_amd.General.Learner import Learner
    set_default_device(DEV)
    Learner.verbose = False
    cards = [11, 5, 7]
    rs = np.random.RandomState(9)
    def batch(n):
        xcat = torch.from_numpy(np.stack([rs.randint(0, c, size=n) for c in cards], 1).astype(np.int64)).to(DEV)
        xcont = torch.from_numpy(rs.standard_normal((n, 2)).astype(np.float32)).to(DEV)
        return [xcat, xcont], torch.from_numpy((5 + 7 * rs.rand(n)).astype(np.float32)).to(DEV)
    batches = [batch(32) for _ in range(8)]
    n_steps = 64
    lr_sched = [[1e-3 * (1 + 0.2 * (i % 7)), 3e-3 * (1 + 0.1 * (i % 5))] for i in range(n_steps)]
    betas_sched = [(0.9 - 0.005 * (i % 11), 0.99 - 0.001 * (i % 3)) for i in range(n_steps)]
    runs = []
    for mode in ('fit', 'steps'):
        torch.manual_seed(1)
        net = StructuredDataNet('cont', 3, 2, [{i: i for i in range(c)} for c in cards], [24, 12, 1], output_range=[5, 12])
        learner = Learner('/tmp/nnl_graph_fit2', Data(batches, 32, 'cont'), net, optimizer='Adam')
        learner.init_optimizer(wd=1e-3)
        learner.use_graphs(True, warmup=2)
        if mode == 'fit':
            learner.train_gen_sched(lr_sched, None, betas_sched)
            losses = list(learner.loss_sched)
        else:
            learner.model.train()
            losses = [learner.train1minibatch(*batches[i % 8], lr_sched[i], betas_batch=betas_sched[i]) for i in range(n_steps)]
        runs.append((losses, [p.detach().cpu().numpy().copy() for p in net.parameters()]))
    (lf, pf), (ls, ps) = runs
    assert len(lf) == n_steps
    assert_close(np.array(lf), np.array(ls), 1e-6, 1e-7, 'loss schedule')
    for a, b in zip(pf, ps):
        assert_close(a, b, 1e-6, 1e-8, 'params')
