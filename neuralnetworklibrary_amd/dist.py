"""Synchronous data parallelism for Learner.fit(): one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference is single-GPU ("if a machine has multiple GPUs only 1 of them will be utilized", README.md:11-12); this
module is the MI355X-native addition.  Design for the node's topology (8 GPUs, point-to-point xGMI,
7 links/GPU): a ring all-reduce is per-link bound, so gradients are packed into a few LARGE flat buckets (default
25 MB, ResNet-34's 87 MB -> 4 collectives) that RCCL can spread over all links, and each bucket's all-reduce is launched
from a post-accumulate-grad hook as soon as its last gradient is produced, i.e. in reverse layer order while the rest
of backward is still running on the compute stream (RCCL runs on its own stream).  `finish()` (called by
Optimizer.step) waits for the collectives, and `param.grad` becomes a view of the averaged flat bucket (no copy back).

Numerics: every rank computes the mean loss of its local shard; the gradient of the global-batch mean is the
shard-size-weighted mean sum_r (n_r / n) g_r.  With equal shards that is RCCL's ReduceOp.AVG; for a
ragged global minibatch `begin(weight=n_r * world / n)` scales the rank's bucket before the collective.  BatchNorm uses per-replica batch statistics by default (standard
DDP semantics; at the benchmark's 64 images per GPU that is exactly the reference's batch-statistics population);
`enable_sync_bn` / `Learner.distribute(sync_bn=True)` switches to global-batch statistics (SyncBN kernels in
csrc/batchnorm.hip), which reproduces the single-GPU reference on the same GLOBAL minibatch.
"""
import os

import torch
import torch.distributed as dist

__all__ = ['init_from_env', 'GradSync', 'ShardedBatches', 'shard_bounds', 'enable_sync_bn', 'enable_sync_renorm', 'world_size',
           'rank', 'drop_ctx', 'keyed_mask', 'KeyedDropout']


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets
    them).  backend: 'nccl' (= RCCL on ROCm) when a GPU is visible, else 'gloo'.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        use_gpu = torch.cuda.is_available()
        backend = backend or ('nccl' if use_gpu else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if use_gpu:
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rk, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rk, world_size=world)
    return rk, world, local


def shard_bounds(n, rank_, world):
    """Rows [a, z) of a GLOBAL minibatch of n rows that rank `rank_` of `world` owns, and whether that slice is a GHOST.
    Balanced contiguous cut: the first n % world ranks get one row more, so shard sizes differ by at most one and no rank
    is empty while n >= world.  With n < world the ranks beyond n would be empty — but every rank must run the step (the
    gradient, SyncBN and renorm collectives are rank-uniform) — so such a rank re-uses row (rank mod n) as a ghost: it is
    computed like any other row and enters the gradient average with weight 0."""
    base, rem = divmod(n, world)
    a = rank_ * base + min(rank_, rem)
    z = a + base + (1 if rank_ < rem else 0)
    if z == a and n > 0:
        a = rank_ % n
        return a, a + 1, True
    return a, z, False


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        # [gradients | one "this rank produced a gradient" flag per parameter | one status word]: the flags ride in the same collective,
        # so every rank learns which parameters got a gradient ANYWHERE without a second message; the status word of the FIRST bucket
        # carries "a replay-overlap wait kernel timed out on this rank" to all ranks (GradSync.reduce_overlapped)
        self.flat = torch.zeros(self.numel + len(params) + 1, dtype=torch.float32, device=device)
        self.flags = self.flat[self.numel:self.numel + len(params)]
        self.status = self.flat[self.numel + len(params):]
        self.views, o = [], 0
        for p in params:
            # same memory layout as the parameter (conv weights are stored channels_last): copies in and the optimizer's
            # foreach kernels then see matching strides
            dense = p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.is_contiguous()
            seg = self.flat[o:o + p.numel()]
            self.views.append(seg.as_strided(p.shape, p.stride()) if dense else seg.view(p.shape))
            o += p.numel()
        self.pending = len(params)
        self.ready = [False] * len(params)
        self.handle = None
        self.averaged = False


_FORCE_ALLREDUCE = os.environ.get('NNL_DIST_FORCE_ALLREDUCE') == '1'       # read once at import


class GradSync:
    """Bucketed, backward-overlapped gradient all-reduce (mean over ranks) for `model`'s trainable parameters."""

    def __init__(self, model, bucket_mb=25.0, group=None):
        self.model, self.bucket_bytes, self.group = model, int(bucket_mb * 2 ** 20), group
        self._hooks = []
        self.rebuild()

    def rebuild(self):
        """(Re)build buckets and hooks — call after freezing / unfreezing parameters."""
        for h in self._hooks:
            h.remove()
        for b in getattr(self, 'buckets', []):
            for p in b.params:
                if hasattr(p, '_nnl_grad_dst'):
                    del p._nnl_grad_dst
                if hasattr(p, '_nnl_uses'):
                    del p._nnl_uses
        self._hooks, self.buckets, self._where = [], [], {}
        params = [p for p in self.model.parameters() if p.requires_grad]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:                      # tied weights (LM decoder/embedding) appear once
                seen.add(id(p)); uniq.append(p)
        cur, cur_bytes = [], 0
        for p in reversed(uniq):                       # reverse registration order ~ order gradients become ready
            cur.append(p); cur_bytes += p.numel() * 4
            if cur_bytes >= self.bucket_bytes:
                self.buckets.append(_Bucket(cur, p.device)); cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, cur[0].device))
        self._use_counts = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[id(p)] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                # ops._Conv2d.backward writes the weight gradient straight into the bucket (no copy kernel) when it finds this
                # AND the weight was used exactly once in the step's forward (`_nnl_uses[0] == 1`): a weight shared by several
                # calls (RetinaNet's heads run on 5 pyramid levels) gets one gradient per use that autograd must SUM, so each
                # of them needs its own buffer
                p._nnl_grad_dst = b.views[pi]
                p._nnl_uses = [0]
                self._use_counts.append(p._nnl_uses)
        self._active = False
        self.capturing = False                          # Learner.use_graphs under DP: hooks fill the buckets, collectives run after the replay
        # the replay-overlap protocol's device words + side stream: ONE set per GradSync, shared by every captured step (Learner keeps
        # up to four graphs per GradSync, e.g. the full and the ragged last minibatch) — the step word the graphs bump and the host's
        # replay counter are global, so the flags of whichever graph ran last are comparable with the next wait.  Captured kernels
        # hold raw pointers into `words`: the tensor is never replaced while this object lives (a rebuild with more buckets than its
        # capacity retires it into `_retired_overlaps`, which keeps the memory alive for graphs that may still be replayed)
        ov = getattr(self, 'overlap', None)
        if ov is not None and ov.capacity < len(self.buckets):
            self._retired_overlaps = getattr(self, '_retired_overlaps', []) + [ov]
            ov = None
        self.overlap = ov                               # created by prepare_overlap() (GPU only)
        self.overlap_ok = True                          # False after a wait kernel's time-out: later replays reduce after the replay
        self.overlap_launches, self.last_signalled = 0, None   # collectives enqueued behind a wait kernel / signals of the last replayed graph (diagnostics)
        self.weight = 1.0
        self.direct_writes, self.steps = 0, 0           # gradients that arrived in place / backward passes (diagnostics)

    def begin(self, weight=1.0):
        """Arm the hooks for one backward pass (Learner.train1minibatch calls this before the forward).
        weight: this rank's share of the global minibatch relative to an equal split, n_local * world / n_global (1.0 for
        equal shards, 0.0 for a ghost shard): the bucket is scaled by it before the averaging collective."""
        for b in self.buckets:
            b.pending, b.ready, b.handle, b.averaged = len(b.params), [False] * len(b.params), None, False
        for u in self._use_counts:
            u[0] = 0
        self._next = 0                                   # first bucket whose collective has not been issued yet
        self._active = True
        self.weight = float(weight)
        self.steps += 1

    def _launch(self, b):
        if all(b.ready):
            b.flags.fill_(1.0)                           # (one launch; the flags were averaged / scaled by the last collective)
        else:
            b.flags.copy_(torch.tensor([1.0 if r else 0.0 for r in b.ready], dtype=torch.float32))
        if self.weight != 1.0:
            b.flat[:b.numel].mul_(self.weight)
        # NNL_DIST_FORCE_ALLREDUCE=1: issue the collective even at world_size 1 (exercises the RCCL call path on a 1-GPU box)
        if world_size() > 1 or (dist.is_initialized() and _FORCE_ALLREDUCE):
            # RCCL averages inside the collective (ReduceOp.AVG); gloo has no AVG: sum, then finish() scales
            b.averaged = dist.get_backend(self.group) == 'nccl'
            op = dist.ReduceOp.AVG if b.averaged else dist.ReduceOp.SUM
            b.handle = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    def _on_grad(self, p):
        if not self._active:
            return
        bi, pi = self._where[id(p)]
        b = self.buckets[bi]
        if b.ready[pi]:
            return
        if p.grad.data_ptr() == b.views[pi].data_ptr() and p.grad.stride() == b.views[pi].stride():
            self.direct_writes += 1                     # produced in place by the wgrad kernel
        else:
            b.views[pi].copy_(p.grad)
        b.ready[pi] = True
        b.pending -= 1
        if self.capturing:
            # hipGraph capture of forward + backward: no collective is captured.  On the GPU the complete buckets at the head of the
            # queue get their flag words set and a SIGNAL kernel (csrc/runtime.hip: nnl_dp_signal) right here — i.e. at the point of
            # the captured backward where their last gradient has been written — so that reduce_overlapped() can start bucket k's
            # all-reduce in the MIDDLE of each replay (strictly in bucket order, as the eager path)
            if self._capture_overlap:
                while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
                    bk = self.buckets[self._next]
                    bk.flags.fill_(1.0)
                    self.overlap.signal(self._next)
                    self._next += 1
            return
        # collectives are issued strictly in bucket order on every rank (a rank whose data did not reach some parameter must
        # not pair its bucket k with another rank's bucket j): launch the complete buckets at the head of the queue
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def prepare_overlap(self):
        """BEFORE the capture of a data-parallel step (allocations and stream creation must not be captured): the device words and side
        stream of the replay-overlap protocol.  Not available on CPU tensors (the gloo tests) or with NNL_DIST_REPLAY_OVERLAP=0: then
        reduce_all() is the replay path."""
        dev = self.buckets[0].flat.device if self.buckets else None
        self._capture_overlap = not (dev is None or dev.type != 'cuda' or os.environ.get('NNL_DIST_REPLAY_OVERLAP', '1') == '0')
        if self._capture_overlap and self.overlap is None:
            self.overlap = _ReplayOverlap(len(self.buckets), dev)
        self._next = 0

    _capture_overlap = False

    def capture_begin(self):
        "first thing INSIDE the capture (Learner._GraphedStep): the replay counter's bump"
        if self._capture_overlap:
            self.overlap.bump()

    def capture_end(self):
        """-> how many buckets got a signal in the captured backward (the others — parameters without a gradient — are reduced after the
        replay), or None when the capture carries no overlap protocol.  The captured step keeps the number and hands it to
        reduce_overlapped(): it belongs to THAT graph, not to the GradSync."""
        signalled = self._next if self._capture_overlap else None
        self._next, self._capture_overlap = 0, False
        return signalled

    def reduce_overlapped(self, weight=1.0, signalled=None):
        """After a captured forward + backward has been ENQUEUED (graph.replay() returned): for every bucket, in order, a wait kernel on the
        side stream (until the replay's signal for that bucket) followed by its all-reduce — the collectives of the early buckets run
        under the rest of the replayed backward; buckets that completed without a signal wait for the whole replay.  Then finish().
        signalled: capture_end()'s value for the graph that was just replayed."""
        ov = self.overlap
        if ov is None or signalled is None:
            return self.reduce_all(weight)
        ov.replays += 1                                   # the graph bumped the device step word: keep the host's count in step ALWAYS
        if not self.overlap_ok:
            # after a time-out on this rank: the round-3 behaviour (collectives behind the whole replay) — the sticky error word still
            # travels in the first bucket, so that every rank raises at this step
            self.buckets[0].status.copy_(ov.err)
            self.reduce_all(weight)
            ov.stage_err(self.buckets[0].status)
            return
        self._active, self.weight, self._next = True, float(weight), 0
        self.last_signalled = signalled
        main = torch.cuda.current_stream()
        with torch.cuda.stream(ov.side):
            # a time-out seen by an EARLIER step's wait kernels (the word is sticky) rides in the first bucket's status word: one step
            # later every rank knows that some rank stepped on incomplete gradients, and all of them raise at the same step
            self.buckets[0].status.copy_(ov.err)
        for k, b in enumerate(self.buckets):
            b.pending, b.ready, b.handle, b.averaged = 0, [True] * len(b.params), None, False
            if k < signalled:
                ov.wait(k)                                # enqueued on ov.side
            else:
                ov.side.wait_stream(main)                 # no signal in the capture: after the whole replay
            with torch.cuda.stream(ov.side):
                if k >= signalled:
                    b.flags.fill_(1.0)
                self._launch_filled(b)
            self.overlap_launches += int(k < signalled)
        self._next = len(self.buckets)
        main.wait_stream(ov.side)                         # (the averaged buckets are consumed on the main stream)
        self.finish()
        ov.stage_err(self.buckets[0].status)              # async copies behind this step's work: this rank's error word, all ranks' status

    def _launch_filled(self, b):
        "the collective of a bucket whose flag words are already set (replay path)"
        if self.weight != 1.0:
            b.flat[:b.numel].mul_(self.weight)
        if world_size() > 1 or (dist.is_initialized() and _FORCE_ALLREDUCE):
            b.averaged = dist.get_backend(self.group) == 'nccl'
            op = dist.ReduceOp.AVG if b.averaged else dist.ReduceOp.SUM
            b.handle = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    def raise_if_overlap_error(self, synced=False):
        """A wait kernel of the replay path ran into its time bound (the signal never came): the bucket's all-reduce then ran on
        partially written gradients and the optimizer stepped on them.  Called right after the per-step `loss.item()` of a replayed
        data-parallel step (synced=True: the pinned copies were enqueued before that sync, no second one) and at epoch end.
        This rank switches to collectives-after-the-replay at once (no wait kernels; the collective sequence is unchanged, so that
        needs no agreement).  The RAISE is rank-uniform: a lone rank raises at the step itself; with several ranks the sticky error
        word travels in the next step's first bucket (status word) and EVERY rank raises after that step's loss read — no rank is
        left alone in a collective.  At epoch end the decision is made with one MAX all-reduce.  The words are reset on raising."""
        ov = self.overlap
        if ov is None:
            return
        w = world_size()
        if synced:
            local, anywhere = int(ov.host_err[0]) != 0, float(ov.host_status[0]) > 0.0
        else:
            local = int(ov.err.item()) != 0
            anywhere = local
            if w > 1:
                t = torch.tensor([1.0 if local else 0.0], device=ov.err.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                anywhere = float(t.item()) > 0.0
        if local:
            self.overlap_ok = False
        if anywhere or (local and w == 1):
            ov.err.zero_()
            ov.host_err.zero_(); ov.host_status.zero_()
            self.buckets[0].status.zero_()
            self.overlap_ok = False
            raise RuntimeError('data-parallel replay: a bucket wait kernel timed out on %s (no signal from the captured backward within '
                               '%.0f s); a step used incomplete gradients' % ('this rank' if local else 'another rank', ov.timeout_us * 1e-6))

    def reduce_all(self, weight=1.0):
        """After the replay of a captured forward + backward (which filled every bucket): all-reduce all buckets now, in order."""
        self._active, self.weight, self._next = True, float(weight), 0
        for b in self.buckets:
            b.pending, b.ready, b.handle, b.averaged = 0, [True] * len(b.params), None, False
        self.finish()

    def finish(self):
        """Wait for every bucket, average, and point param.grad at the averaged bucket views.
        Every bucket is reduced on every rank every step (the collective sequence never depends on which parameters a rank's
        data happened to reach).  A parameter that received no gradient on ANY rank keeps `grad = None`, so the optimizer skips
        it exactly as the reference's does (torch.optim ignores grad-less parameters); one that received a gradient only on
        other ranks gets their average — the rank learns which case it is from the flag words reduced with the bucket."""
        if not self._active:
            return
        self._active = False
        w = world_size()
        for b in self.buckets[self._next:]:
            if b.pending > 0:                          # parameters that received no gradient on this rank this step
                for pi, p in enumerate(b.params):
                    if not b.ready[pi]:
                        if p.grad is None:
                            b.views[pi].zero_()
                        else:                          # (a gradient that arrived outside the hooks)
                            b.views[pi].copy_(p.grad)
                            b.ready[pi] = True
            self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
            if w > 1 and not b.averaged:
                b.flat.mul_(1.0 / w)
            missing = [pi for pi in range(len(b.params)) if not b.ready[pi]]
            anywhere = b.flags.tolist() if missing else None        # host read only in the rare grad-less case
            for pi, (p, v) in enumerate(zip(b.params, b.views)):
                if b.ready[pi] or anywhere[pi] > 0.0:
                    p.grad = v
                else:
                    p.grad = None


class _ReplayOverlap:
    """Device words and the side stream of the replay-overlap protocol (csrc/runtime.hip: nnl_dp_bump / _signal / _wait): step[0] counts
    replays (bumped by the graph itself), flag[k] = the replay that has completed bucket k, err != 0 after a wait kernel's time-out."""

    def __init__(self, n_buckets, device):
        from . import _lib
        self._lib = _lib
        self.capacity = max(int(n_buckets), 64)
        self.words = torch.zeros(self.capacity + 2, dtype=torch.int32, device=device)  # [step, err, flag_0 ... flag_{n-1}]
        self.step, self.err, self.flags = self.words[0:1], self.words[1:2], self.words[2:]
        self.side = torch.cuda.Stream(device=device)
        self.replays = 0                                                               # host mirror of the device step word
        self.timeout_us = int(float(os.environ.get('NNL_DIST_WAIT_SECONDS', 60)) * 1e6)   # wall time (device clock), not polls
        self.host_err = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.host_status = torch.zeros(1, dtype=torch.float32).pin_memory()

    def stage_err(self, status):
        self.host_err.copy_(self.err, non_blocking=True)
        self.host_status.copy_(status, non_blocking=True)

    def _s(self):
        return torch.cuda.current_stream().cuda_stream

    def bump(self):
        self._lib.check(self._lib.lib.nnl_dp_bump(self.step.data_ptr(), self._s()))

    def signal(self, k):
        self._lib.check(self._lib.lib.nnl_dp_signal(self.flags[k:k + 1].data_ptr(), self.step.data_ptr(), self._s()))

    def wait(self, k):
        self._lib.check(self._lib.lib.nnl_dp_wait(self.flags[k:k + 1].data_ptr(), self.replays, self.timeout_us, self.err.data_ptr(), self.side.cuda_stream))


class ShardedBatches:
    """Wrap an iterable of (x, y) GLOBAL minibatches: each rank gets its contiguous slice along dim 0 (samples; for the
    language model dim 0 is the stream dimension, so a rank keeps the same streams — and their carried hidden state —
    from batch to batch, Text.py:254-263).  len() is unchanged, so schedules are identical on every rank."""

    def __init__(self, batches, rank_=None, world=None):
        self.batches = batches
        self.rank = rank() if rank_ is None else rank_
        self.world = world_size() if world is None else world
        self.dp_info = None          # (rows of the last yielded shard that count, rows of its GLOBAL minibatch): Learner reads it

    def __len__(self):
        return len(self.batches)

    def _cut(self, t, a, z):
        if isinstance(t, (list, tuple)):
            return [self._cut(v, a, z) for v in t]
        return t[a:z]

    @staticmethod
    def _rows(t):
        while isinstance(t, (list, tuple)):
            t = t[0]
        return t.shape[0]

    def __iter__(self):
        for x, y in self.batches:
            n = self._rows(y)
            a, z, ghost = shard_bounds(n, self.rank, self.world)
            self.dp_info = (0 if ghost else z - a, n)
            yield self._cut(x, a, z), self._cut(y, a, z)


def enable_sync_bn(model, group=None, comm=None):
    """Mark every BatchNorm module of `model` for cross-replica statistics: in training mode ops.bn_act then
    normalises with the mean / variance of the GLOBAL batch (all ranks' rows) — what the single-GPU reference computes on
    the same global minibatch — at the cost of one small all_gather (forward) and one all_reduce (backward) per BN layer.
    A no-op at world size 1 unless `comm` is given (tests inject a fake communicator)."""
    from .ops import DistComm
    if comm is None and world_size() == 1:
        return model
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.nnl_sync = (group, comm or DistComm)
    return model


def enable_sync_renorm(model, capacity, group=None, comm=None):
    """Data-parallel StructuredDataNet: `nn.Embedding(max_norm=1.5)` renormalises the looked-up rows IN PLACE during the
    forward (reference General/Layers.py:56-76), so ranks that look up different rows would let the replicated tables drift
    apart.  Marks the tabular front end to all-gather the step's indices ([capacity, n_cat] int64 per rank; capacity = the
    per-rank full batch size) and renormalise the union (ops.tab_embed_concat).  No-op at world size 1 unless `comm`."""
    from .ops import DistComm
    if comm is None and world_size() == 1:
        return model
    for m in model.modules():
        if hasattr(m, 'embeddings') and hasattr(m, '_plan'):
            m.nnl_dp = (group, comm or DistComm, int(capacity))
    return model


# ---- dropout masks keyed by (seed, step, call, GLOBAL sample index) ------------------------------------------------------
# The reference draws every dropout mask from torch's global RNG stream (General/Layers.py:74-76, Text.py:443-475, nn.Dropout in
# the heads).  Under data parallelism each rank owns a different stream, so (a) the per-sample masks of a rank's shard are not
# the masks the single-process run gives those samples and (b) masks over PARAMETERS (the LSTM weight drop [4H, H], the
# vocabulary-row mask [V, 1]) differ between ranks — every replica then differentiates a different network.  With
# `Learner.use_keyed_dropout(seed)` every mask element is a pure function of (seed, training step, index of the mask request
# within the step, element index), the element index being taken in the GLOBAL minibatch for per-sample masks: an N-rank run with
# dropout ON reproduces the 1-rank run on the same global minibatches.  Integer hashing only
# (splitmix64 finaliser in wrapping int64 arithmetic): bit-identical on every device and torch version.
class _DropCtx:
    def __init__(self):
        self.enabled, self.seed, self.step, self.calls, self.offset, self.global_rows = False, 0, 0, 0, 0, None

    def begin_step(self, step, offset=0, global_rows=None):
        "called by Learner.train1minibatch: `offset` = index of this rank's first row in the global minibatch of `global_rows` rows"
        self.step, self.calls, self.offset, self.global_rows = int(step), 0, int(offset), global_rows


drop_ctx = _DropCtx()


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _mix(x):
    "splitmix64 finaliser on an int64 tensor (logical shifts emulated by masking the sign extension)"
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * _s64(0x94D049BB133111EB)
    return x ^ ((x >> 31) & ((1 << 33) - 1))


def keyed_mask(shape, p, device, sample_dim=None):
    """Bernoulli(1 - p) / (1 - p) mask of `shape` under the keyed scheme, or None when it is not enabled (the caller then draws
    from torch's stream as the reference does).  sample_dim: the dimension that indexes the samples of the minibatch (None: a
    parameter-shaped mask, identical on every rank)."""
    c = drop_ctx
    if not c.enabled:
        return None
    c.calls += 1
    shape = tuple(int(v) for v in shape)
    gshape = list(shape)
    idx = torch.zeros(shape, dtype=torch.int64, device=device)
    if sample_dim is not None and c.global_rows is not None:
        gshape[sample_dim] = max(int(c.global_rows), shape[sample_dim] + c.offset)
    stride = 1
    for d in reversed(range(len(shape))):
        ar = torch.arange(shape[d], dtype=torch.int64, device=device)
        if d == sample_dim:
            ar = ar + c.offset
        idx = idx + (ar * stride).view([-1 if k == d else 1 for k in range(len(shape))])
        stride *= gshape[d]
    key = _s64(c.seed * 0x9E3779B97F4A7C15 + c.step * 0xD1B54A32D192ED03 + c.calls * 0x8CB92BA72F3D8DD7)
    h = _mix(_mix(idx + key) + _s64(0x9E3779B97F4A7C15))
    u = ((h >> 40) & 0xFFFFFF).to(torch.float32) * (1.0 / (1 << 24))
    return (u >= p).to(torch.float32) * (1.0 / (1.0 - p)) if p < 1 else torch.zeros(shape, device=device)


class KeyedDropout(torch.nn.Dropout):
    """nn.Dropout whose mask comes from `keyed_mask` (rows = samples) when the keyed scheme is on; exactly nn.Dropout otherwise.
    sample_dim: which dimension of the input indexes the samples (None: parameter-shaped input, e.g. the LSTM weight drop)."""

    def __init__(self, p=0.5, sample_dim=0):
        super().__init__(p)
        self.sample_dim = sample_dim

    def forward(self, x):
        if self.training and self.p > 0 and drop_ctx.enabled:
            return x * keyed_mask(x.shape, self.p, x.device, self.sample_dim)
        return super().forward(x)
