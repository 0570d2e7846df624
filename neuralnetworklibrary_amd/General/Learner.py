"""`Learner` — the training / evaluation loop of the drop-in API.

Mirror of the reference's General/Learner.py (class Learner :64-887): same constructor, same public
methods and keyword arguments, same numerical behaviour of the loop, including its quirks:
  * a full validation pass before epoch 0 sets `min_loss` (Learner.py:566);
  * last-batch learning-rate scaling `lr * bs / data.bs` (Learner.py:503-505);
  * EMA of the minibatch loss with constants .98/.02 and debiasing `/(1-.98^(i+1))` (Learner.py:610-611);
  * `model.reset()` is never called between train and val (Learner.py:346,417,587,839);
  * early stop when val_loss > 20*min_loss (Learner.py:672-675); find_lr break_fac (Learner.py:866);
  * `get_sched` converts only `start_val` from list to array (the `end_value` typo, Learner.py:716).
What is new (MI355X-first, not in the reference):
  * device-agnostic placement through Core.default_device() instead of hard `.cuda()` (Learner.py:107);
  * one-process-per-GPU data parallelism: when torch.distributed is initialised, gradients are
    all-reduced over RCCL by `neuralnetworklibrary_amd.dist.GradSync` (attached to the Optimizer), the
    last-batch lr scale uses the GLOBAL batch size, evaluation sums are all-reduced, and only rank 0
    prints / saves;
  * plotting is imported lazily (matplotlib is UI, out of scope for the hot path).
"""
import copy
import os
import time
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from .Core import (ARR, bn_types, combine_models, correct_foldername, default_device, linear_space, list_mult,
                   outer_mult, to_cuda)
from .LossesMetrics import AUC
from .Optimizer import Optimizer, get_param_dict

__all__ = ['Learner', 'HipMSELoss', 'HipCrossEntropyLoss', 'end_metrics', 'SGD_Mom', 'Adam2', 'opt_dict', 'loss_func_dict', 'plot_confusion_matrix']

# registries (General/Learner.py:16-21)
end_metrics = {'auc': AUC}
SGD_Mom = partial(optim.SGD, momentum=0.9)
Adam2 = partial(optim.Adam, betas=(0.9, 0.99))
opt_dict = {'default': SGD_Mom, 'SGD_Mom': SGD_Mom, 'SGD': optim.SGD, 'Adam': optim.Adam, 'Adam2': Adam2}
class HipCrossEntropyLoss(nn.CrossEntropyLoss):
    """nn.CrossEntropyLoss() (the default loss of the 'cat' / 'single_label' target types, reference General/Learner.py:20)
    on the online-softmax HIP kernels (csrc/text.hip: one pass for max and sum-exp, one for the gradient) when the inputs are
    CUDA class-index targets with the default options; anything else (CPU tensors in host-logic tests, class weights, label
    smoothing, probability targets) takes torch's implementation."""

    def forward(self, input, target):
        plain = (self.weight is None and self.reduction == 'mean' and self.ignore_index == -100 and self.label_smoothing == 0.0)
        if plain and input.is_cuda and input.dim() >= 2 and target.dtype == torch.long and target.dim() == input.dim() - 1:
            from ..ops_text import cross_entropy_nd
            return cross_entropy_nd(input, target)
        return super().forward(input, target)


class HipMSELoss(nn.MSELoss):
    """nn.MSELoss() (the default loss of the 'cont' target type, reference General/Learner.py:20) on the HIP kernels of csrc/loss.hip
    (one launch forward, one backward — the collaborative-filtering / structured-data steps are launch-bound) when prediction
    and target are same-shape CUDA tensors, the target needs no gradient and the reduction is 'mean'; torch's otherwise."""

    def forward(self, input, target):
        if (self.reduction == 'mean' and input.is_cuda and target.is_cuda and input.shape == target.shape and input.numel() > 0
                and not target.requires_grad and input.dtype == torch.float32):
            from ..ops import mse_loss
            return mse_loss(input, target)
        return super().forward(input, target)


loss_func_dict = {'cont': HipMSELoss(), 'cat': HipCrossEntropyLoss(), 'single_label': HipCrossEntropyLoss(),
                  'multi_label': nn.BCEWithLogitsLoss()}


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _rank():
    d = _dist()
    return d.get_rank() if d else 0


def _batch_size(y_batch):
    return len(y_batch) if type(y_batch) != list else len(y_batch[0])


def _raise_if_index_error(local=False):
    """The gather kernels (embeddings, tabular front end, softmax-CE targets) skip an out-of-range index and raise a device
    flag instead of faulting; the reference's nn.Embedding / CrossEntropyLoss would have raised.  Read the flag once per
    epoch / evaluate / predict (one D2H copy) and raise on EVERY rank of a data-parallel job (MAX over the ranks of the two
    flag words: index out of range -> IndexError, persistent-LSTM grid-barrier time-out -> NnlError; the same mapping as
    ops.raise_if_index_error).  local=True: no collective (predict())."""
    if default_device().type != 'cuda':
        return
    from .. import ops
    d = _dist()
    if d is None or local:
        ops.raise_if_index_error()
        return
    flag = ops._flags(default_device())                 # [index word, LSTM time-out word]
    total = flag.clone()
    d.all_reduce(total, op=d.ReduceOp.MAX)
    v = total.tolist()
    code = (1 if v[0] else 0) | (2 if v[1] else 0)
    if code != 0:
        flag.zero_()
        ops.raise_for_flag(code)


def _own_rows(dl, y_batch):
    "rows of this shard that count: a sharding loader's ghost shard (dist.shard_bounds) counts 0"
    info = getattr(dl, 'dp_info', None)
    return info[0] if info is not None else _batch_size(y_batch)


def plot_confusion_matrix(cm, classes, normalize=False, title='Confusion matrix', cmap=None):
    "Plot a confusion matrix (General/Learner.py:32-60; UI helper)."
    import itertools
    import matplotlib.pyplot as plt
    if normalize:
        cm = cm.astype('float') / cm.sum(axis=1)[:, np.newaxis]
    print(cm)
    plt.imshow(cm, interpolation='nearest', cmap=cmap or plt.cm.Blues)
    plt.title(title)
    plt.colorbar()
    ticks = np.arange(len(classes))
    plt.xticks(ticks, classes, rotation=45)
    plt.yticks(ticks, classes)
    fmt = '.2f' if normalize else 'd'
    thresh = cm.max() / 2.
    for i, j in itertools.product(range(cm.shape[0]), range(cm.shape[1])):
        plt.text(j, i, format(cm[i, j], fmt), horizontalalignment="center",
                 color="white" if cm[i, j] > thresh else "black")
    plt.ylabel('True label')
    plt.xlabel('Predicted label')
    plt.tight_layout()


def _tensor_leaves(b):
    "flat list of the tensors of a (possibly nested list) batch; None marks a non-tensor leaf"
    if torch.is_tensor(b):
        return [b]
    if isinstance(b, (list, tuple)):
        return [t for v in b for t in _tensor_leaves(v)]
    return [None]


def _tree_map(f, b):
    return f(b) if torch.is_tensor(b) else [_tree_map(f, v) for v in b]


class _PendingLoss:
    """The loss of a replayed step whose device->host copy is in flight (Learner.fit's inner loop reads it AFTER it has launched the
    next step: a replayed tabular / collab step is 40-310 us and the per-step `loss.item()` of the reference, General/Learner.py:516,
    left the GPU idle for 25-50 us while the host staged the next minibatch)."""
    __slots__ = ('host', 'event', 'slot')

    def __init__(self, host, event, slot):
        self.host, self.event, self.slot = host, event, slot

    def result(self):
        self.event.synchronize()
        return float(self.host[self.slot])


class _GraphedStep:
    """One captured training step for one input signature (Learner.use_graphs)."""

    def __init__(self, warmup):
        self.left, self.graph, self.x, self.y, self.loss = max(int(warmup), 1), None, None, None, None

    def run(self, learner, x_batch, y_batch, defer=False):
        """Non-DP: the whole step (forward, loss, backward, fused optimizer) is one graph.  Data parallel: the graph holds forward,
        loss and backward — the gradient hooks' copies into the all-reduce buckets are captured with it — and each replay is
        followed, eagerly, by the bucket all-reduces (RCCL is not captured) and the fused optimizer launch.  The collectives then
        used to start after the whole backward; since round 4 the captured backward carries one signal kernel per bucket and the
        collectives start behind per-bucket wait kernels on a side stream, i.e. in the middle of the replay (dist.py: reduce_overlapped)."""
        opt, gs = learner.optimizer, learner.grad_sync
        dp = gs is not None
        if self.graph is None:
            if self.left > 0:                         # eager steps first: lazy state (optimizer moments, gather plans,
                self.left -= 1                        # workspaces) must exist before the capture
                return None
            dev = default_device()
            self.x = _tree_map(lambda t: t.detach().to(dev, copy=True), x_batch)
            self.y = _tree_map(lambda t: t.detach().to(dev, copy=True), y_batch)
            if isinstance(x_batch, tuple):
                self.x = tuple(self.x)
            opt.opt.zero_grad()
            if dp:
                for m in learner._dp_prepare_mods:    # collectives on the step's INPUTS (the tabular renorm sync's index all-gather):
                    m.nnl_dp_prepare(self.x)          # eagerly, before the capture — the captured forward reads their static result
                gs.begin(1.0)
                gs.capturing = True                   # hooks fill the buckets (+ per-bucket signal kernels) but launch no collective
                gs.prepare_overlap()                  # device words + side stream of the replay overlap (allocated OUTSIDE the capture)
            else:
                opt.prepare_capture()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                # data parallel: RCCL's watchdog thread polls the events of earlier collectives from ANOTHER thread; in the default
                # "global" capture mode that invalidates the capture ("operation not permitted when stream is capturing")
                with torch.cuda.graph(graph, capture_error_mode='thread_local' if dp else 'global'):   # records; nothing executes until replay()
                    if dp:
                        gs.capture_begin()            # the replay counter's bump: the first node of the graph
                    if dev.type == 'cuda':
                        from .. import ops
                        ops.prepare_forward(learner.model)   # all Winograd filters in ONE captured launch (the eager steps built the batch;
                                                             # round 4 captured one filter-transform launch per convolution instead)
                    y_pred = learner.predict1minibatch(self.x)
                    self.loss = learner.loss_func(y_pred, self.y)
                    learner._backward(self.loss)
                    if not dp:
                        opt.step()
            finally:
                if dp:
                    self.signalled = gs.capture_end()    # buckets signalled inside THIS graph (None: no overlap protocol captured)
                    gs.capturing, gs._active = False, False
            self.graph = graph
            self.opt_capture = None if dp else opt.captured()
            if not dp:
                opt.stage_captured()                     # the captured step's own lr / decay / hyper values, before its first replay
        else:
            for dst, src in zip(_tensor_leaves(self.x) + _tensor_leaves(self.y), _tensor_leaves(x_batch) + _tensor_leaves(y_batch)):
                dst.copy_(src, non_blocking=True)
            if not dp:
                opt.replay_step(self.opt_capture)
            else:
                for m in learner._dp_prepare_mods:    # (after the new batch has reached the static inputs, before the replay)
                    m.nnl_dp_prepare(self.x)
        self.graph.replay()
        if defer and not dp:
            # fit()'s pipelined loop: start the scalar's copy into one of two pinned slots and hand back a handle; the static loss
            # tensor is overwritten by the NEXT replay, which is ordered behind this copy on the stream
            if getattr(self, '_host', None) is None:
                self._host = torch.empty(2, dtype=torch.float32).pin_memory()
                self._host_views = [self._host[0:1], self._host[1:2]]
                self._events = [torch.cuda.Event(), torch.cuda.Event()]
                self._loss_src = self.loss.detach().reshape(1) if self.loss.dtype == torch.float32 else None     # a VIEW of the graph's static loss tensor
                self._slot = 0
                self._stream = torch.cuda.current_stream()                     # (Event.record() without a stream costs 10 us of Python)
            self._slot ^= 1
            self._host_views[self._slot].copy_(self._loss_src if self._loss_src is not None else self.loss.detach().reshape(1).float(), non_blocking=True)
            self._events[self._slot].record(self._stream)
            return _PendingLoss(self._host, self._events[self._slot], self._slot)
        if dp:
            # the replay is only ENQUEUED here: the per-bucket wait kernels + all-reduces go to a side stream now and run under the rest of
            # the replayed backward (dist.GradSync.reduce_overlapped; CPU / NNL_DIST_REPLAY_OVERLAP=0: all buckets after the replay)
            gs.reduce_overlapped(learner._dp_weight, getattr(self, 'signalled', None))
            opt.step()
            loss = self.loss.item()                      # (the step's one host sync; the overlap error word's copy was enqueued before it)
            gs.raise_if_overlap_error(synced=True)
            return loss
        return self.loss.item()


class Learner(object):
    """Model + data + optimizer + loss, with fit / fit_cycles / fit_one_cycle / find_lr / evaluate / predict
    (General/Learner.py:64-115 for the arguments and attributes)."""

    verbose = True   # print epoch tables (rank 0 only)

    def __init__(self, PATH, data, model, optimizer='default', loss_func='default', use_moving_avg=True):
        PATH = correct_foldername(PATH)
        os.makedirs(PATH + 'models', exist_ok=True)
        self.PATH, self.data, self.model = PATH, data, model.to(default_device())
        self.loss_sched, self.lr_sched, self.mom_sched, self.betas_sched = [], [], [], []
        self.moving_avg_loss, self.use_moving_avg = 0, use_moving_avg
        self.target_type = data.target_type
        self.loss_func = loss_func_dict[self.target_type] if loss_func == 'default' else loss_func
        self.optimizer = Optimizer(opt_dict[optimizer], self.model) if isinstance(optimizer, str) else optimizer
        self.bn_frozen = None
        self.grad_sync = None
        self._graph_warmup, self._graphs, self._graph_stateless = None, {}, None
        self._loss_host, self._loss_event = None, None
        self._dp_weight, self._dp_equal_shards, self._dp_prepare_mods = 1.0, False, []
        # launch-bound heads (CollabFilterNet, StructuredDataNet: ~100 tiny launches per step, 2.2 - 2.4x slower eager than replayed)
        # mark themselves `nnl_default_graphs`: whole-step hipGraph replay is then ON by default on the GPU, exactly as if the notebook
        # had called learner.use_graphs() — every legality check of _graphed_step still applies per step (tensor batches, training
        # mode, fused optimizer, no keyed dropout, no collectives inside the forward); learner.use_graphs(False) or
        # NNL_DEFAULT_GRAPHS=0 switches it off.
        if (getattr(self.model, 'nnl_default_graphs', False) and default_device().type == 'cuda'
                and os.environ.get('NNL_DEFAULT_GRAPHS', '1') != '0'):
            self.use_graphs(True)

    # ---- data parallelism (new) -----------------------------------------------------
    def distribute(self, bucket_mb=25.0, sync_bn=False, equal_shards=False):
        """Make this learner one rank of a synchronous data-parallel job (one process per GPU).
        equal_shards=True is a promise that every minibatch gives every rank the same number of rows (drop_last loaders,
        synthetic benchmarks): it skips the per-step batch-size agreement described at `_dp_batch_sizes`."""
        from .. import dist as nnl_dist
        self._dp_equal_shards = bool(equal_shards)
        d = _dist()
        if d is not None and d.get_world_size() > 1:
            # replicas start identical: rank 0's parameters and buffers (BN running statistics) everywhere
            with torch.no_grad():
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    t = t.data
                    if t.is_contiguous():
                        d.broadcast(t, 0)
                    elif t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
                        d.broadcast(t.permute(0, 2, 3, 1), 0)          # conv filters are stored KRSC: a dense view
                    else:
                        c = t.contiguous()
                        d.broadcast(c, 0)
                        t.copy_(c)
        self.grad_sync = nnl_dist.GradSync(self.model, bucket_mb=bucket_mb)
        self.optimizer.attach_grad_sync(self.grad_sync)
        if sync_bn:
            nnl_dist.enable_sync_bn(self.model)
        nnl_dist.enable_sync_renorm(self.model, capacity=self.data.bs)     # tabular max_norm renorm over all ranks' lookups
        return self

    def use_keyed_dropout(self, seed=0, flag=True):
        """MI355X addition: draw every dropout mask of the product layers — EmbeddingDrop row masks, the
        tabular continuous mask, nn.Dropout in the heads, LockedDropout, the vocabulary-row mask and the LSTM weight drop — as a
        pure function of (seed, training step, request index, element index in the GLOBAL minibatch) instead of torch's per-process
        RNG stream, so that an N-rank data-parallel run with dropout ON reproduces the 1-rank run (see dist.keyed_mask).  Off by
        default: the reference's masks come from torch's stream."""
        from .. import dist as nnl_dist
        nnl_dist.drop_ctx.enabled, nnl_dist.drop_ctx.seed = bool(flag), int(seed)
        self._kd_step = 0
        return self

    def use_graphs(self, flag=True, warmup=2):
        """MI355X addition: capture the WHOLE training step (forward, loss, backward, fused optimizer) in a hipGraph after
        `warmup` eager steps per input shape and replay it afterwards — for the launch-bound heads (collaborative
        filtering, structured data: ~100 tiny launches per step) the step time is the host's launch rate otherwise.
        Learning rate / weight decay / momentum / betas / Adam bias corrections are read from device memory by the fused
        optimizer kernel, so schedules keep working.  Only for models whose forward is stateless between minibatches
        (NOT the language model: its carried hidden state is Python-side — modules marked `nnl_stateful_forward` keep the step
        eager) and tensor-valued targets; a minibatch of another shape (the ragged last one) runs eagerly.  Invalidated by
        freeze / unfreeze / load."""
        self._graph_warmup, self._graphs, self._dp_graph_ok, self._graph_stateless = (int(warmup) if flag else None), {}, None, None
        return self

    def _reattach(self):
        self._graphs = {}
        if self.grad_sync is not None:
            self.grad_sync.rebuild()
            self.optimizer.attach_grad_sync(self.grad_sync)

    def _dp_batch_sizes(self, bs):
        """(rows of this rank's shard that count, rows of the GLOBAL minibatch, rows of a full global minibatch, world) under data
        parallelism.  The decision is RANK-UNIFORM — every rank takes the same branch for the same minibatch, whatever its own
        shard looks like (a ragged global batch of 7 over 2 ranks is 4 + 3: only one of them is short):
          * the sharding loaders (dist.ShardedBatches, device_data.DeviceBatches) publish `dp_info` = (local, global) with each
            shard, no communication needed;
          * `distribute(equal_shards=True)`: global = local x world by promise;
          * anything else: one small all-reduce of the local row count per step (a host sync, but always correct)."""
        d = _dist()
        world = d.get_world_size() if d else 1
        full = self.data.bs * world
        info = getattr(self.data.train_dl, 'dp_info', None)
        if info is not None:
            return info[0], info[1], full, world
        if d is None or getattr(self, '_dp_equal_shards', False):
            return bs, bs * world, full, world
        t = torch.tensor([float(bs)], device=default_device())
        d.all_reduce(t)
        return bs, int(t.item()), full, world

    # ---- (1) save / load (General/Learner.py:119-153) ---------------------------------------------
    def save(self, filename, save_optimizer=False):
        if _rank() != 0:
            return
        state = {'model_state': self.model.state_dict()}
        if save_optimizer:
            state['optimizer_state'] = self.optimizer.opt.state_dict()
        torch.save(state, self.PATH + 'models/' + filename + '.pt')

    def load(self, filename, saved_optimizer=False):
        path = self.PATH + 'models/' + filename + '.pt'
        if os.path.isfile(path):
            state = torch.load(path, map_location=default_device())
            self.model.load_state_dict(state['model_state'])
            if saved_optimizer:
                self.optimizer.opt.load_state_dict(state['optimizer_state'])
                self._graphs = {}                     # the optimizer state tensors were replaced
        else:
            print("no file found at '{}'".format(path))

    # ---- (2) plotting (General/Learner.py:158-228; UI, lazily imports matplotlib) -------------------
    @staticmethod
    def smooth_timeseries(s, r):
        """Centered moving average of radius r with shrinking windows at both ends (Learner.py:159-184)."""
        N = len(s)
        out = np.zeros(N)
        for i in range(r):
            out[i] = sum(s[0:2 * i + 1]) / (2 * i + 1)
            out[N - 1 - i] = sum(s[N - 1 - 2 * i:N + 1]) / (2 * i + 1)
        for i in range(r, N - r):
            out[i] = sum(s[i - r:i + r + 1]) / (2 * r + 1)
        return list(out)

    def _default_radius(self):
        return max(5, int(len(self.data.train_dl) / 50))

    def plot_loss_sched(self, smoothing_radius='default'):
        import matplotlib.pyplot as plt
        r = self._default_radius() if smoothing_radius == 'default' else smoothing_radius
        plt.plot(self.smooth_timeseries(self.loss_sched, r))
        plt.xlabel('minibatch'); plt.ylabel('train loss')

    def plot_lr_sched(self):
        import matplotlib.pyplot as plt
        plt.plot(self.lr_sched); plt.xlabel('minibatch'); plt.ylabel('learning rate')

    def plot_mom_sched(self):
        import matplotlib.pyplot as plt
        plt.plot(self.mom_sched); plt.xlabel('minibatch'); plt.ylabel('momentum')

    def plot_beta_sched(self):
        import matplotlib.pyplot as plt
        plt.plot([b1 for (b1, b2) in self.betas_sched]); plt.xlabel('minibatch'); plt.ylabel('beta_1')

    def plot_lr_and_loss_sched(self, smoothing_radius='default'):
        import matplotlib.pyplot as plt
        r = self._default_radius() if smoothing_radius == 'default' else smoothing_radius
        fig = plt.figure(figsize=(12, 6))
        sp = fig.add_subplot(1, 2, 1); plt.plot(self.lr_sched); sp.set(xlabel='minibatch', ylabel='learning rate')
        sp = fig.add_subplot(1, 2, 2); plt.plot(self.smooth_timeseries(self.loss_sched, r))
        sp.set(xlabel='minibatch', ylabel='loss')

    # ---- (3) freezing (General/Learner.py:237-272): each call rebuilds the Optimizer ----------------
    def _new_optimizer(self):
        self.optimizer = Optimizer(self.optimizer.opt_func, self.model)
        self._reattach()

    def freeze(self):
        for p in self.model.parameters():
            p.requires_grad = False
        for p in self.model.head.parameters():
            p.requires_grad = True
        self._new_optimizer()

    def unfreeze(self):
        for p in self.model.parameters():
            p.requires_grad = True
        self._new_optimizer()

    def bn_freeze(self, freeze_type='non_head'):
        for m in self.model.modules():
            if isinstance(m, bn_types):
                for p in m.parameters():
                    p.requires_grad = False
        if freeze_type == 'non_head':
            for m in self.model.head.modules():
                if isinstance(m, bn_types):
                    for p in m.parameters():
                        p.requires_grad = True
        self._new_optimizer()
        self.bn_frozen = freeze_type

    def bn_unfreeze(self):
        for m in self.model.modules():
            if isinstance(m, bn_types):
                for p in m.parameters():
                    p.requires_grad = True
        self._new_optimizer()
        self.bn_frozen = None

    def _apply_bn_frozen(self):
        """Re-freeze BN modules after model.train() (General/Learner.py:589-594, :841-846)."""
        if self.bn_frozen in ['all', 'non_head']:
            for m in self.model.modules():
                if isinstance(m, bn_types):
                    m.training = False
        if self.bn_frozen == 'non_head':
            for m in self.model.head.modules():
                if isinstance(m, bn_types):
                    m.training = True

    # ---- (4) prediction / evaluation ----------------------------------------------------------------
    def predict1minibatch(self, x_batch):
        "model(*x) for list inputs else model(x)  (General/Learner.py:277-284)"
        return self.model(*x_batch) if isinstance(x_batch, list) else self.model(x_batch)

    def predict(self, dl, correct_probs=True, thresh=0.05, max_overlap=0.5,
                rel_thresh=None, top_k=1000, max_boxes=20, dup=None, inc=None):
        """Predictions for a whole dataloader, post-processed per target_type (General/Learner.py:286-393)."""
        if self.target_type == 'bbox' and dl not in ['val', 'test']:
            raise ValueError("Must use 'val' or 'test' dataloader for target_type = bbox.")
        which = dl
        if dl == 'val':
            dl = self.data.val_dl
        elif dl == 'test':
            dl = self.data.test_dl
        self.model.eval()
        out = []
        with torch.no_grad():
            for j, (x_batch, y_batch) in enumerate(dl):
                x_batch = to_cuda(x_batch)
                y_pred = self.predict1minibatch(x_batch)
                if isinstance(y_pred, tuple):
                    y_pred = y_pred[0]
                # per-batch results stay on the device; ONE device->host copy after the loop (the reference copies and
                # synchronises every minibatch: ARR(y_pred), Learner.py:356-372)
                if self.target_type == 'cont':
                    out.append(y_pred)
                elif self.target_type in ['cat', 'single_label', 'text_classify']:
                    out.append([y_pred if not correct_probs else F.log_softmax(y_pred, dim=1).exp(), None])
                elif self.target_type == 'multi_label':
                    sig = y_pred.sigmoid()
                    out.append([sig if correct_probs else y_pred, sig])
                elif self.target_type == 'bbox':
                    anchors, reg, clas = y_pred
                    B, Cl, Sc = self.model.BBoxPredictor(x_batch, reg, clas, anchors, thresh, max_overlap,
                                                         rel_thresh, top_k, max_boxes, dup, inc)
                    B, Cl, Sc = B[0], Cl[0], Sc[0]          # bs = 1 for 'val' / 'test' bbox loaders
                    ds = self.data.val_ds if which == 'val' else self.data.test_ds
                    out.append([list_mult(B, 1 / ds.images[j]['scale']), Cl, Sc])
        # LOCAL check: predict() has no collective (`if rank == 0: learner.predict('test')` after DP training must not hang);
        # the rank-uniform all-reduce of the flag lives in fit / evaluate / find_lr, which contain collectives anyway
        _raise_if_index_error(local=True)
        if self.target_type == 'cont':
            return ARR(torch.cat(out))
        if self.target_type in ['cat', 'single_label', 'text_classify']:
            probs = ARR(torch.cat([o[0] for o in out]))
            return [probs, probs.argmax(axis=1)]
        if self.target_type == 'multi_label':
            return [ARR(torch.cat([o[0] for o in out])), np.around(ARR(torch.cat([o[1] for o in out]))).astype(int)]
        return out

    def _allreduce_sums(self, values):
        d = _dist()
        if d is None:
            return values
        t = torch.tensor(values, dtype=torch.float64, device=default_device())
        d.all_reduce(t)
        return t.tolist()

    def evaluate(self, dataset_type, metrics=[]):
        """Size-weighted mean loss (+ accuracy and metrics for 'val')  (General/Learner.py:395-485).
        The reference synchronises 2-3 times per minibatch (`loss.item()`, the accuracy count, every metric); here the sums
        are accumulated ON THE DEVICE in fp64 — the same additions of the same fp32 values in the same order, so the result
        is bit-identical — and read back once per call."""
        use_end = any((m in end_metrics) for m in metrics if isinstance(m, str))
        self.model.eval()
        dev = default_device()
        n_seen = 0

        def as_f64(v):
            return v.detach().to(dev, torch.float64).reshape(()) if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float64, device=dev)

        if dataset_type == 'train':
            acc = torch.zeros((), dtype=torch.float64, device=dev)
            with torch.no_grad():
                for x_batch, y_batch in self.data.train_dl:
                    bs = _own_rows(self.data.train_dl, y_batch)
                    x_batch, y_batch = to_cuda(x_batch), to_cuda(y_batch)
                    acc += bs * as_f64(self.loss_func(self.predict1minibatch(x_batch), y_batch))
                    n_seen += bs
            total_loss, n_seen = self._allreduce_sums([acc.item(), n_seen])
            _raise_if_index_error()
            return total_loss / n_seen

        if dataset_type == 'val':
            Y, YPRED = [], []
            acc = torch.zeros(2 + len(metrics), dtype=torch.float64, device=dev)      # [loss, num_correct, metrics...]
            with torch.no_grad():
                for x_batch, y_batch in self.data.val_dl:
                    bs = _own_rows(self.data.val_dl, y_batch)
                    x_batch, y_batch = to_cuda(x_batch), to_cuda(y_batch)
                    y_pred = self.predict1minibatch(x_batch)
                    acc[0] += bs * as_f64(self.loss_func(y_pred, y_batch))
                    n_seen += bs
                    if use_end:
                        YPRED.append(y_pred); Y.append(y_batch)
                    for i, m in enumerate(metrics):
                        if isinstance(m, str) and m in end_metrics:
                            continue
                        acc[2 + i] += bs * as_f64(m(y_pred, y_batch))
                    if bs == 0:
                        pass                                  # ghost shard of a data-parallel loader: counts nothing
                    elif self.target_type in ['cat', 'single_label']:
                        acc[1] += (y_pred.max(dim=1)[1] == y_batch).sum()
                    elif self.target_type == 'multi_label':
                        acc[1] += (y_pred.sigmoid().round() == y_batch).sum()
            if use_end:
                YPRED, Y = torch.cat(YPRED), torch.cat(Y)
                for i, m in enumerate(metrics):
                    if isinstance(m, str) and m in end_metrics:
                        acc[2 + i] = n_seen * as_f64(end_metrics[m]()(YPRED, Y))
            host = acc.tolist()                                                      # the only device->host sync
            _raise_if_index_error()
            sums = self._allreduce_sums([host[0], n_seen, host[1]] + host[2:])
            total_loss, n_seen, num_correct = sums[0], sums[1], sums[2]
            metric_values = np.array(sums[3:]) / n_seen
            results = [total_loss / n_seen]
            if self.target_type in ['cat', 'single_label']:
                results.append(num_correct / n_seen)
            elif self.target_type == 'multi_label':
                results.append(num_correct / (n_seen * len(self.data.categories)))
            if metrics:
                results.append(metric_values)
            return results

    # ---- (5) training --------------------------------------------------------------------------------
    def train1minibatch(self, x_batch, y_batch, lr_batch, mom_batch=None, betas_batch=None, _defer=False):
        """One optimizer update on one minibatch; returns the minibatch loss as a float
        (General/Learner.py:490-516).  _defer (fit()'s loop only): a replayed step may return a _PendingLoss handle instead, whose
        float is read after the next step has been launched."""
        bs = _batch_size(y_batch)
        self._dp_weight = 1.0
        kd_offset = 0
        if self.grad_sync is not None:
            local, bs, full_bs, world = self._dp_batch_sizes(bs)
            # this rank's share of the global minibatch relative to an equal split (local_bs / global_bs weighting)
            self._dp_weight = local * world / max(bs, 1)
            if getattr(self.data.train_dl, 'dp_info', None) is not None:
                from ..dist import shard_bounds
                kd_offset = shard_bounds(bs, _rank(), world)[0]
            else:
                kd_offset = _rank() * local
        else:
            full_bs = self.data.bs
        if getattr(self, '_kd_step', None) is not None:
            from ..dist import drop_ctx
            if drop_ctx.enabled:
                drop_ctx.begin_step(self._kd_step, kd_offset, bs)
                self._kd_step += 1
        if bs < full_bs:
            lr_batch = list_mult(lr_batch, bs / full_bs)
        opt = self.optimizer
        opt.set_params(lr_batch, opt.wd, opt.bn_wd, opt.clip, **get_param_dict(mom_batch, betas_batch))
        if self._graph_warmup is not None:
            loss = self._graphed_step(x_batch, y_batch, defer=_defer)
            if loss is not None:
                return loss
        return self._eager_step(x_batch, y_batch)

    def _eager_step(self, x_batch, y_batch):
        """zero_grad -> forward -> loss -> backward -> Optimizer.step -> the loss as a float (General/Learner.py:506-516).
        The reference calls `loss.item()` last, which drains the GPU once per step.  Here the scalar starts its device->host
        copy (pinned buffer + event) as soon as the forward has produced it, and the float is read after the backward and the
        optimizer kernels have been LAUNCHED: the value is the same, but the host is already preparing the next minibatch while
        the GPU finishes this one, so the device never idles between steps."""
        opt = self.optimizer
        opt.opt.zero_grad()
        if self.grad_sync is not None:
            self.grad_sync.begin(self._dp_weight)
        on_gpu = default_device().type == 'cuda'
        if on_gpu:
            from .. import ops
            ops.prepare_forward(self.model)              # the Winograd filters of all conv layers in one launch (ops.prepare_forward)
        try:
            y_pred = self.predict1minibatch(x_batch)
            loss = self.loss_func(y_pred, y_batch)
            early = loss.is_cuda and loss.numel() == 1 and not torch.cuda.is_current_stream_capturing()
            if early:
                if self._loss_host is None:
                    self._loss_host = torch.empty((), dtype=torch.float32).pin_memory()
                    self._loss_event = torch.cuda.Event()
                self._loss_host.copy_(loss.detach().reshape(()).float(), non_blocking=True)
                self._loss_event.record()
            self._backward(loss)                         # (closes the prepared-filter window in its own `finally`)
        except BaseException:
            if on_gpu:
                ops.finish_backward()                    # the window must not outlive a failed step: the next forward may follow a weight change
            raise
        opt.step()
        if early:
            self._loss_event.synchronize()
            return float(self._loss_host)
        return loss.item()

    def _backward(self, loss):
        "loss.backward() with the per-pass preparation of the HIP layers (all conv filter transposes in one launch)"
        if loss.is_cuda:
            from .. import ops
            ops.prepare_backward(self.model)
            try:
                loss.backward()
            finally:
                ops.finish_backward()
        else:
            loss.backward()

    def _graphed_step(self, x_batch, y_batch, defer=False):
        "use_graphs(): replay (or first capture) the step for this input signature; None -> run it eagerly"
        leaves = _tensor_leaves(x_batch) + _tensor_leaves(y_batch)
        if any(t is None for t in leaves) or not self.optimizer.graph_capturable() or not self.model.training:
            return None
        if getattr(self, '_graph_stateless', None) is None:
            # a forward that carries Python-side state from one minibatch to the next (the language model's hidden state,
            # Text.py:547-550) or draws per-call host randomness (the weight-drop seed) cannot be replayed: such modules mark
            # themselves `nnl_stateful_forward` and the step stays eager whatever use_graphs() was asked for
            self._graph_stateless = not any(getattr(m, 'nnl_stateful_forward', False) for m in self.model.modules())
        if not self._graph_stateless:
            return None
        from ..dist import drop_ctx
        if drop_ctx.enabled:
            # keyed masks bake (step, request index) — Python ints — into the captured kernels' arguments: a replay would
            # apply the capture step's masks for ever.  torch's own Dropout is graph-safe (Philox offset), the keyed path is not.
            return None
        if self.grad_sync is not None:
            if getattr(self, '_dp_graph_ok', None) is None:
                # collectives INSIDE the forward / backward are not captured: SyncBN (between kernels of both passes) keeps the step
                # eager; the tabular renorm sync only needs the step's INPUTS, so its all-gather runs before the graph
                # (StructuredDataNet.nnl_dp_prepare) and the step replays (round 5)
                mods = list(self.model.modules())
                self._dp_prepare_mods = [m for m in mods if getattr(m, 'nnl_dp', None) is not None and hasattr(m, 'nnl_dp_prepare')]
                self._dp_graph_ok = not any(getattr(m, 'nnl_sync', None) is not None
                                            or (getattr(m, 'nnl_dp', None) is not None and not hasattr(m, 'nnl_dp_prepare')) for m in mods)
            if not self._dp_graph_ok:
                return None
        key = (tuple((tuple(t.shape), t.dtype) for t in leaves), self.optimizer.clip is not None and bool(self.optimizer.clip))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 4:
                return None
            g = self._graphs[key] = _GraphedStep(self._graph_warmup)
        return g.run(self, x_batch, y_batch, defer)

    @staticmethod
    def display_training_results(col_names, values, run_times):
        "Epoch table (General/Learner.py:518-526)."
        if _rank() != 0 or not Learner.verbose:
            return
        print("epoch".ljust(8) + "".join(c.ljust(12) for c in col_names) + '\n')
        for n, row in enumerate(values):
            print(str(n).ljust(8) + "".join('{:.5f}'.format(v).ljust(12) for v in row) + run_times[n])

    def train_gen_sched(self, lr_sched, mom_sched, betas_sched, metrics=[], print_batch=False,
                        save_name=None, save_method='best', swa_freq=None):
        """Train with explicit per-minibatch schedules (General/Learner.py:528-678)."""
        if save_name is None:
            save_method = None
        n_batches = len(self.data.train_dl)
        if len(lr_sched) % n_batches != 0:
            raise ValueError("len(lr_sched) must be an integer multiple of len(learner.data.train_dl).")
        num_epochs = len(lr_sched) // n_batches

        self.loss_sched, self.lr_sched, self.mom_sched, self.betas_sched = [], [], [], []
        self.moving_avg_loss = 0
        min_loss = self.evaluate('val')[0]                      # pre-training validation pass (:566)
        if save_name:
            self.save(save_name)

        classify = self.target_type in ['cat', 'single_label', 'multi_label']
        values, run_times = [], []
        col_names = ['train_loss', 'val_loss'] + (['accuracy'] if classify else []) + (['metrics'] if metrics != [] else [])
        self.display_training_results(col_names, values, run_times)

        if swa_freq:
            swa_model, swa_count = copy.deepcopy(self.model), 1

        debiased = 0.
        for n in range(num_epochs):
            t0 = time.time()
            self.model.train()
            self._apply_bn_frozen()

            # replayed steps (use_graphs): the loss of step i is read after step i + 1 has been launched, so the GPU does not idle while
            # the host stages the next minibatch; the values, their order and everything derived from them are unchanged
            defer = self._graph_warmup is not None and self.grad_sync is None and print_batch is False
            pending = None

            def consume(loss_value):
                self.loss_sched.append(loss_value)
                self.moving_avg_loss = self.moving_avg_loss * 0.98 + loss_value * 0.02
                return self.moving_avg_loss / (1 - 0.98 ** len(self.loss_sched))

            for j, (x_batch, y_batch) in enumerate(self.data.train_dl):
                tb = time.time()
                x_batch, y_batch = to_cuda(x_batch), to_cuda(y_batch)
                i = n * n_batches + j
                self.lr_sched.append(lr_sched[i])
                if mom_sched:
                    self.mom_sched.append(mom_sched[i])
                    loss = self.train1minibatch(x_batch, y_batch, lr_sched[i], mom_batch=mom_sched[i], _defer=defer)
                elif betas_sched:
                    self.betas_sched.append(betas_sched[i])
                    loss = self.train1minibatch(x_batch, y_batch, lr_sched[i], betas_batch=betas_sched[i], _defer=defer)
                else:
                    loss = self.train1minibatch(x_batch, y_batch, lr_sched[i], _defer=defer)
                if pending is not None:
                    debiased = consume(pending.result())
                    pending = None
                if isinstance(loss, _PendingLoss):
                    pending = loss
                    continue
                debiased = consume(loss)

                if (print_batch is True) or (type(print_batch) == int and not isinstance(print_batch, bool)
                                             and (j % print_batch) == 0):
                    self._print_batch(j, debiased, loss, metrics, x_batch, y_batch, time.time() - tb)
            if pending is not None:
                debiased = consume(pending.result())
                pending = None

            _raise_if_index_error()                       # bad ids met by this epoch's gathers (checked once, not per step)
            if self.grad_sync is not None:
                self.grad_sync.raise_if_overlap_error()   # a bucket wait kernel of the replayed data-parallel step timed out
            train_loss = debiased if self.use_moving_avg else self.evaluate('train')

            res = self.evaluate('val', metrics)
            val_loss = res[0]
            row = [train_loss, val_loss]
            if classify:
                row.append(res[1])
            if metrics != []:
                row += [mv for mv in res[-1]]
            values.append(row)

            mins, secs = divmod(time.time() - t0, 60)
            run_times.append("  epoch run time: %d min, %.2f sec" % (mins, secs))
            self.display_training_results(col_names, values, run_times)

            if val_loss < min_loss:
                min_loss = val_loss
                if save_method == 'best':
                    self.save(save_name)
            if save_method == 'all':
                self.save(save_name + '_' + str(n))

            if swa_freq and (n + 1) % swa_freq == 0:
                swa_model = combine_models([swa_model, self.model], [swa_count / (swa_count + 1), 1 / (swa_count + 1)])
                swa_count += 1

            if val_loss > 20 * min_loss:
                if _rank() == 0:
                    print('val_loss increased too much, stopping training early')
                break

        if swa_freq:
            self.model = swa_model
            self._graphs, self._graph_stateless = {}, None   # captured steps point at the replaced model's parameters

    def _print_batch(self, j, debiased, loss, metrics, x_batch, y_batch, dt):
        if _rank() != 0:
            return
        if j == 0:
            names = ['avg_loss', 'batch_loss'] + (['batch_metrics'] if len(metrics) > 0 else [])
            print("batch".ljust(8) + "".join(c.ljust(12) for c in names))
        vals = [debiased, loss]
        if len(metrics) > 0:
            with torch.no_grad():
                y_pred = self.predict1minibatch(x_batch)
                vals += [m(y_pred, y_batch).item() for m in metrics]
        print(str(j).ljust(8) + "".join('{:.5f}'.format(v).ljust(12) for v in vals) + ("batch run time: %.2f" % dt))

    def init_optimizer(self, wd=None, bn_wd=None, clip=None):
        "Fix wd / bn_wd / clip for one training period; unspecified values keep the last ones (:680-688)."
        WD = wd if wd else self.optimizer.wd
        BN_WD = bn_wd if (bn_wd is not None) else self.optimizer.bn_wd
        CLIP = clip if clip else self.optimizer.clip
        self.optimizer.set_params(lr=0, wd=WD, bn_wd=BN_WD, clip=CLIP)

    @staticmethod
    def get_sched(sched_type, N, start_val, end_val):
        """N schedule points of type 'linear' | 'cos' | 'exp' | 'poly' from start_val to end_val; values may be
        per-layer-group vectors (General/Learner.py:691-728)."""
        if type(start_val) == list:
            start_val = np.array(start_val)
        # NB: the reference assigns the converted end_val to a misspelt name (Learner.py:716), so a list
        # end_val stays a list; numpy broadcasting makes every formula below work on it unchanged.
        if sched_type == 'linear':
            return list(linear_space(start_val, end_val, N))
        if sched_type == 'cos':
            s = 0.5 * (np.cos(np.linspace(0, np.pi, N)) + 1)
            return list(end_val + outer_mult(start_val - end_val, s))
        if sched_type == 'exp':
            return list(np.exp(linear_space(np.log(start_val), np.log(end_val), N)))
        if sched_type == 'poly':
            p = np.log(end_val / start_val) / np.log(N)
            return [start_val * i ** p for i in range(1, N + 1)]

    def _check_lr_list(self, lr, name):
        if type(lr) == list and len(lr) != len(self.model.layer_groups):
            raise ValueError("If <%s> is a list, must have len(%s) = len(learner.model.layer_groups)." % (name, name))

    def fit(self, lr, num_epochs, wd=None, bn_wd=None, clip=None, momentum=None, betas=None,
            metrics=[], print_batch=False, save_name=None, save_method='best', swa_freq=None):
        "Constant lr / momentum / betas for num_epochs (General/Learner.py:730-744)."
        self._check_lr_list(lr, 'lr')
        self.init_optimizer(wd, bn_wd, clip)
        N = num_epochs * len(self.data.train_dl)
        self.train_gen_sched([lr] * N, [momentum] * N if momentum else None, [betas] * N if betas else None,
                             metrics, print_batch, save_name, save_method, swa_freq)

    def fit_cycles(self, lr_start, lr_end, num_cycles, cycle_type='cos', base_length=1, cycle_mult=1,
                   wd=None, bn_wd=None, clip=None, momentum=None, betas=None, metrics=[],
                   print_batch=False, save_name=None, save_method='best', swa_freq=None):
        "Annealed lr with restarts (General/Learner.py:746-774)."
        self._check_lr_list(lr_start, 'lr_start')
        self._check_lr_list(lr_end, 'lr_end')
        self.init_optimizer(wd, bn_wd, clip)
        lr_sched = []
        mom_sched = [] if momentum else None
        betas_sched = [] if betas else None
        cycle_length = base_length
        for c in range(num_cycles):
            if c > 0:
                cycle_length = cycle_length * cycle_mult
            N = len(self.data.train_dl) * cycle_length
            lr_sched += self.get_sched(cycle_type, N, lr_start, lr_end)
            if momentum:
                mom_sched += [momentum] * N
            if betas:
                betas_sched += [betas] * N
        self.train_gen_sched(lr_sched, mom_sched, betas_sched, metrics, print_batch, save_name, save_method, swa_freq)

    def fit_one_cycle(self, lr_max, num_epochs, div_fac=25, start_pct=0.3, wd=None, bn_wd=None,
                      clip=None, mom_min=0.85, mom_max=0.95, beta_min=0.85, beta_max=0.95,
                      metrics=[], print_batch=False, save_name=None, save_method='best'):
        "One-cycle policy (General/Learner.py:776-802)."
        self._check_lr_list(lr_max, 'lr_max')
        if type(lr_max) == list:
            lr_max = np.array(lr_max)
        self.init_optimizer(wd, bn_wd, clip)
        N = num_epochs * len(self.data.train_dl)
        N1, N2 = int(N * start_pct), N - int(N * start_pct)
        lr_min = lr_max / div_fac
        lr_sched = self.get_sched('linear', N1, lr_min, lr_max) + self.get_sched('cos', N2, lr_max, lr_min / 1e4)
        mom_sched, betas_sched = None, None
        pg0 = self.optimizer.opt.param_groups[0]
        if 'momentum' in pg0:
            mom_sched = self.get_sched('linear', N1, mom_max, mom_min) + self.get_sched('cos', N2, mom_min, mom_max)
        if 'betas' in pg0:
            b = self.get_sched('linear', N1, beta_max, beta_min) + self.get_sched('cos', N2, beta_min, beta_max)
            betas_sched = [(float(v), 0.99) for v in b]
        self.train_gen_sched(lr_sched, mom_sched, betas_sched, metrics, print_batch, save_name, save_method)

    def find_lr(self, lr_min=1e-5, lr_max=1.0, wd=None, bn_wd=None, clip=None, momentum=None,
                betas=None, length='1epoch', break_fac=3, sched_type='exp', smoothing_radius='default',
                plot_start_batch=0, plot=True):
        """LR range test; model/optimizer state is restored afterwards (General/Learner.py:804-887).
        `plot=False` (new) skips the matplotlib figure."""
        self._check_lr_list(lr_max, 'lr_max')
        self._check_lr_list(lr_min, 'lr_min')
        self.save('temp', save_optimizer=True)
        d = _dist()
        if d:
            d.barrier()
        self.moving_avg_loss = 0
        self.loss_sched, self.lr_sched, self.mom_sched, self.betas_sched = [], [], [], []
        self.init_optimizer(wd, bn_wd, clip)
        self.model.train()
        self._apply_bn_frozen()

        n_batches = len(self.data.train_dl)
        N = n_batches if length == '1epoch' else length
        num_epochs = int(np.ceil(N / n_batches))
        lr_sched = self.get_sched(sched_type, N, lr_min, lr_max)
        for n in range(num_epochs):
            for j, (x_batch, y_batch) in enumerate(self.data.train_dl):
                i = n * n_batches + j
                x_batch, y_batch = to_cuda(x_batch), to_cuda(y_batch)
                if momentum:
                    loss = self.train1minibatch(x_batch, y_batch, lr_sched[i], mom_batch=momentum)
                elif betas:
                    loss = self.train1minibatch(x_batch, y_batch, lr_sched[i], betas_batch=betas)
                else:
                    loss = self.train1minibatch(x_batch, y_batch, lr_sched[i])
                self.loss_sched.append(loss)
                self.lr_sched.append(lr_sched[i])
                self.moving_avg_loss = self.moving_avg_loss * 0.98 + loss * 0.02
                debiased = self.moving_avg_loss / (1 - 0.98 ** (i + 1))
                if i == 0:
                    initial_loss = debiased
                # as upstream, a break only leaves the current epoch's loop (Learner.py:866-867)
                if (break_fac and debiased > break_fac * initial_loss) or i == N - 1:
                    break

        _raise_if_index_error()
        if plot and _rank() == 0:
            import matplotlib.pyplot as plt
            fig = plt.figure(figsize=(12, 6))
            r = int(max(5, N / 50)) if smoothing_radius == 'default' else smoothing_radius
            smooth = self.smooth_timeseries(self.loss_sched, r)
            sp = fig.add_subplot(1, 2, 1)
            plt.plot(range(plot_start_batch, len(self.lr_sched)), self.lr_sched[plot_start_batch:])
            sp.set(xlabel='minibatch', ylabel='learning rate')
            sp = fig.add_subplot(1, 2, 2)
            plt.plot(self.lr_sched[plot_start_batch:], smooth[plot_start_batch:])
            if sched_type == 'linear':
                sp.set(xlabel='learning rate', ylabel='train loss')
            elif sched_type == 'exp':
                sp.set_xscale('log'); sp.set(xlabel='learning rate (log scale)', ylabel='train loss')

        self.load('temp', saved_optimizer=True)
