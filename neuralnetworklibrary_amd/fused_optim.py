"""Host side of the fused multi-tensor optimizer step (K8, nnl_optim_step): builds the device descriptor / chunk tables
for an `Optimizer` wrapper and runs weight decay + clip + SGD-momentum / Adam for all parameters in 1 (3 with clipping)
launches.  The torch optimizer object stays the owner of the state tensors (`opt.state[p]` is filled with the tensors the
kernel updates), so `opt.state_dict()` / `load_state_dict()` — used by Learner.save / load / find_lr — keep working."""
import numpy as np
import torch
import torch.optim as optim

from ._lib import check, lib, ptr, stream

_DESC = np.dtype([('param', '<u8'), ('grad', '<u8'), ('s1', '<u8'), ('s2', '<u8'), ('numel', '<i8'), ('lr', '<f4'), ('decay', '<f4')])


def _dense_like(a, b):
    return a.shape == b.shape and a.stride() == b.stride()


class FusedStep:
    """One instance per Optimizer wrapper; `supported(opt)` says whether the torch optimizer can be replaced."""

    @staticmethod
    def supported(opt, params):
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return False
        if not all(p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)) for p in params):
            return False
        g0 = opt.param_groups[0]
        if isinstance(opt, optim.SGD):
            return all(g.get('dampening', 0) == 0 and not g.get('nesterov', False) and g.get('weight_decay', 0) == 0
                       and not g.get('maximize', False) for g in opt.param_groups)
        if type(opt) is optim.Adam:
            return all(not g.get('amsgrad', False) and g.get('weight_decay', 0) == 0 and not g.get('maximize', False)
                       and not g.get('capturable', False) and not torch.is_tensor(g['lr']) for g in opt.param_groups) and 'betas' in g0
        return False

    def __init__(self, opt):
        self.opt = opt
        self.kind = 0 if isinstance(opt, optim.SGD) else 1
        self.params, self.group_of = [], []
        for gi, g in enumerate(opt.param_groups):
            for p in g['params']:
                self.params.append(p)
                self.group_of.append(gi)
        self.device = self.params[0].device
        chunk = int(lib.nnl_optim_chunk_elems())
        ct, co = [], []
        for ti, p in enumerate(self.params):
            for off in range(0, p.numel(), chunk):
                ct.append(ti); co.append(off)
        self.n_chunks = len(ct)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=self.device)
        self.chunk_off = torch.tensor(co, dtype=torch.int64, device=self.device)
        n = len(self.params)
        # one upload per step: [n descriptors | 8 floats of hyper-parameters]; two pinned staging buffers alternate so that
        # step t+1 can be prepared while step t's copy is still in flight
        self.table_bytes = n * _DESC.itemsize
        self.host = [torch.empty(self.table_bytes + 32, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev = torch.empty(self.table_bytes + 32, dtype=torch.uint8, device=self.device)
        # replayed steps: the per-step values (lr, decay per tensor + 8 hyper floats) reach the device through `dyn`, uploaded eagerly
        # from a ring of pinned buffers before each replay and patched into the table by a captured kernel (nnl_optim_patch)
        self.dyn = torch.zeros(2 * n + 8, dtype=torch.float32, device=self.device)
        self._ring, self._ring_ev, self._ring_k = None, None, 0
        self._capture_dyn = None
        self.clip_ws = torch.empty(self.n_chunks + 2, dtype=torch.float32, device=self.device)
        self.flip = 0
        self.steps = 0
        self._capture_buf = None            # staging buffer for the next captured step (Learner.use_graphs)
        self._init_state()

    def _init_state(self):
        "state tensors live in opt.state (created here like torch would create them lazily at the first step)"
        for p in self.params:
            st = self.opt.state[p]
            if self.kind == 0:
                if st.get('momentum_buffer') is None:
                    st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            else:
                if 'exp_avg' not in st:
                    if not hasattr(self, '_shared_step'):
                        self._shared_step = torch.tensor(0.0, dtype=torch.float32)
                    st['step'] = self._shared_step
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def uniform_hyper(self):
        "momentum / betas / eps must be equal across param groups for the single fused launch"
        gs = self.opt.param_groups
        if self.kind == 0:
            return len({g['momentum'] for g in gs}) == 1
        return len({(tuple(g['betas']), g['eps']) for g in gs}) == 1

    def _advance(self):
        "bump the step counters; returns the hyper-parameter vector the kernel reads (include/nnl.h, nnl_optim_step)"
        g0 = self.opt.param_groups[0]
        self.steps += 1
        if self.kind == 0:
            return [float(g0['momentum']), 0., 0., 0., 1., 1.]
        # fresh states share ONE counter tensor; loaded ones may not.  The list of distinct counters is cached against the identity of
        # the first one (a load_state_dict replaces the tensors): walking every parameter's state cost 8 us per replayed step
        st0 = self.opt.state[self.params[0]]['step']
        cache = getattr(self, '_step_cache', None)
        if cache is None or cache[0] is not st0:
            seen, uniq = set(), []
            for p in self.params:
                t = self.opt.state[p]['step']
                if id(t) not in seen:
                    seen.add(id(t))
                    uniq.append(t)
            cache = self._step_cache = (st0, uniq)
        step = int(st0) + 1
        for t in cache[1]:
            t += 1
        b1, b2 = (float(b) for b in g0['betas'])
        return [0., b1, b2, float(g0['eps']), 1.0 - b1 ** step, float(np.sqrt(1.0 - b2 ** step))]

    def _advance_hyper(self, clip):
        """the 8 floats nnl_optim_step reads: {momentum | 1 - beta1, beta1, beta2, eps, bc1, sqrt(bc2), clip max_norm, 1 - beta2}.
        1 - beta is formed HERE in double and rounded once, as torch.optim.Adam does (`addcmul_(g, g, value=1 - beta2)`): formed on
        the device in fp32, 1 - 0.999f is off by 1.3e-5 relative, i.e. 2.7e-6 absolute on a weight after three lr = 0.1 steps (the
        direct G9 comparison with the reference found it)."""
        h = self._advance()
        if self.kind == 1:
            b1, b2 = (float(b) for b in self.opt.param_groups[0]['betas'])
            return [1.0 - b1] + h[1:] + [float(clip or 0.), 1.0 - b2]
        return h + [float(clip or 0.), 0.]

    def step(self, lrs, decays, clip):
        """lrs / decays: one value per torch param group (decay = 1 - wd*lr or 1.0)."""
        from . import ops
        ops.side_join()                                     # weight gradients computed on the side stream (ops._Side) must have landed
        self._init_state()                                  # state may have been replaced by opt.load_state_dict
        desc = np.zeros(len(self.params), dtype=_DESC)
        keep = []
        for i, p in enumerate(self.params):
            st = self.opt.state[p]
            g = p.grad
            if g is not None and not (_dense_like(g, p) and g.dtype == torch.float32):
                g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                g2.copy_(g)
                p.grad = g = g2
            keep.append(g)
            desc[i] = (p.data_ptr(), 0 if g is None else g.data_ptr(),
                       (st['momentum_buffer'] if self.kind == 0 else st['exp_avg']).data_ptr(),
                       0 if self.kind == 0 else st['exp_avg_sq'].data_ptr(), p.numel(),
                       lrs[self.group_of[i]], decays[self.group_of[i]])
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # the captured H2D copy node re-reads its pinned source at every replay, so each captured step owns one staging
            # buffer (allocated by prepare_capture(): hipHostMalloc is illegal while capturing); replay_update() rewrites
            # lr / decay / hyper in it.  `keep` pins the graph-pool gradient tensors the table points at.
            if self._capture_buf is None:
                raise RuntimeError("FusedStep.step() under stream capture without prepare_capture()")
            h, self._capture_buf = self._capture_buf, None
            self.last_capture = (h, keep)
        else:
            h = self.host[self.flip]
            self.flip ^= 1
        hn = h.numpy()
        hn[:self.table_bytes] = desc.view(np.uint8)
        hyper_vec = self._advance_hyper(clip)
        hn[self.table_bytes:].view(np.float32)[:] = hyper_vec
        self.dev.copy_(h, non_blocking=True)
        hyper = self.dev.data_ptr() + self.table_bytes
        if capturing:
            # the captured upload re-reads the pinned image at every replay: the host never rewrites it again; what changes per step
            # comes through `dyn` (stage_last() / replay_update(), eager) and is patched in by this captured kernel
            g = np.asarray(self.group_of)
            self._capture_dyn = (np.asarray(lrs, dtype=np.float32)[g], np.asarray(decays, dtype=np.float32)[g], hyper_vec)
            check(lib.nnl_optim_patch(ptr(self.dev), hyper, ptr(self.dyn), len(self.params), stream()))
        check(lib.nnl_optim_step(ptr(self.dev), ptr(self.chunk_tensor), ptr(self.chunk_off), self.n_chunks, self.kind,
                                 hyper, 1 if clip else 0, ptr(self.clip_ws), stream()))

    def prepare_capture(self):
        "call before torch.cuda.graph(...) around a step(): allocates the staging buffer that capture will bake in"
        self._init_state()
        self._capture_buf = torch.empty(self.table_bytes + 32, dtype=torch.uint8).pin_memory()

    def _stage_dyn(self, lr_t, decay_t, hyper_vec):
        "eager, stream-ordered upload of one step's values into `dyn` from the next pinned ring slot (8 slots: a slot is reused only after its copy has completed)"
        n = len(self.params)
        if self._ring is None:
            self._ring = [torch.empty(2 * n + 8, dtype=torch.float32).pin_memory() for _ in range(8)]
            self._ring_ev = [None] * 8
        k = self._ring_k
        self._ring_k = (k + 1) % 8
        if self._ring_ev[k] is not None:
            self._ring_ev[k].synchronize()
        else:
            self._ring_ev[k] = torch.cuda.Event()
        r = self._ring[k].numpy()
        r[0:2 * n:2] = lr_t
        r[1:2 * n:2] = decay_t
        r[2 * n:] = hyper_vec
        self.dyn.copy_(self._ring[k], non_blocking=True)
        self._ring_ev[k].record(self._stream())             # (Event.record() without a stream looks the current one up: 10 us of Python)

    def _stream(self):
        "torch's current stream, looked up once per thread-visible change of it (the replay path always runs on one stream)"
        s = getattr(self, '_cur_stream', None)
        if s is None:
            s = self._cur_stream = torch.cuda.current_stream()
        return s

    def stage_last(self):
        "right after a capture, before its first replay: the values the captured step() was called with"
        if self._capture_dyn is not None:
            self._cur_stream = torch.cuda.current_stream()  # the stream the graph is replayed on
            self._stage_dyn(*self._capture_dyn)
            self._capture_dyn = None

    def replay_update(self, capture, lrs, decays, clip):
        """Before replaying a captured training step: this step's per-tensor lr / decay and hyper-parameter vector go to `dyn`
        (the captured nnl_optim_patch writes them into the table); the pinned image the captured upload re-reads is never touched
        again, so the host may stage step i + 1 while step i is still running (Learner.fit's deferred loss read-back)."""
        g = np.asarray(self.group_of)
        self._stage_dyn(np.asarray(lrs, dtype=np.float32)[g], np.asarray(decays, dtype=np.float32)[g], self._advance_hyper(clip))
