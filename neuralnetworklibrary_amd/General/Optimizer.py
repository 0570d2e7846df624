"""`Optimizer` wrapper of the drop-in API (mirror of the reference's General/Optimizer.py:9-96).

Same contract: 2*NL torch param groups `[reg_1..reg_NL, bn_1..bn_NL]` built from `model.param_groups`
(Optimizer.py:36-39); `set_params` broadcasts per-layer-group hyper-parameters with `i % NL`
(Optimizer.py:41-52); `step` = decoupled weight decay `X *= 1 - wd_g*lr_g` on the reg groups (and the bn
groups iff bn_wd) -> global-norm clip over model.parameters() -> `opt.step()` (Optimizer.py:58-70).

MI355X additions (absent upstream): when a data-parallel `GradSync` is attached
(`attach_grad_sync`), `step()` first waits for the bucketed RCCL all-reduce of the gradients that was
launched from backward hooks, so every rank applies the identical update.
"""
import torch

from .Core import LIST, trainable_params

__all__ = ['get_param_dict', 'Optimizer']


def get_param_dict(momentum=None, betas=None):
    "Dictionary of extra optimizer hyper-parameters (General/Optimizer.py:9-14)."
    d = {}
    if momentum:
        d['momentum'] = momentum
    if betas:
        d['betas'] = betas
    return d


class Optimizer(object):
    """Wrapper around a torch.optim optimizer class with per-layer-group lr, decoupled weight decay and
    global gradient clipping.  Arguments/attributes as in the reference (General/Optimizer.py:16-39)."""

    def __init__(self, opt_func, model, wd=None, bn_wd=True, clip=None):
        self.model, self.opt_func, self.NL = model, opt_func, len(model.layer_groups)
        self.lr, self.wd, self.bn_wd, self.clip = [0] * self.NL, wd, bn_wd, clip
        self.opt = opt_func([{'params': trainable_params(pg), 'lr': 0} for pg in model.param_groups])
        self.grad_sync = None
        self._fused = None

    def attach_grad_sync(self, grad_sync):
        self.grad_sync = grad_sync

    def set_params(self, lr, wd=None, bn_wd=True, clip=None, **kwargs):
        lr = LIST(lr, self.NL)
        if wd:
            wd = LIST(wd, self.NL)
        for par in kwargs:
            kwargs[par] = LIST(kwargs[par], self.NL, Tuple=False)
        self.lr, self.wd, self.bn_wd, self.clip = lr, wd, bn_wd, clip
        kwargs['lr'] = lr
        for par, vals in kwargs.items():
            for i, pg in enumerate(self.opt.param_groups):
                pg[par] = vals[i % self.NL]

    def grad_clip(self):
        if self.clip:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)

    def _fused_stepper(self):
        """The fused HIP multi-tensor step (K8) when every trainable parameter is a dense fp32 CUDA tensor and the torch
        optimizer is plain SGD(momentum) / Adam; otherwise None (torch's own kernels run, e.g. in CPU host-logic tests)."""
        if self._fused is None:
            self._fused = False
            import os
            params = [p for g in self.opt.param_groups for p in g['params']]
            if params and params[0].is_cuda and os.environ.get('NNL_FUSED_OPTIM', '1') != '0':
                from ..fused_optim import FusedStep
                if FusedStep.supported(self.opt, params):
                    self._fused = FusedStep(self.opt)
        return self._fused or None

    def _fused_args(self):
        "per torch-param-group learning rates and decoupled weight-decay factors (1 - wd_g*lr_g, or 1.0 where it does not apply)"
        groups = self.opt.param_groups
        lrs = [float(g['lr']) for g in groups]
        decays = []
        for i in range(len(groups)):
            layer, is_bn = i % self.NL, i >= self.NL
            apply = bool(self.wd) and (not is_bn or self.bn_wd)
            decays.append(1 - self.wd[layer] * self.lr[layer] if apply else 1.0)
        return lrs, decays

    def graph_capturable(self):
        """True when step() is a fixed sequence of launches whose hyper-parameters are read from device memory (data parallel: the
        step itself stays outside the graph, after the eager all-reduces — Learner._GraphedStep)"""
        fused = self._fused_stepper()
        return fused is not None and fused.uniform_hyper()

    def prepare_capture(self):
        self._fused.prepare_capture()

    def captured(self):
        "handle of the step() that was just recorded under stream capture (pass it to replay_step)"
        return self._fused.last_capture

    def stage_captured(self):
        "right after a capture: upload the captured step's own per-step values before its first replay (FusedStep.stage_last)"
        self._fused.stage_last()

    def replay_step(self, capture):
        "Host half of step() when the launches themselves are replayed from a captured hipGraph (Learner.use_graphs)."
        lrs, decays = self._fused_args()
        self._fused.replay_update(capture, lrs, decays, self.clip)

    def step(self):
        if torch.cuda.is_available():
            from .. import ops
            ops.side_join()                                 # (weight gradients on the side stream: a no-op unless a backward used it)
        if self.grad_sync is not None:
            self.grad_sync.finish()
        fused = self._fused_stepper()
        if fused is not None and fused.uniform_hyper():
            lrs, decays = self._fused_args()
            fused.step(lrs, decays, self.clip)
            return
        if self.wd:
            reg_groups = self.opt.param_groups[:self.NL]
            bn_groups = self.opt.param_groups[self.NL:]
            with torch.no_grad():
                for lr, wd, pg1, pg2 in zip(self.lr, self.wd, reg_groups, bn_groups):
                    decay = 1 - wd * lr
                    if pg1['params']:
                        torch._foreach_mul_(pg1['params'], decay)
                    if self.bn_wd and pg2['params']:
                        torch._foreach_mul_(pg2['params'], decay)
        self.grad_clip()
        self.opt.step()

    def print_summary(self, print_param_groups=True):
        print('optimizer.model = ', self.model)
        print('optimizer.opt = ', self.opt)
        if print_param_groups:
            for pg in self.opt.param_groups:
                print(pg)
                print('')
        for name in ['NL', 'lr', 'wd', 'clip', 'bn_wd']:
            print('optimizer.%s = ' % name, getattr(self, name))

    def print_params_grads(self):
        for j, pg in enumerate(self.model.param_groups):
            print('PG', j, '=', pg)
            print('')
            for i, X in enumerate(pg.parameters()):
                print('parameter', i, '=', X)
                print('parameter', i, 'grad =', X.grad)
                if X.grad is None:
                    print('')
            print('')
