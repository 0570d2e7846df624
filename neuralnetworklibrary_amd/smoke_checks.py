"""Small GPU-vs-oracle checks used by __graft_entry__.smoke() (imports oracle/ as the checker only)."""
import torch


def _close(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    ok = bool((err <= atol + rtol * b.abs()).all())
    if not ok:
        raise AssertionError(f'smoke: {what} mismatch, max abs err {err.max().item():.3e}')


def check_collab(device):
    from oracle import reference_math as RM
    from . import ops
    g = torch.Generator().manual_seed(0)
    n, nu, ni, D = 64, 943, 1682, 30
    x = torch.stack([torch.randint(0, nu, (n,), generator=g), torch.randint(0, ni, (n,), generator=g)], 1)
    ps = [torch.randn(nu, D, generator=g) * .3, torch.randn(ni, D, generator=g) * .3,
          torch.randn(nu, 1, generator=g), torch.randn(ni, 1, generator=g)]
    dy = torch.randn(n, generator=g)
    cpu = [p.clone().requires_grad_(True) for p in ps]
    ref = RM.embdotbias(x, *cpu, [0.8, 5.2]); ref.backward(dy)
    gpu = [p.clone().to(device).requires_grad_(True) for p in ps]
    out = ops.embdotbias(x.to(device), *gpu, [0.8, 5.2]); out.backward(dy.to(device))
    _close(out, ref, 1e-5, 1e-5, 'embdotbias fwd')
    for a, b in zip(gpu, cpu):
        _close(a.grad, b.grad, 1e-3, 1e-5, 'embdotbias grad')


CHECKS = [check_collab]


def run_all(device='cuda:0'):
    for c in CHECKS:
        c(device)
        print('  smoke check passed:', c.__name__)
