"""Device-resident input pipelines for the array-backed datasets of the reference.

The reference feeds `Learner.fit` through `torch.utils.data.DataLoader` workers: per minibatch a Python `__getitem__` loop,
a numpy collate (`StructuredDataCollater`, StructuredData.py:849-869), a pinned H2D copy and `to_cuda` (Learner.py:599).
For the collaborative-filtering and structured-data heads the whole dataset is a few hundred MB at most (Rossmann:
844 k rows x (32 int64 + 14 fp32) = 263 MB; MovieLens-20M: 20 M x (2 int64 + 1 fp32) = 400 MB) — noise next to 288 GB of HBM
— and a training step takes 0.1-2 ms, so the loader, not the GPU, sets the epoch time.  `DeviceBatches` keeps the arrays in
HBM, draws the epoch permutation on the device and gathers each minibatch there: no workers, no per-step H2D, no host sync.

Data parallelism: every rank holds the full arrays and draws the SAME permutation (seed + epoch), then takes its contiguous
slice of each GLOBAL minibatch — the same partition `dist.ShardedBatches` makes of host batches — so `len()` and the
schedules agree on every rank and the union of the ranks' minibatches is exactly the single-process minibatch.
"""
import torch

from .dist import shard_bounds

__all__ = ['DeviceBatches']


def _map(f, x):
    return [_map(f, v) for v in x] if isinstance(x, (list, tuple)) else f(x)


class DeviceBatches:
    """Iterable of (x, y) minibatches gathered on `device` from whole-dataset tensors.

    x: tensor [N, ...] or (nested) list of such tensors (yielded as a list, as the reference's collaters do); y: tensor [N, ...].
    bs: PER-RANK batch size.  shuffle: new permutation every epoch (torch.Generator on the device, seed + epoch).
    rank / world: data-parallel slice of each global minibatch of bs*world samples.  The last minibatch may be ragged
    (the reference's loaders use drop_last=False; Learner scales its learning rate, Learner.py:503-505)."""

    def __init__(self, x, y, bs, shuffle=False, device=None, seed=0, rank=0, world=1):
        from .General.Core import default_device
        self.device = torch.device(device if device is not None else default_device())
        to_dev = lambda t: torch.as_tensor(t).to(self.device)
        self.x, self.y = _map(to_dev, x), to_dev(y)
        self.n = len(self.y)
        self.bs, self.shuffle, self.seed, self.rank, self.world = int(bs), shuffle, int(seed), int(rank), int(world)
        self.epoch = 0
        self._gen = None
        self.dp_info = None          # (rows of the last yielded shard that count, rows of its GLOBAL minibatch): Learner reads it

    def __len__(self):
        g = self.bs * self.world
        return (self.n + g - 1) // g

    def _perm(self):
        if not self.shuffle:
            return None
        if self._gen is None:
            self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self.seed + self.epoch)          # identical on every rank
        return torch.randperm(self.n, device=self.device, generator=self._gen)

    def __iter__(self):
        perm = self._perm()
        self.epoch += 1
        g = self.bs * self.world
        for b in range(len(self)):
            lo = b * g
            hi = min(lo + g, self.n)
            a, z, ghost = shard_bounds(hi - lo, self.rank, self.world)   # balanced contiguous cut, as dist.ShardedBatches
            a, z = lo + a, lo + z
            self.dp_info = (0 if ghost else z - a, hi - lo)
            if perm is None:
                take = lambda t: t[a:z]
            else:
                idx = perm[a:z]
                take = lambda t: t.index_select(0, idx)
            yield _map(take, self.x), take(self.y)
