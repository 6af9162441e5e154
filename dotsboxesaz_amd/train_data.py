"""Training DATA path on the GPU: the consumer side of the self-play rows.

Mirrors, on top of the C ABI (`dbaz_dataset_*`, `dbaz_symmetry_apply`; HIP kernels in csrc/replay.hip):

  * `ReplayStore`          the `fresh` -> `data` bookkeeping of coach.train_nn (coach.py:56-66): rows of a
                           generation get `training` = 1 / -1 by `df.sample(frac=train_split)`; the rows
                           themselves STAY IN HBM as packed replay rows (engine.replay_rows_dev() or the RCCL
                           all-gathered tensor of `self_play.all_gather_rows`)
  * `ReplayDataset`        utils.HDFStoreDataset (utils/utils.py:61-91): where-clause on the generation,
                           training flag, `df.sample(min(n_samples, n))`, optional pos_average; `__len__` /
                           `__getitem__` as the reference; `.loader()` is the device fast path
  * `SymmetriesGenerator`  dots_boxes_nn.SymmetriesGenerator (dots_boxes/dots_boxes_nn.py:11-58): callable
                           `(boards, policies) -> (boards, policies)` drawing `random.randint(0, 7)`
  * `DeviceLoader`         DataLoader(shuffle, drop_last) + symmetries of the train loop (nn.py:186-216):
                           index batches come from torch's own samplers (same RNG stream as the reference's
                           loader), gather + transform run in ONE HIP kernel per batch

The host RNG calls are the reference's (`np.random.choice` behind DataFrame.sample, torch's RandomSampler,
`random.randint`), so a run seeded like the reference selects the same rows, batches and transforms.
"""
import random

import numpy as np


def _sample_locs(n, size):
    """DataFrame.sample(n=size) without replacement: pandas draws np.random.choice(n, size, replace=False)."""
    return np.random.choice(n, size=size, replace=False)


class ReplayStore:
    """Per-generation packed replay rows on one GPU plus their `training` flags."""

    def __init__(self, engine):
        self.engine = engine
        self.chunks = []  # dicts: generation, rows (device), n, train_locs, val_locs

    def add_generation(self, generation, rows, train_split=0.9):
        """rows: torch uint8 CUDA tensor [n, row_bytes] (kept alive here) or the tuple of
        engine.replay_rows_dev() (only valid until the engine's next call of it).
        coach.py:59-63: train = sample(frac), val = the rest in original order."""
        n = rows[1] if isinstance(rows, tuple) else int(rows.shape[0])
        size = int(round(train_split * n))
        train_locs = _sample_locs(n, size) if n else np.zeros(0, dtype=np.int64)
        mask = np.ones(n, dtype=bool)
        mask[train_locs] = False
        self.chunks.append(dict(generation=int(generation), rows=rows, n=n, train_locs=train_locs.astype(np.int32),
                                val_locs=np.nonzero(mask)[0].astype(np.int32)))

    def dataset(self, train=True, min_generation=0, n_samples=int(1e12), pos_average=False, slot=None):
        """slot: which of the engine's resident datasets to build into (default: 0 for train, 1 for validation)."""
        return ReplayDataset(self, train, min_generation, n_samples, pos_average, slot)

    def drop_before(self, generation):
        """Releases the rows of generations older than `generation`.  The reference's HDF file only grows; here the window lives in
        HBM, and the window's lower edge (coach.py:148-149, train.window_where) never moves back, so what lies below it is dead:
        a long run keeps at most the ~21 generations of the widest window on the device.  Returns the number of generations dropped."""
        keep = [c for c in self.chunks if c["generation"] >= generation]
        dropped = len(self.chunks) - len(keep)
        self.chunks = keep
        return dropped


class ReplayDataset:
    """utils.HDFStoreDataset over rows in HBM.  Building it runs the HIP dataset kernels; an engine
    handle keeps up to four datasets resident (`slot`); building into a slot replaces its content."""

    def __init__(self, store, train=True, min_generation=0, n_samples=int(1e12), pos_average=False, slot=None):
        e = self.engine = store.engine
        self.slot = (0 if train else 1) if slot is None else int(slot)
        e.dataset_select(self.slot)
        chunks = [c for c in store.chunks if c["generation"] >= min_generation]
        cand = [(ci, loc) for ci, c in enumerate(chunks) for loc in (c["train_locs"] if train else c["val_locs"])]
        take = _sample_locs(len(cand), min(int(n_samples), len(cand))) if cand else []
        # stage chunk by chunk (one gather kernel per generation), then hand the dataset order over
        # as a permutation of the staged rows
        per_chunk = [[] for _ in chunks]
        where = []
        for t in take:
            ci, loc = cand[int(t)]
            where.append((ci, len(per_chunk[ci])))
            per_chunk[ci].append(loc)
        e.dataset_begin()
        base, bases = 0, []
        for ci, c in enumerate(chunks):
            bases.append(base)
            if per_chunk[ci]:
                e.dataset_add_rows(c["rows"], np.asarray(per_chunk[ci], dtype=np.int32))
                base += len(per_chunk[ci])
        order = np.asarray([bases[ci] + k for ci, k in where], dtype=np.int32)
        self.n = e.dataset_finish(pos_average, order if len(order) else None)
        self.pos_average = bool(pos_average)
        self._host = None

    def __len__(self):
        return self.n

    def _arrays(self):
        if self._host is None:
            e = self.engine
            e.dataset_select(self.slot)
            x, pi, z = e.dataset_fetch()
            self._host = (x.astype(np.float32).reshape(-1, 3, e.H, e.W), pi, z)
        return self._host

    def __getitem__(self, index):
        f, p, v = self._arrays()
        return f[index], p[index], np.asarray([v[index]])

    def loader(self, batch_size, shuffle=True, drop_last=True, symmetries=None):
        return DeviceLoader(self, batch_size, shuffle, drop_last, symmetries)


class _Indices:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class DeviceLoader:
    """for boards, pi, z in loader: float32 CUDA tensors [B,3,H,W], [B,A], [B,1]."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=True, symmetries=None):
        from torch.utils import data
        self.ds = dataset
        self.symmetries = symmetries
        self._idx = data.DataLoader(_Indices(len(dataset)), batch_size, shuffle=shuffle, drop_last=drop_last)

    def __len__(self):
        return len(self._idx)

    def __iter__(self):
        e = self.ds.engine
        for idx in self._idx:
            sym = self.symmetries.draw() if self.symmetries is not None else 0
            e.dataset_select(self.ds.slot)
            yield e.dataset_batch(idx.numpy(), sym)


class SymmetriesGenerator:
    """Drop-in for params.nn.train_params.symmetries (dots_boxes_nn.py:11-58) on CUDA tensors."""

    def __init__(self, engine):
        self.engine = engine

    def draw(self):
        """The reference draws random.randint(0, 7) per batch.  Its rotations (4..7) fail on a
        non-square board; here they degrade to the flip part of the transform."""
        sym = random.randint(0, 7)
        return sym if self.engine.H == self.engine.W else sym & 3

    def __call__(self, boards, policies):
        sym = self.draw()
        if sym == 0:
            return boards, policies
        return self.engine.symmetry_apply(sym, boards.contiguous(), policies.contiguous())

    forward = __call__
