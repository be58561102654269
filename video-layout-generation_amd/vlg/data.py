"""Clip feeds for the layout-token step: synthetic clips and an N-bucketed loader.

Replaces the reference's Cityscapes triplet pipeline (reference src/folder.py:14-46,85-104,
src/data.py:28-52) for the configurations BASELINE.json names; there is no dataset in this
environment, so every clip is synthetic (Spec N) and says so.

A clip is T+1 frames of N object slots; inputs are frames 0..T-1, targets frames 1..T
(the reference likewise predicts the third frame of a triplet from the first two,
src/folder.py:34-35).  Sharding follows torch's DistributedSampler as used by the reference
(src/trainer.py:145-152): rank r takes clips r, r+world, r+2*world, ... of the epoch's
permutation, and set_epoch() reseeds that permutation (src/trainer.py:158-162).
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence

import torch

BATCH_KEYS = ("slot_class", "slot_box", "tgt_class", "tgt_box", "valid")


def synthetic_clips(n_clips: int, T: int, N: int, n_classes: int = 20, seed: int = 1024,
                    variable_n: bool = False, min_valid: int = 8) -> Dict[str, torch.Tensor]:
    """CPU tensors for n_clips clips (leading dim = clip).  Seed default = reference src/main.py:121."""
    g = torch.Generator().manual_seed(seed)
    cls = torch.randint(0, n_classes, (n_clips, T + 1, N), generator=g, dtype=torch.int64)
    box = torch.rand((n_clips, T + 1, N, 4), generator=g, dtype=torch.float32) * 0.9 + 0.05
    c = box[..., :2]
    wh = torch.minimum(box[..., 2:], 2 * torch.minimum(c, 1 - c))      # keep boxes inside the unit square
    box = torch.cat([c, wh], dim=-1).contiguous()
    valid = torch.ones((n_clips, T, N), dtype=torch.float32)
    n_valid = torch.full((n_clips,), N, dtype=torch.int64)
    if variable_n:
        n_valid = torch.randint(min(min_valid, N), N + 1, (n_clips,), generator=g)
        valid = (torch.arange(N)[None, None, :] < n_valid[:, None, None]).float().expand(n_clips, T, N).contiguous()
        cls_in = cls[:, :T].clone()
        cls_in[valid == 0] = n_classes                                  # reserved id marks a padded slot
        cls = torch.cat([cls_in, cls[:, T:]], dim=1)
    return {"slot_class": cls[:, :T].contiguous(), "slot_box": box[:, :T].contiguous(),
            "tgt_class": cls[:, 1:].clamp(max=n_classes - 1).contiguous(),
            "tgt_box": box[:, 1:].contiguous(), "valid": valid, "n_valid": n_valid}


def to_device(batch: Dict[str, torch.Tensor], device: torch.device, keys=BATCH_KEYS) -> Dict[str, torch.Tensor]:
    return {k: batch[k].to(device, non_blocking=True).contiguous() for k in keys}


def device_prefetch(cpu_batches, device: torch.device, keys=BATCH_KEYS) -> Iterator[Dict[str, torch.Tensor]]:
    """Pinned-memory staging + one-batch-ahead H2D on a side stream (what the reference gets from DataLoader(pin_memory=True)
    + .cuda(non_blocking=True), src/trainer.py:149-152,184-187): while the step consumes batch i, batch i+1 is gathered
    into a pinned host buffer and copied to HBM on a second HIP stream; the consumer's stream only waits for that copy's
    event.  Three rotating slots of pinned + device buffers: a slot is rewritten only after the copy that read its pinned
    half has finished (host-side event wait) and after the step that used its device half has been enqueued (stream-side
    event wait).  Yields the SAME bytes in the SAME order as the plain path (tests/test_hip_variable_n.py).

    LIFETIME CONTRACT: a yielded batch aliases one of THREE rotating device slots - it is valid until the consumer asks
    for the batch after the next one (i.e. for two iterations); anything that keeps batches longer (list(loader), a held
    validation batch, a debugging hook) must clone them or build the loader with prefetch=False, which returns fresh
    tensors (tests/test_hip_variable_n.py::test_prefetched_batches_alias_three_slots pins this)."""
    side = torch.cuda.Stream(device=device)
    slots = 3
    bufs: list = [dict() for _ in range(slots)]          # slot -> {shape signature: (pinned dict, device dict)}
    copied: list = [None] * slots                        # H2D of the slot finished (side stream)
    done: list = [None] * slots                          # consumer finished with the slot's device buffers (its stream)

    def stage(i: int, b: Dict[str, torch.Tensor]) -> tuple:
        s = i % slots
        sig = tuple((k, tuple(b[k].shape), b[k].dtype) for k in keys)
        if sig not in bufs[s]:
            bufs[s][sig] = ({k: torch.empty(b[k].shape, dtype=b[k].dtype, pin_memory=True) for k in keys},
                            {k: torch.empty(b[k].shape, dtype=b[k].dtype, device=device) for k in keys})
        pinned, dev = bufs[s][sig]
        if copied[s] is not None:
            copied[s].synchronize()                      # the previous copy out of this pinned slot is complete
        for k in keys:
            pinned[k].copy_(b[k])
        if done[s] is not None:
            side.wait_event(done[s])                     # the step that read this device slot has been enqueued and finished
        with torch.cuda.stream(side):
            for k in keys:
                dev[k].copy_(pinned[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        copied[s] = ev
        return s, dev

    it = iter(cpu_batches)
    i = 0
    first = next(it, None)
    staged = stage(0, first) if first is not None else None
    while staged is not None:
        cur, dev = staged
        nb = next(it, None)
        i += 1
        staged = stage(i, nb) if nb is not None else None        # batch i+1's H2D is in flight while batch i is consumed
        torch.cuda.current_stream(device).wait_event(copied[cur])
        yield dict(dev)
        e = torch.cuda.Event()
        e.record(torch.cuda.current_stream(device))
        done[cur] = e


def shard_indices(n_items: int, rank: int, world: int, epoch: int, seed: int, shuffle: bool = True) -> List[int]:
    """DistributedSampler arithmetic: pad the permutation to a multiple of world by wrapping,
    then rank r takes every world-th index starting at r."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n_items, generator=g).tolist()
    else:
        order = list(range(n_items))
    total = (n_items + world - 1) // world * world
    order += order[: total - n_items]
    return order[rank:total:world]


class ClipLoader:
    """Fixed-shape loader: every batch is (B, T, N) clips from one synthetic pool."""

    def __init__(self, clips: Dict[str, torch.Tensor], batch: int, rank: int = 0, world: int = 1,
                 seed: int = 1024, shuffle: bool = True, device: Optional[torch.device] = None, keys=BATCH_KEYS,
                 prefetch: bool = True):
        """device = None: CPU batches.  device = a HIP device: device batches, staged through pinned memory one batch
        ahead on a side stream (device_prefetch) unless prefetch=False (plain synchronous copies).  Prefetched batches
        alias three rotating device slots: a batch stays valid for two iterations (see device_prefetch); consumers that
        retain batches pass prefetch=False."""
        self.clips, self.batch, self.rank, self.world = clips, batch, rank, world
        self.seed, self.shuffle, self.device, self.epoch = seed, shuffle, device, 0
        self.keys = tuple(keys)
        self.prefetch = prefetch
        self.n = clips[self.keys[0]].shape[0]

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _indices(self) -> List[int]:
        return shard_indices(self.n, self.rank, self.world, self.epoch, self.seed, self.shuffle)

    def __len__(self) -> int:
        return len(self._indices()) // self.batch          # drop the ragged tail: shapes stay static

    def _cpu_batches(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = self._indices()
        for i in range(len(self)):
            sel = torch.tensor(idx[i * self.batch:(i + 1) * self.batch])
            yield {k: self.clips[k][sel] for k in self.keys}

    def _deliver(self, cpu_batches) -> Iterator[Dict[str, torch.Tensor]]:
        if self.device is None:
            return cpu_batches
        if self.prefetch and self.device.type == "cuda":
            return device_prefetch(cpu_batches, self.device, self.keys)
        return (to_device(b, self.device, self.keys) for b in cpu_batches)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        return iter(self._deliver(self._cpu_batches()))


class BucketedClipLoader(ClipLoader):
    """Variable-N clips, bucketed: clips are grouped by their valid-slot count rounded up to a
    multiple of `bucket`, each batch comes from one bucket and is cropped to that bucket's N, so
    padded slots cost nothing beyond the bucket granularity.  Batches are dealt to ranks
    round-robin in descending-N order so every rank sees the same bucket (same token count) at
    the same step - otherwise the step time is the slowest rank's."""

    def __init__(self, clips, batch, bucket: int = 8, **kw):
        super().__init__(clips, batch, **kw)
        self.bucket = bucket
        N = clips["slot_class"].shape[2]
        nv = clips["n_valid"].clamp(min=1)
        self.bucket_n = ((nv + bucket - 1) // bucket * bucket).clamp(max=N)

    def _batches(self) -> List[tuple]:
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        out = []
        for bn in sorted(set(self.bucket_n.tolist()), reverse=True):
            members = torch.nonzero(self.bucket_n == bn).flatten()
            if self.shuffle:
                members = members[torch.randperm(len(members), generator=g)]
            per_step = self.batch * self.world
            for i in range(len(members) // per_step):
                chunk = members[i * per_step:(i + 1) * per_step]
                out.append((bn, chunk[self.rank::self.world]))
        return out

    def __len__(self) -> int:
        return len(self._batches())

    def _cpu_batches(self):
        for bn, sel in self._batches():
            b = {}
            for k in BATCH_KEYS:
                t = self.clips[k][sel]
                b[k] = t[:, :, :bn].contiguous()
            yield b
