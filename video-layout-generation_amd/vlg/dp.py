"""Data-parallel gradient exchange for the flat gradient buffer (one process per GPU).

Replaces the two DistributedDataParallel wrappers and the scalar loss all-reduce of the
reference (reference src/trainer.py:113,115,256,381-386) with what the flat buffer makes
possible on an xGMI mesh:

* the gradient buffer is cut into a few LARGE contiguous buckets in the order backward
  finishes them (head + final norm first, then layer L-1 ... 0, embeddings last); each bucket
  is all-reduced (SUM) as soon as it is complete, asynchronously, so RCCL traffic over xGMI
  overlaps the backward of the earlier layers;
* the 4 loss scalars ride in the 4 floats after the last parameter, inside the first bucket -
  no separate 4-byte collective (reference trainer.py:256 issued one per step);
* the division by world size that DDP applies is folded into the Adam kernel's grad_scale.

The class only needs `torch.distributed` and a flat tensor, so it runs unchanged on CPU
tensors with the gloo backend (tests/test_dp_gloo.py) and on HIP tensors with RCCL
(backend "nccl" in PyTorch-ROCm).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def bucket_ranges(layout: "Dict[str, Tuple[int, Tuple[int, ...]]]", n_params: int, n_layers: int,
                  tail_extra: int = 4) -> List[Tuple[str, int, int]]:
    """[(tag, start, end)] in backward completion order.  `tail_extra` floats after the last
    parameter (the loss scalars) belong to the first bucket."""
    out = [("head", layout["lnf_g"][0], n_params + tail_extra)]
    for l in reversed(range(n_layers)):
        start = layout["l%d.ln1_g" % l][0]
        end = layout["l%d.ln1_g" % (l + 1)][0] if l + 1 < n_layers else layout["lnf_g"][0]
        out.append(("l%d" % l, start, end))
    out.append(("embed", 0, layout["l0.ln1_g"][0]))
    return out


class GradReducer:
    """Bucketed asynchronous all-reduce over one flat buffer."""

    def __init__(self, flat: torch.Tensor, buckets: List[Tuple[str, int, int]], group=None,
                 always_communicate: bool = False):
        self.flat, self.group = flat, group
        self.always = always_communicate      # run the collectives even at world size 1 (1-GPU RCCL rehearsal)
        self.buckets = {tag: (s, e) for tag, s, e in buckets}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pending: list = []
        covered = sorted((s, e) for _, s, e in buckets)
        pos = 0
        for s, e in covered:
            if s != pos:
                raise ValueError("buckets must tile the flat buffer without gaps (gap at %d)" % pos)
            pos = e
        if pos != flat.numel():
            raise ValueError("buckets cover %d floats, buffer has %d" % (pos, flat.numel()))

    @property
    def grad_scale(self) -> float:
        """Factor that turns the summed gradient into DDP's mean (reference trainer.py:113)."""
        return 1.0 / self.world

    def ready(self, tag: str) -> None:
        """Called by backward when every gradient of bucket `tag` has been written (stream-ordered)."""
        if self.world == 1 and not self.always:
            return
        s, e = self.buckets[tag]
        self.pending.append((tag, dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def wait(self, keep=()) -> None:
        """Make the current stream wait for every outstanding bucket (no host sync on RCCL), except the tags in
        `keep`, which stay in flight (the optimiser can update the finished ranges meanwhile)."""
        rest = []
        for tag, w in self.pending:
            if tag in keep:
                rest.append((tag, w))
            else:
                w.wait()
        self.pending = rest


def all_reduce_mean_(tensors, world: int, group=None) -> None:
    """Trainer.sync semantics (reference src/trainer.py:381-386): in-place SUM then / gpus."""
    for t in tensors:
        dist.all_reduce(t, group=group)
        t.div_(world)
