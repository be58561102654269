"""CPU: clip feeds - synthetic generator, DistributedSampler arithmetic, N-bucketed batching."""
import torch

from oracle import layout_spec as O
from vlg.data import BucketedClipLoader, ClipLoader, shard_indices, synthetic_clips


def test_generator_matches_oracle_spec():
    for var in (False, True):
        a = synthetic_clips(6, 4, 16, seed=11, variable_n=var, min_valid=3)
        b = O.synthetic_batch(6, 4, 16, seed=11, variable_n=var, min_valid=3)
        for k in b:
            assert torch.equal(a[k], b[k]), k
    c = synthetic_clips(8, 4, 16, seed=2, variable_n=True, min_valid=3)
    pad = c["valid"] == 0
    assert pad.any() and bool((c["slot_class"][pad] == 20).all())          # reserved id marks padded slots
    box = c["slot_box"]
    assert float((box[..., 0] - box[..., 2] / 2).min()) >= -1e-6 and float((box[..., 0] + box[..., 2] / 2).max()) <= 1 + 1e-6


def test_shard_indices_is_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(37))
    for world in (1, 2, 8):
        for epoch in (0, 3):
            for rank in range(world):
                ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=1024)
                ref.set_epoch(epoch)
                assert shard_indices(37, rank, world, epoch, 1024) == list(ref)


def test_ranks_partition_the_epoch():
    clips = synthetic_clips(32, 4, 8, seed=1)
    seen = []
    for rank in range(4):
        ld = ClipLoader(clips, batch=2, rank=rank, world=4, seed=7)
        ld.set_epoch(1)
        assert len(ld) == 4
        for b in ld:
            assert b["slot_class"].shape == (2, 4, 8)
            seen += [tuple(x.flatten().tolist()) for x in b["slot_class"]]
    assert len(seen) == 32 and len(set(seen)) == 32


def test_bucketed_loader_equal_tokens_per_rank_and_no_lost_valid_slots():
    clips = synthetic_clips(256, 4, 64, seed=5, variable_n=True, min_valid=8)
    per_rank = []
    for rank in range(2):
        ld = BucketedClipLoader(clips, batch=4, bucket=8, rank=rank, world=2, seed=3)
        shapes = []
        for b in ld:
            n = b["slot_class"].shape[2]
            assert n % 8 == 0 and b["slot_class"].shape[:2] == (4, 4)
            assert float(b["valid"][:, :, n - 8:].sum()) > 0              # bucket is tight to 8 slots
            shapes.append(n)
        per_rank.append(shapes)
    assert per_rank[0] == per_rank[1] and len(set(per_rank[0])) > 1      # same N at the same step on every rank
    # cropping never drops a valid slot
    ld = BucketedClipLoader(clips, batch=4, bucket=8, seed=3)
    tot = sum(float(b["valid"].sum()) for b in ld)
    used = sum(len(sel) for _, sel in ld._batches())
    assert tot > 0 and used <= 256
