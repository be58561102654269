"""GPU: BASELINE.json configs[4] shape - variable-N (padded / masked) clips with N-bucketed batching
through the real Trainer: every bucket size runs through the same engine workspace, padded slots
contribute neither loss nor gradient, and the result matches the oracle on the same bucketed batch."""
import random

import pytest
import torch

from conftest import assert_close
from helpers import reference_args
from oracle import layout_spec as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("attention", ["slot", "clip"])
def test_bucketed_variable_n_training(tmp_path, monkeypatch, attention):
    """BASELINE configs[4] through the real Trainer; attention = "clip" (VLG_ATTENTION): the per-clip option on bucketed batches -
    padded slots are masked as keys (the trainer builds the engine with padded_slots = variable_n)."""
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    monkeypatch.setenv("VLG_ATTENTION", attention)
    from trainer import Trainer
    from vlg.spec import param_shapes
    random.seed(1024)
    args = reference_args(tmp_path / "exp", batch_size=4, epochs=1, print_freq=1, n_frames=8, n_slots=32, d_model=64,
                          n_layers=2, train_clips=64, val_clips=16, variable_n=1)
    tr = Trainer(args)
    assert tr.cfg.attention == attention and tr.engine.padded_slots
    shapes = set()
    p = O.init_params(param_shapes(tr.cfg), seed=1024)
    checked = 0
    for batch in tr.train_loader:
        n = batch["slot_class"].shape[2]
        shapes.add(n)
        assert n % 8 == 0 and n <= 32
        assert batch["slot_class"].is_cuda                # the Trainer's loaders deliver device batches (pinned prefetch)
        dev_batch = {k: v.to(tr.device) for k, v in batch.items()}
        batch = {k: v.cpu() for k, v in batch.items()}    # the oracle's copy
        loss = tr.engine.forward_backward(dev_batch)
        if checked < 3:                                   # oracle comparison on the first few buckets (initial weights)
            parts, grads = O.loss_and_grads(p, batch, tr.cfg.n_layers, attention=attention)
            assert_close(loss, torch.tensor(parts), rtol=1e-4, atol=1e-6, what="loss N=%d" % n)
            for name in ("cls_emb", "l0.qkv_w", "l1.ff2_w", "head_w"):
                g, w = tr.engine.named_grads()[name], grads[name]
                s = max(float(w.abs().max()), 1e-6)
                assert_close(g / s, w / s, rtol=1e-4, atol=2e-5, what="grad %s N=%d" % (name, n))
            checked += 1
        # padded slots: zero gradient at the head outputs
        B, T = batch["slot_class"].shape[:2]
        dout = tr.engine.dout[:B * T * n].view(B, n, T, -1).permute(0, 2, 1, 3)
        pad = (batch["valid"] == 0).to(tr.device)
        assert float(dout[pad].abs().max()) == 0.0 if bool(pad.any()) else True
    assert len(shapes) >= 3, shapes                       # several bucket sizes went through one workspace
    tr.set_epoch(0)
    tr.train()
    m = tr.validate()
    assert m["loss"] > 0 and m["loss"] == m["loss"]


@pytest.mark.parametrize("bucketed", [False, True])
def test_pinned_prefetch_delivers_the_same_batches(dev, bucketed):
    """vlg.data.device_prefetch (pinned staging, next batch's H2D on a side stream while the current one is consumed,
    three rotating slots) must hand over the same bytes in the same order as plain synchronous copies - with a
    consumer that keeps the device busy on every batch, over more batches than there are slots, and (bucketed) with
    the batch shape changing between buckets."""
    from vlg.data import BATCH_KEYS, BucketedClipLoader, ClipLoader, synthetic_clips
    clips = synthetic_clips(96, T=8, N=16, seed=3, variable_n=bucketed, min_valid=4)
    cls = BucketedClipLoader if bucketed else ClipLoader
    kw = dict(batch=4, rank=1, world=2, seed=9, shuffle=True)
    plain = cls(clips, device=dev, prefetch=False, **kw)
    fast = cls(clips, device=dev, prefetch=True, **kw)
    cpu = cls(clips, **kw)
    for ld in (plain, fast, cpu):
        ld.set_epoch(2)
    assert len(plain) == len(fast) == len(cpu) > 5
    busy = torch.randn(2048, 2048, device=dev)
    got = []
    for b in fast:
        busy = busy @ busy * 1e-3                                   # the consumer's stream is never idle
        got.append({k: b[k].clone() for k in BATCH_KEYS})           # slots are reused: keep a copy, stream-ordered
    torch.cuda.synchronize()
    want = list(plain)
    ref = list(cpu)
    assert len(got) == len(want) == len(ref)
    for g, w, r in zip(got, want, ref):
        for k in BATCH_KEYS:
            assert g[k].is_cuda and g[k].is_contiguous() and g[k].shape == w[k].shape
            assert torch.equal(g[k], w[k]) and torch.equal(g[k].cpu(), r[k]), k


def test_prefetched_batches_alias_three_slots(dev):
    """The lifetime contract of vlg.data.device_prefetch: a yielded batch aliases one of three rotating
    device slots, so a batch HELD across more than two iterations is overwritten - and prefetch=False hands out fresh
    tensors that are not."""
    from vlg.data import ClipLoader, synthetic_clips
    clips = synthetic_clips(64, T=4, N=8, seed=5)
    kw = dict(batch=4, seed=9, shuffle=False)
    held = list(ClipLoader(clips, device=dev, prefetch=True, **kw))[:4]
    torch.cuda.synchronize()
    fresh = list(ClipLoader(clips, device=dev, prefetch=False, **kw))[:4]
    assert held[0]["slot_box"].data_ptr() == held[3]["slot_box"].data_ptr()        # slot 0 reused by batch 3
    assert len({b["slot_box"].data_ptr() for b in held[:3]}) == 3
    assert not torch.equal(held[0]["slot_box"], fresh[0]["slot_box"])               # ... and its contents are gone
    assert len({b["slot_box"].data_ptr() for b in fresh}) == 4
