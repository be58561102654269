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


def test_bucketed_variable_n_training(tmp_path, monkeypatch):
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    from trainer import Trainer
    from vlg.spec import param_shapes
    random.seed(1024)
    args = reference_args(tmp_path / "exp", batch_size=4, epochs=1, print_freq=1, n_frames=8, n_slots=32, d_model=64,
                          n_layers=2, train_clips=64, val_clips=16, variable_n=1)
    tr = Trainer(args)
    shapes = set()
    p = O.init_params(param_shapes(tr.cfg), seed=1024)
    checked = 0
    for batch in tr.train_loader:
        n = batch["slot_class"].shape[2]
        shapes.add(n)
        assert n % 8 == 0 and n <= 32
        dev_batch = {k: v.to(tr.device) for k, v in batch.items()}
        loss = tr.engine.forward_backward(dev_batch)
        if checked < 3:                                   # oracle comparison on the first few buckets (initial weights)
            parts, grads = O.loss_and_grads(p, batch, tr.cfg.n_layers)
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
