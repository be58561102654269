"""CPU: the Trainer surface main.py consumes (reference src/main.py:62-82), driven with the
oracle-backed test double in place of the HIP engine."""
import logging
import os

import pytest
import torch

from helpers import OracleEngine, oracle_factory, reference_args

SMALL = dict(batch_size=4, epochs=2, print_freq=1, n_frames=4, n_slots=8, d_model=64, n_layers=1,
             train_clips=12, val_clips=8)


@pytest.fixture
def workdir(tmp_path, monkeypatch):
    src = tmp_path / "src"
    src.mkdir()
    monkeypatch.chdir(src)          # the reference runs from src/ and writes ../predict, ../checkpoint
    return tmp_path


def test_product_engine_refuses_cpu(workdir):
    """No silent fallback: without a HIP device (or without the .so) the real Trainer raises."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trainer import Trainer
    with pytest.raises(Exception):
        Trainer(reference_args(workdir / "exp", **SMALL))


def test_epoch_loop_like_main_worker(workdir, caplog):
    from trainer import Trainer
    args = reference_args(workdir / "exp", **SMALL)
    tr = Trainer(args, engine_factory=oracle_factory)
    assert os.path.isdir("../predict")                      # reference src/trainer.py:107-108
    assert hasattr(tr, "model") and tr.model.eval() is tr.model
    losses = []
    with caplog.at_level(logging.INFO):
        for epoch in range(args.epochs):                    # reference src/main.py:76-82
            tr.set_epoch(epoch)
            assert tr.epoch == epoch + 1
            tr.train()
            m = tr.validate()
            assert set(m) == {"loss"} and isinstance(m["loss"], float)
            losses.append(m["loss"])
            tr.save_checkpoint(m)
    assert losses[1] < losses[0]
    assert any("Epoch [1/2][1/3] load [" in r.message and "loss [" in r.message for r in caplog.records)
    assert os.path.exists("../checkpoint/002.pth") and os.path.exists("../checkpoint/latest.pth")
    assert tr.global_step == 6
    rows = open(os.path.join(args.path, "scalars.tsv")).read().splitlines()
    assert sum(r.startswith("train/gen loss GAN") for r in rows) == 6 and sum(r.startswith("val/loss") for r in rows) == 2


def test_checkpoint_schema_and_resume(workdir):
    from trainer import Trainer
    args = reference_args(workdir / "exp", **SMALL)
    tr = Trainer(args, engine_factory=oracle_factory)
    tr.set_epoch(0)
    tr.train()
    tr.save_checkpoint({"loss": 1.0})
    ck = torch.load("../checkpoint/001.pth", weights_only=True)
    assert {"epoch", "arch", "gridnet", "optimizer"} <= set(ck)          # SURVEY.md section 5 schema
    assert ck["arch"] == "CoordGridNet" and ck["epoch"] == 1
    # --resume restores weights, Adam moments, step count and epoch: the next step is bitwise identical
    a2 = reference_args(workdir / "exp2", resume="../checkpoint/latest.pth", **SMALL)
    tr2 = Trainer(a2, engine_factory=oracle_factory)
    assert tr2.epoch == 1 and tr2.engine.step_count == tr.engine.step_count
    assert torch.equal(tr2.engine.params, tr.engine.params)
    saved = tr.engine.params.clone()
    batch = next(iter(tr.train_loader))
    tr.engine.train_step(batch)
    tr2.engine.train_step(batch)
    assert torch.equal(tr2.engine.params, tr.engine.params)
    # --ckpt loads 'gridnet' (+ 'optimizer') at build time, reference src/trainer.py:85-92
    a3 = reference_args(workdir / "exp3", ckpt="../checkpoint/latest.pth", **SMALL)
    tr3 = Trainer(a3, engine_factory=oracle_factory)
    assert torch.equal(tr3.engine.params, saved)
    assert tr3.engine.step_count == ck["optimizer"]["step"]
    # arch mismatch is refused like reference src/trainer.py:407-408
    a4 = reference_args(workdir / "exp4", resume="../checkpoint/latest.pth", arch="GridNet", **SMALL)
    with pytest.raises(AssertionError):
        Trainer(a4, engine_factory=oracle_factory)


def test_flip_is_shared_and_consistent(workdir):
    """cx -> 1-cx on inputs and targets, decided by the shared-seed `random` stream (trainer.py:200-206)."""
    import random
    from trainer import Trainer
    tr = Trainer(reference_args(workdir / "exp", **SMALL), engine_factory=oracle_factory)
    batch = next(iter(tr.train_loader))
    random.seed(1)
    outs = [tr._flip(batch) for _ in range(8)]
    random.seed(1)
    draws = [random.random() < 0.5 for _ in range(8)]
    assert any(draws) and not all(draws)
    for o, flipped in zip(outs, draws):
        want = 1.0 - batch["slot_box"][..., 0] if flipped else batch["slot_box"][..., 0]
        assert torch.equal(o["slot_box"][..., 0], want)
        assert torch.equal(o["slot_box"][..., 1:], batch["slot_box"][..., 1:])
        assert torch.equal(o["slot_class"], batch["slot_class"])


def test_rollout_shapes(workdir):
    from trainer import Trainer
    tr = Trainer(reference_args(workdir / "exp", **SMALL), engine_factory=oracle_factory)
    batch = next(iter(tr.val_loader))
    c, b = tr.generate_sequence(batch["slot_class"], batch["slot_box"], steps=8)   # 8 steps, trainer.py:460
    assert c.shape == (4, 8, 8) and b.shape == (4, 8, 8, 4)
    assert int(c.min()) >= 0 and int(c.max()) < 20 and float(b.min()) > 0 and float(b.max()) < 1
