"""GPU: the real Trainer (HIP engine) driven the way reference src/main.py:62-82 drives it, side by
side with the same Trainer on the oracle-backed double: same clips, same flips, same Adam -> the
validation losses must agree to 1e-4."""
import random

import pytest
import torch

from helpers import oracle_factory, reference_args

pytestmark = pytest.mark.gpu

SMALL = dict(batch_size=4, epochs=2, print_freq=1, n_frames=8, n_slots=8, d_model=64, n_layers=2,
             train_clips=16, val_clips=8)


def run(tmp, tag, factory):
    from trainer import Trainer
    random.seed(1024)                                       # main.worker seeds the shared flip stream (main.py:57)
    args = reference_args(tmp / tag, **SMALL)
    tr = Trainer(args, engine_factory=factory)
    vals = []
    for epoch in range(args.epochs):
        tr.set_epoch(epoch)
        tr.train()
        vals.append(tr.validate()["loss"])
    return tr, vals


def test_trainer_on_gpu_matches_oracle_trainer(tmp_path, monkeypatch):
    src = tmp_path / "src"
    src.mkdir()
    monkeypatch.chdir(src)
    hip_tr, hip_vals = run(tmp_path, "hip", None)
    assert hip_tr.device.type == "cuda"
    cpu_tr, cpu_vals = run(tmp_path, "cpu", oracle_factory)
    for a, b in zip(hip_vals, cpu_vals):
        assert abs(a - b) <= 1e-4 * abs(b), (hip_vals, cpu_vals)
    assert hip_vals[1] < hip_vals[0]
    # checkpoint written by the HIP trainer restores into a fresh one bit for bit
    hip_tr.save_checkpoint({"loss": hip_vals[-1]})
    from trainer import Trainer
    again = Trainer(reference_args(tmp_path / "again", resume="../checkpoint/latest.pth", **SMALL))
    assert torch.equal(again.engine.params, hip_tr.engine.params)
    assert torch.equal(again.engine.exp_avg_sq, hip_tr.engine.exp_avg_sq)
    c, b = again.generate_sequence(*[next(iter(again.val_loader))[k] for k in ("slot_class", "slot_box")], steps=8)
    assert c.shape == (4, 8, 8) and b.shape == (4, 8, 8, 4) and bool(torch.isfinite(b).all())


def test_trainer_runs_the_reference_model(tmp_path, monkeypatch):
    """VLG_MODEL=gridnet: main.py's default --arch CoordGridNet trains the reference's own model and losses
    (reference src/trainer.py:193-258) on synthetic frame triplets through the same Trainer surface."""
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    monkeypatch.setenv("VLG_MODEL", "gridnet")
    monkeypatch.setenv("VLG_IMG_SIZE", "32")
    from trainer import Trainer
    random.seed(1024)
    args = reference_args(tmp_path / "exp", batch_size=2, epochs=2, print_freq=1, train_clips=8, val_clips=4, lr=2e-3)
    tr = Trainer(args)
    assert tr.image_mode and tr.device.type == "cuda"
    vals = []
    for epoch in range(args.epochs):
        tr.set_epoch(epoch)
        tr.train()
        vals.append(tr.validate()["loss"])
    assert vals[1] < vals[0]
    tr.save_checkpoint({"loss": vals[-1]})
    ck = torch.load("../checkpoint/latest.pth", weights_only=True)
    assert ck["arch"] == "CoordGridNet" and tuple(ck["gridnet"]["lateral_in.conv.0.conv.weight"].shape) == (32, 12, 3, 3)
    again = Trainer(reference_args(tmp_path / "again", resume="../checkpoint/latest.pth", batch_size=2, epochs=1,
                                   print_freq=1, train_clips=8, val_clips=4))
    assert torch.equal(again.engine.engine.net.params, tr.engine.engine.net.params)


@pytest.mark.parametrize("precision,tol", [("fp32x3", 1e-4), ("bf16", 3e-2)])
def test_trainer_precision_knob(tmp_path, monkeypatch, precision, tol):
    """VLG_PRECISION selects the projection mode behind the unchanged Trainer surface: fp32x3 (fp32 tensors, split
    bf16 MFMAs) must track the oracle-backed Trainer at the fp32 bar; bf16 (bf16 MFMA + bf16 activation storage) at its
    stated looser one; a checkpoint of the bf16 mode restores the master weights AND the bf16 shadow."""
    src = tmp_path / "src"
    src.mkdir()
    monkeypatch.chdir(src)
    monkeypatch.setenv("VLG_PRECISION", precision)
    hip_tr, hip_vals = run(tmp_path, "hip", None)
    assert hip_tr.engine.precision == precision
    monkeypatch.delenv("VLG_PRECISION")
    _, cpu_vals = run(tmp_path, "cpu", oracle_factory)
    for a, b in zip(hip_vals, cpu_vals):
        assert abs(a - b) <= tol * abs(b), (hip_vals, cpu_vals)
    assert hip_vals[1] < hip_vals[0]
    if precision == "bf16":
        monkeypatch.setenv("VLG_PRECISION", precision)
        hip_tr.save_checkpoint({"loss": hip_vals[-1]})
        from trainer import Trainer
        again = Trainer(reference_args(tmp_path / "again", resume="../checkpoint/latest.pth", **SMALL))
        assert torch.equal(again.engine.params, hip_tr.engine.params)
        assert torch.equal(again.engine.params_bf16, hip_tr.engine.params.to(torch.bfloat16))
        assert torch.equal(hip_tr.engine.params_bf16, hip_tr.engine.params.to(torch.bfloat16))


def test_trainer_reads_the_reference_dataset_layout(tmp_path, monkeypatch):
    """VLG_MODEL=gridnet with --train_dir / --val_dir pointing at frame triplets in the reference's Cityscapes layout
    (src/folder.py:14-46): Trainer feeds them through vlg/cityscapes.py, the frozen HED net supplies the edge channels
    (trainer.py:190-197) and an epoch trains; `VLG_WITH_VGG=1` adds the VGG19 term of CombinedLoss."""
    from test_cityscapes_cpu import write_tree
    root = tmp_path / "data"
    write_tree(str(root / "train"), "aachen", 7, list(range(0, 16)), hw=(32, 32), seed=1)     # 9 triplets
    write_tree(str(root / "val"), "bonn", 2, list(range(0, 10)), hw=(32, 32), seed=2)         # 3 triplets
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    monkeypatch.setenv("VLG_MODEL", "gridnet")
    monkeypatch.setenv("VLG_IMG_SIZE", "32")
    monkeypatch.setenv("VLG_WITH_VGG", "1")
    from trainer import Trainer
    random.seed(1024)
    args = reference_args(tmp_path / "exp", batch_size=2, epochs=2, print_freq=1, lr=2e-3,
                          train_dir=str(root / "train"), val_dir=str(root / "val"))
    tr = Trainer(args)
    assert tr.image_mode and tr.engine.on_disk and tr.engine.engine.hed is not None and tr.engine.engine.vgg is not None
    assert len(tr.train_loader) == 4 and len(tr.val_loader) == 1
    vals = []
    for epoch in range(args.epochs):
        tr.set_epoch(epoch)
        tr.train()
        vals.append(tr.validate()["loss"])
    assert all(v == v and v > 0 for v in vals) and vals[1] < vals[0], vals


def test_gridnet_mode_loads_a_reference_checkpoint(tmp_path, monkeypatch):
    """--ckpt as reference src/trainer.py:85-92 reads it: {'gridnet': state_dict, 'optimizer': torch.optim.Adam
    state_dict over model.parameters()}.  Built here with torch itself over parameters in the reference's registration
    order; the Trainer must continue EXACTLY where that optimiser stood: one more step on the HIP engine equals one more
    torch step fed the HIP gradients.  The checkpoint the Trainer writes loads back into a torch Adam."""
    from oracle import gridnet_spec as G
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    monkeypatch.setenv("VLG_MODEL", "gridnet")
    monkeypatch.setenv("VLG_IMG_SIZE", "32")
    shapes = G.param_shapes(10, coord=True)
    p0 = G.test_params(shapes, seed=8)
    tp = [torch.nn.Parameter(p0[k].clone()) for k in shapes]              # reference order = parameters() order
    opt = torch.optim.Adam(tp, lr=2e-3, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        for q in tp:
            q.grad = torch.randn(q.shape, generator=g) * 0.01
        opt.step()
    weights = {k: q.detach().clone() for k, q in zip(shapes, tp)}
    torch.save({"epoch": 3, "arch": "CoordGridNet", "gridnet": weights, "optimizer": opt.state_dict()}, str(tmp_path / "ref.pth"))
    from trainer import Trainer
    random.seed(1024)
    tr = Trainer(reference_args(tmp_path / "exp", batch_size=2, epochs=1, print_freq=1, train_clips=4, val_clips=2,
                                lr=2e-3, ckpt=str(tmp_path / "ref.pth")))
    eng = tr.engine.engine
    sd = eng.state_dict()
    assert all(torch.equal(sd[k], weights[k]) for k in shapes)
    assert eng.step_count == 2
    m = eng.net.unpack(eng.exp_avg)
    st = opt.state_dict()["state"]
    assert all(torch.equal(m[k], st[i]["exp_avg"]) for i, k in enumerate(shapes))
    # one step on the HIP engine == one torch.optim step on the same gradients
    batch = {k: v.to(tr.device) for k, v in next(iter(tr.train_loader)).items()}
    eng.train_step(batch)
    grads = eng.net.named_grads()
    for k, q in zip(shapes, tp):
        q.grad = grads[k].clone()
    opt.step()
    after = eng.state_dict()
    for k, q in zip(shapes, tp):
        assert torch.allclose(after[k], q.detach(), rtol=1e-5, atol=1e-7), k
    # what the Trainer saves is a torch.optim.Adam state_dict again
    tr.save_checkpoint({"loss": 0.0})
    ck = torch.load("../checkpoint/latest.pth", weights_only=True)
    opt2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(s)) for s in shapes.values()], lr=1.0)
    opt2.load_state_dict(ck["optimizer"])
    assert int(opt2.state_dict()["state"][0]["step"]) == 3 and opt2.state_dict()["param_groups"][0]["betas"][0] == 0.5
    # a state of the wrong architecture is refused with a clear message, not a KeyError
    bad = dict(ck["optimizer"])
    bad["param_groups"] = [dict(bad["param_groups"][0], params=list(range(5)))]
    with pytest.raises(ValueError):
        eng.load_optimizer(bad)
