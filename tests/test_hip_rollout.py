"""GPU: the autoregressive rollouts behind Trainer.generate_sequence / eval_generate_sequence (reference
src/trainer.py:429-476, the entry reference src/main.py:64-67 calls).

Pixel model (VLG_MODEL=gridnet): vlg.image_engine.FrameRollout on the HIP nets vs oracle/rollout_spec.py (the reference's
loop restated line by line over the GridNet / HED restatements that are pinned bit for bit to the reference modules).
Token model: Trainer.generate_sequence vs oracle/layout_spec.py.

An argmax feeds every step back into the next, so the comparison is teacher-forced: step i of the restatement is
evaluated on the HIP rollout's own state before step i, then predicted frames must agree to 1e-4 and predicted class
ids everywhere except where the two best logits are closer than the fp32 forward's own error."""
import os
import random

import numpy as np
import pytest
import torch

from helpers import reference_args
from oracle import gridnet_spec as G
from oracle import hned_spec as HS
from oracle import layout_spec as O
from oracle import rollout_spec as R

pytestmark = pytest.mark.gpu


def _ids_agree(got_ids, logits, what):
    """got_ids (b,1,H,W) float vs argmax of the restatement's logits (b,C,H,W): equal wherever the decision is not a
    near-tie (top-2 margin > 1e-4 of the logit scale)."""
    top2 = logits.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    want = logits.argmax(dim=1)
    clear = margin > 1e-4 * float(logits.abs().max())
    got = got_ids[:, 0].long()
    assert bool((got[clear] == want[clear]).all()), what
    assert float(clear.float().mean()) > 0.9, what           # the check is not vacuous (random-weight logits sit close together)
    assert bool(((got >= 0) & (got < logits.shape[1])).all())


@pytest.mark.parametrize("arch", ["GridNet", "CoordGridNet"])
def test_frame_rollout_matches_the_restated_loop(dev, arch):
    from vlg.hned import HNEDHIP
    from vlg.image_engine import FrameRollout, ImageEngine
    b, H, W, filt, steps = 2, 32, 48, (8, 16, 24), 3
    coord = arch == "CoordGridNet"
    eng = ImageEngine(4, 16, 16, dev, arch=arch, filters=filt)           # training engine of another size: weights only
    p = G.test_params(G.param_shapes(10, filt, coord=coord), seed=6)
    eng.load_state_dict(p)
    hp = HS.test_params(2)
    hed = HNEDHIP(1, 16, 16, dev)
    hed.load_state_dict(hp)
    g = torch.Generator().manual_seed(12)
    img1, img2 = torch.randn(b, 3, H, W, generator=g), torch.randn(b, 3, H, W, generator=g)
    seg1 = torch.randint(0, 20, (b, 1, H, W), generator=g).float()
    seg2 = torch.randint(0, 20, (b, 1, H, W), generator=g).float()
    ro = FrameRollout(eng, hed, b, H, W)
    pq = ro.run(img1.to(dev), img2.to(dev), seg1.to(dev), seg2.to(dev), steps=steps)
    pimg, qseg = pq[0].cpu(), pq[1].cpu()
    assert tuple(pimg.shape) == (b, 3 * (steps + 2), H, W) and tuple(qseg.shape) == (b, steps + 2, H, W)
    assert torch.equal(pimg[:, :3], img1) and torch.equal(pimg[:, 3:6], img2)            # trainer.py:455-458, 470
    assert torch.equal(qseg[:, 0:1], seg1) and torch.equal(qseg[:, 1:2], seg2)
    with torch.no_grad():
        for i in range(steps):
            ia, ib = pimg[:, 3 * i:3 * i + 3], pimg[:, 3 * i + 3:3 * i + 6]
            sa, sb = qseg[:, i:i + 1], qseg[:, i + 1:i + 2]
            logits, img_next, _ = R.step(p, hp, coord, ia, ib, sa, sb)
            got = pimg[:, 3 * i + 6:3 * i + 9]
            scale = float(img_next.abs().max())
            assert float((got - img_next).abs().max()) <= 1e-4 * scale, ("frame", i)
            _ids_agree(qseg[:, i + 2:i + 3], logits, "segmentation ids of step %d" % i)
        # and the un-forced loop agrees on its first prediction (no argmax has been fed back yet)
        p_w, q_w = R.generate_sequence(p, hp, coord, img1, img2, seg1, seg2, steps=1)
    assert float((pimg[:, 6:9] - p_w[:, 6:9]).abs().max()) <= 1e-4 * float(p_w[:, 6:9].abs().max())
    # the twin reads the training engine's weights: an update there is seen here without a copy
    eng.net.params.mul_(0.5)
    p2, _ = ro.run(img1.to(dev), img2.to(dev), seg1.to(dev), seg2.to(dev), steps=1)
    assert float((p2[:, 6:9].cpu() - pimg[:, 6:9]).abs().max()) > 1e-3


def test_argmax_kernel_takes_the_first_maximum(dev):
    from vlg import hip
    x = torch.randn(3, 20, 7 * 9, generator=torch.Generator().manual_seed(0))
    x[0, 4, :10] = 9.0
    x[0, 11, :10] = 9.0                                                   # exact tie: torch.argmax returns the first
    out = torch.empty(3, 1, 63, device=dev)
    hip.call("vlg_argmax_nchw", x.to(dev).data_ptr(), out.data_ptr(), 3, 20, 63, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out.cpu()[:, 0].long(), x.argmax(dim=1))


def _write_inputs(tmp, size, seg_size):
    from PIL import Image
    rng = np.random.RandomState(5)
    paths = {}
    for name in ("img1", "img2"):
        a = rng.randint(0, 256, size=(size, size, 3), dtype=np.uint8)
        Image.fromarray(a, "RGB").save(str(tmp / (name + ".png")))
        paths[name] = (str(tmp / (name + ".png")), a)
    for name in ("seg1", "seg2"):
        a = rng.randint(0, 20, size=(seg_size, seg_size), dtype=np.uint8)
        Image.fromarray(a, "L").save(str(tmp / (name + ".png")))
        paths[name] = (str(tmp / (name + ".png")), a)
    return paths


def test_eval_generate_sequence_entry(tmp_path, monkeypatch, dev):
    """main.py:64-67: trainer.model.eval(); trainer.eval_generate_sequence(img1, img2, seg1, seg2) with four FILE paths
    (trainer.py:429-451) -> 8-step rollout, two .npy files in ../predict (trainer.py:474-476)."""
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    monkeypatch.setenv("VLG_MODEL", "gridnet")
    monkeypatch.setenv("VLG_IMG_SIZE", "32")
    hp = HS.test_params(3)
    torch.save({"generator": hp}, str(tmp_path / "hed.pth"))              # the authors' file keeps it under 'generator' (:99)
    monkeypatch.setenv("VLG_HED_CKPT", str(tmp_path / "hed.pth"))
    from trainer import Trainer
    random.seed(1024)
    files = _write_inputs(tmp_path, 32, 64)
    tr = Trainer(reference_args(tmp_path / "exp", batch_size=2, epochs=1, print_freq=1, train_clips=4, val_clips=2,
                                img1=files["img1"][0], img2=files["img2"][0], seg1=files["seg1"][0], seg2=files["seg2"][0]))
    tr.model.eval()                                                       # main.py:65
    out = tr.eval_generate_sequence(files["img1"][0], files["img2"][0], files["seg1"][0], files["seg2"][0])
    p, q = out
    assert p.shape == (1, 30, 32, 32) and q.shape == (1, 10, 32, 32) and np.isfinite(p).all()
    saved = sorted(os.listdir("../predict"))
    assert len(saved) == 2 and saved[0].endswith("_img.npy") and saved[1].endswith("_seg.npy")
    assert np.array_equal(np.load(os.path.join("../predict", saved[0])), p)
    # inputs as trainer.py:439-450 prepares them: nearest 2x downscale by cv2's rule = even rows / columns
    assert np.array_equal(q[0, 0], files["seg1"][1][0::2, 0::2].astype(np.float32))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    want1 = (files["img1"][1].transpose(2, 0, 1).astype(np.float32) / 255.0 - mean) / std
    np.testing.assert_allclose(p[0, :3], want1, rtol=1e-6, atol=1e-6)
    # first predicted frame vs the restated loop on the trainer's own weights
    sd = tr.model.state_dict()
    with torch.no_grad():
        logits, img_next, _ = R.step(sd, hp, True, torch.from_numpy(p[:, 0:3]), torch.from_numpy(p[:, 3:6]),
                                     torch.from_numpy(q[:, 0:1]), torch.from_numpy(q[:, 1:2]))
    assert float((torch.from_numpy(p[:, 6:9]) - img_next).abs().max()) <= 1e-4 * float(img_next.abs().max())
    _ids_agree(torch.from_numpy(q[:, 2:3]), logits, "first predicted segmentation")
    # an unreadable path logs and returns (trainer.py:436-438)
    assert tr.eval_generate_sequence(str(tmp_path / "nope.png"), files["img2"][0], files["seg1"][0], files["seg2"][0]) is None


@pytest.mark.parametrize("attention", ["slot", "clip"])
def test_layout_rollout_values_and_entry(tmp_path, monkeypatch, dev, attention):
    """Token mode: generate_sequence's 8 predictions vs the CPU restatement (teacher-forced on the HIP rollout's own
    window), and eval_generate_sequence says what it cannot do instead of logging a false 'path not exists'.  With the
    per-clip attention option the rollout masks padded slots (reserved class id) as keys, as training does."""
    (tmp_path / "src").mkdir()
    monkeypatch.chdir(tmp_path / "src")
    monkeypatch.delenv("VLG_MODEL", raising=False)
    monkeypatch.setenv("VLG_ATTENTION", attention)
    from trainer import Trainer
    cfgk = dict(batch_size=3, epochs=1, print_freq=1, n_frames=8, n_slots=8, d_model=64, n_layers=2, train_clips=6, val_clips=3)
    tr = Trainer(reference_args(tmp_path / "exp", **cfgk))
    batch = next(iter(tr.val_loader))
    cls0, box0 = batch["slot_class"].cpu(), batch["slot_box"].cpu()
    steps = 8
    out_c, out_b = tr.generate_sequence(cls0, box0, steps=steps)
    assert out_c.shape == (3, steps, 8) and out_b.shape == (3, steps, 8, 4)
    params = {k: v.cpu() for k, v in tr.engine.named_params().items()}
    cls, box = cls0.clone(), box0.clone()
    with torch.no_grad():
        for i in range(steps):
            logits, raw = O.forward(params, cls, box, tr.cfg.n_layers, attention=attention,
                                    valid=(cls < tr.cfg.n_classes).float())
            last = logits[:, -1]                                                   # (B,N,C)
            top2 = last.topk(2, dim=-1).values
            clear = (top2[..., 0] - top2[..., 1]) > 1e-4 * float(last.abs().max())
            assert bool((out_c[:, i][clear] == last.argmax(-1)[clear]).all()), i
            want_b = torch.sigmoid(raw[:, -1])
            assert float((out_b[:, i] - want_b).abs().max()) <= 1e-4, i
            cls = torch.cat([cls[:, 1:], out_c[:, i][:, None]], dim=1)            # slide the window on the HIP predictions
            box = torch.cat([box[:, 1:], out_b[:, i][:, None]], dim=1)
    # layout mode has no pixel inputs: the main.py:64-67 entry logs an error and returns, as the reference's does for
    # unusable inputs (trainer.py:436-438) - no traceback
    assert tr.eval_generate_sequence("a.png", "b.png", "c.png", "d.png") is None
