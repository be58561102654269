"""GPU: the reference-real pixel ops (TRUE parity targets) through the C ABI, checked against
(1) golden vectors produced by the reference itself (tests/golden/image_losses_*.npz,
prep_input.npz, adam_beta05.npz) and (2) the plain-C oracle on more shapes, including the
reference's full 256x256 frame size and ragged edges.  fp32, tolerance 1e-4 as BASELINE.json states."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close
from oracle import image_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from vlg import hip
    hip.load()
    return hip


def S():
    return torch.cuda.current_stream().cuda_stream


def run_loss(H, dev, name, a, b, *dims):
    ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    g = torch.full_like(ad, float("nan"))
    loss = torch.zeros(1, device=dev)
    scratch = torch.zeros(H.load().vlg_image_loss_scratch(), device=dev)
    H.call(name, ad.data_ptr(), bd.data_ptr(), g.data_ptr(), loss.data_ptr(), scratch.data_ptr(), *dims, 1.0, S())
    return float(loss.item()), g.cpu().numpy()


def run_ce(H, dev, logits, target, scale=1.0):
    ld, td = torch.from_numpy(logits).to(dev), torch.from_numpy(target).to(dev)
    g = torch.full_like(ld, float("nan"))
    loss = torch.zeros(1, device=dev)
    scratch = torch.zeros(H.load().vlg_image_loss_scratch(), device=dev)
    b, C = logits.shape[:2]
    H.call("vlg_ce_nchw", ld.data_ptr(), td.data_ptr(), g.data_ptr(), loss.data_ptr(), scratch.data_ptr(), b, C,
           int(np.prod(logits.shape[2:])), scale, S())
    return float(loss.item()), g.cpu().numpy()


@pytest.mark.parametrize("hw", [16, 64])
def test_losses_match_reference_golden(H, dev, hw):
    z = np.load(os.path.join(GOLDEN, "image_losses_%d.npz" % hw))
    a, b = z["a"], z["b"]
    cases = (("vlg_gradient_loss", "gradient", (6, hw, hw)), ("vlg_ssim_loss", "ssim", (2, 3, hw, hw)),
             ("vlg_l1_mean", "l1", (a.size,)))
    for fn, key, dims in cases:
        v, g = run_loss(H, dev, fn, a, b, *dims)
        assert abs(v - float(z[key + "_value"])) <= 1e-4 * abs(float(z[key + "_value"])), key
        np.testing.assert_allclose(g, z[key + "_grad"], rtol=1e-4, atol=1e-7, err_msg=key)
    v, g = run_ce(H, dev, z["logits"], z["target"])
    assert abs(v - float(z["ce_value"])) <= 1e-4 * float(z["ce_value"])
    np.testing.assert_allclose(g, z["ce_grad"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("b,H_,W_", [(1, 256, 256), (2, 37, 53), (1, 3, 3), (3, 32, 33)])
def test_losses_match_c_oracle(H, dev, b, H_, W_):
    rng = np.random.default_rng(b * 1000 + H_)
    x = rng.random((b, 3, H_, W_), dtype=np.float32)
    y = (x + 0.2 * rng.standard_normal((b, 3, H_, W_)).astype(np.float32)).clip(0, 1).astype(np.float32)
    for fn, ref, dims in (("vlg_gradient_loss", R.gradient_loss, (b * 3, H_, W_)),
                          ("vlg_ssim_loss", R.ssim_loss, (b, 3, H_, W_)), ("vlg_l1_mean", R.l1_mean, (x.size,))):
        v, g = run_loss(H, dev, fn, x, y, *dims)
        rv, rg = ref(x, y)
        assert abs(v - rv) <= 1e-4 * max(abs(rv), 1e-6), fn
        np.testing.assert_allclose(g, rg, rtol=1e-4, atol=2e-7 * max(1.0, 1e3 / x.size), err_msg=fn)
    logits = (rng.standard_normal((b, 20, H_, W_)) * 3).astype(np.float32)
    target = rng.integers(0, 20, (b, H_, W_))
    target[0, 0, :2] = -100                                      # torch's default ignore_index
    v, g = run_ce(H, dev, logits, target, scale=10.0)            # x10: reference src/trainer.py:250
    rv, rg = R.ce_nchw(logits, target)
    assert abs(v - rv) <= 1e-4 * abs(rv)
    np.testing.assert_allclose(g, 10.0 * rg, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("flip", [0, 1])
def test_prep_input_matches_reference_expressions(H, dev, flip):
    z = np.load(os.path.join(GOLDEN, "prep_input.npz"))
    t = {k: torch.from_numpy(z[k]).to(dev) for k in ("e1", "seg1", "frame1", "frame2", "seg2", "e2", "frame3", "seg3")}
    b, _, Hh, Ww = z["frame1"].shape
    x = torch.full((b, 10, Hh, Ww), float("nan"), device=dev)
    f3 = torch.full((b, 3, Hh, Ww), float("nan"), device=dev)
    s3 = torch.full((b, Hh, Ww), -1, dtype=torch.int64, device=dev)
    H.call("vlg_prep_input", t["e1"].data_ptr(), t["seg1"].data_ptr(), t["frame1"].data_ptr(), t["frame2"].data_ptr(),
           t["seg2"].data_ptr(), t["e2"].data_ptr(), t["frame3"].data_ptr(), t["seg3"].data_ptr(), x.data_ptr(),
           f3.data_ptr(), s3.data_ptr(), b, Hh, Ww, flip, S())
    tag = "flip" if flip else "noflip"
    np.testing.assert_allclose(x.cpu().numpy(), z["x_" + tag], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(f3.cpu().numpy(), z["frame3_" + tag], rtol=1e-6, atol=1e-6)
    assert np.array_equal(s3.cpu().numpy(), z["seg3_" + tag])


def test_adam_matches_reference_golden(H, dev):
    z = np.load(os.path.join(GOLDEN, "adam_beta05.npz"))
    p = torch.from_numpy(z["p0"].copy()).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(1, 4):
        g = torch.from_numpy(z["g%d" % s]).to(dev)
        H.call("vlg_adam_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 16, s, 2e-4, 0.5, 0.999, 1e-8,
               1.0, S())
        assert_close(p, torch.from_numpy(z["p%d" % s]), rtol=1e-6, atol=1e-7, what="adam golden step %d" % s)


def test_flip_is_an_involution_and_losses_are_flip_invariant(H, dev):
    """Size-independent properties at the reference's full frame size (256x256, b=4)."""
    rng = np.random.default_rng(3)
    x = rng.random((4, 3, 256, 256), dtype=np.float32)
    y = rng.random((4, 3, 256, 256), dtype=np.float32)
    for fn, dims in (("vlg_gradient_loss", (12, 256, 256)), ("vlg_ssim_loss", (4, 3, 256, 256)), ("vlg_l1_mean", (x.size,))):
        v0, g0 = run_loss(H, dev, fn, x, y, *dims)
        v1, g1 = run_loss(H, dev, fn, x[..., ::-1].copy(), y[..., ::-1].copy(), *dims)
        assert abs(v0 - v1) <= 1e-5 * abs(v0), fn
        np.testing.assert_allclose(g1[..., ::-1], g0, rtol=1e-4, atol=1e-9, err_msg=fn)
        v2, _ = run_loss(H, dev, fn, x, x, *dims)
        assert abs(v2) < 1e-6, fn                                   # identical images -> zero loss
