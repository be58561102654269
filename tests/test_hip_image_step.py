"""GPU: the reference's own training-step body (reference src/trainer.py:193-258, SURVEY.md Appendix-A repairs,
HED edges as inputs, VGG term omitted) on the HIP kernels vs its CPU restatement (oracle/image_step_spec.py,
built on pieces pinned to the reference's outputs).  fp32; values 1e-4; gradients strict in the smooth (slopes = 1)
network and kink-tolerant with real slopes (oracle.gridnet_spec.test_params explains why)."""
import pytest
import torch

from oracle import gridnet_spec as G
from oracle import image_step_spec as S
from test_hip_gridnet import check_grads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("arch,linear", [("GridNet", True), ("GridNet", False), ("CoordGridNet", True)])
def test_reference_step_body(dev, arch, linear, flip):
    from vlg.image_engine import ImageEngine, synthetic_frames
    b, H, W, filt = 2, 32, 40, (8, 16, 24)
    coord = arch == "CoordGridNet"
    eng = ImageEngine(b, H, W, dev, arch=arch, filters=filt)
    p = G.test_params(G.param_shapes(10, filt, coord=coord), seed=5, linear=linear)
    eng.load_state_dict(p)
    batch = synthetic_frames(b, H, W, seed=21)
    parts, grads = S.loss_and_grads(p, batch, coord, flip)
    eng.forward({k: v.to(dev) for k, v in batch.items()}, flip=flip)
    eng.backward()
    got = eng.losses.cpu()
    for i, name in enumerate(("l1", "gradient", "ssim", "ce")):
        assert abs(float(got[i]) - parts[i]) <= 1e-4 * abs(parts[i]), (name, float(got[i]), parts[i])
    assert abs(float(eng.total()) - parts[4]) <= 1e-4 * abs(parts[4])
    check_grads(eng.net.named_grads(), grads, linear, tol=3e-4)
    if not linear:
        strict_step_gradients(eng, p, batch, coord)


def strict_step_gradients(eng, p, batch, coord, tol=3e-4):
    """REAL slopes, no kink allowance.  The step's gradient is discontinuous in two places: the PReLU branches and the
    losses' own kinks (|a - b| of L1, ||da| - |db|| of GradientLoss, SSIM's clamp, reference src/loss.py:20-25,68-91),
    so two fp32 forwards that agree to 2e-7 can still take different sides on a few pixels.  Each smooth piece is
    therefore compared AT THE SAME POINT: (1) the loss gradients the kernels produced vs autograd of the restated
    losses evaluated on the HIP outputs; (2) the network backward vs the restatement on the branch pattern the HIP
    forward took (test_hip_gridnet.branch_pattern also bounds where the patterns may differ), fed those gradients."""
    import torch.nn.functional as F
    from test_hip_gridnet import branch_pattern, rel_close
    std_arr = torch.tensor([0.448, 0.448, 0.450])[None, :, None, None]          # trainer.py:121
    a = eng.img.cpu().requires_grad_(True)
    sgm = eng.seg.cpu().requires_grad_(True)
    f3, seg3 = eng.f3.cpu(), eng.seg3.cpu()
    (40 * F.l1_loss(a, f3) + 20 * (S.gradient_loss(a, f3) + S.ssim_loss(a, f3)) + 10 * F.cross_entropy(sgm, seg3)).backward()
    dimg_raw = a.grad / std_arr                                                  # through img = (img - mean) / std, :212
    rel_close(eng.dseg, sgm.grad, tol=1e-4, what="d loss / d seg")
    rel_close(eng.dtmp, dimg_raw, tol=1e-4, what="d loss / d img")
    positive, _ = branch_pattern(eng.net, p, eng.x10.cpu(), coord)
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    seg_o, img_o = G.forward(q, eng.x10.cpu(), coord, branches=G.Branches(positive=positive))
    ((seg_o * sgm.grad).sum() + (img_o * dimg_raw).sum()).backward()
    check_grads(eng.net.named_grads(), {k: v.grad for k, v in q.items()}, True, tol=tol)


def test_adam_steps_reduce_the_loss(dev):
    from vlg.image_engine import ImageEngine, synthetic_frames
    eng = ImageEngine(2, 32, 32, dev, arch="CoordGridNet", filters=(8, 16, 24), lr=2e-3)
    eng.load_state_dict(G.test_params(G.param_shapes(10, (8, 16, 24), coord=True), seed=1))
    batch = {k: v.to(dev) for k, v in synthetic_frames(2, 32, 32, seed=3).items()}
    first = float(eng.train_step(batch))
    for _ in range(30):
        last = float(eng.train_step(batch))
    assert last < 0.95 * first, (first, last)
    sd = eng.state_dict()                        # still exports in the reference's key/shape format
    assert tuple(sd["lateral_in.conv.0.conv.weight"].shape) == (8, 12, 3, 3)


def test_step_with_hed_edges(dev):
    """Edges computed on the device by the frozen HED net (trainer.py:190-192, fused map) feed the same step: the
    result equals the step run on those edge maps given as inputs, and equals the CPU restatement fed the HED
    restatement's edges."""
    from oracle import hned_spec as HS
    from vlg.image_engine import ImageEngine, synthetic_frames
    b, H, W, filt = 1, 32, 32, (8, 16, 24)
    eng = ImageEngine(b, H, W, dev, arch="CoordGridNet", filters=filt, with_hed=True)
    p = G.test_params(G.param_shapes(10, filt, coord=True), seed=5, linear=True)
    eng.load_state_dict(p)
    hp = HS.test_params(1)
    eng.hed.load_state_dict(hp)
    batch = synthetic_frames(b, H, W, seed=8)
    cpu_batch = dict(batch)
    cpu_batch["e1"] = HS.forward(hp, batch["frame1"])[5]
    cpu_batch["e2"] = HS.forward(hp, batch["frame2"])[5]
    parts, grads = S.loss_and_grads(p, cpu_batch, True)
    dev_batch = {k: v.to(dev) for k, v in batch.items() if k not in ("e1", "e2")}
    eng.forward(dev_batch)
    eng.backward()
    assert abs(float(eng.total()) - parts[4]) <= 1e-4 * abs(parts[4])
    check_grads(eng.net.named_grads(), grads, True, tol=3e-4)


def test_step_with_full_combined_loss(dev):
    """CombinedLoss = VGG + GradientLoss + SSIM (reference src/loss.py:61-62) in the step; the VGG term under
    name-seeded frozen weights (parity unpinned against the reference for that term, see tests/test_hip_vgg.py)."""
    from oracle import vgg_spec as V
    from vlg.image_engine import ImageEngine, synthetic_frames
    b, H, W, filt = 1, 32, 32, (8, 16, 24)
    eng = ImageEngine(b, H, W, dev, arch="GridNet", filters=filt, with_vgg=True)
    p = G.test_params(G.param_shapes(10, filt), seed=5, linear=True)
    eng.load_state_dict(p)
    vp = V.test_params(2)
    eng.vgg.load_state_dict(vp)
    batch = synthetic_frames(b, H, W, seed=4)
    parts, grads = S.loss_and_grads(p, batch, False, vgg_params=vp)
    eng.forward({k: v.to(dev) for k, v in batch.items()})
    eng.backward()
    assert abs(float(eng.total()) - parts[4]) <= 1e-4 * abs(parts[4]), (float(eng.total()), parts[4])
    check_grads(eng.net.named_grads(), grads, False)          # ReLU trunk of the VGG term: kink-tolerant bound
