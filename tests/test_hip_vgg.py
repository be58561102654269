"""GPU: VggLoss (reference src/loss.py:29-49) - value and input gradient against the torch-CPU restatement
oracle/vgg_spec.py.  PARITY UNPINNED against the reference itself for this one term: torchvision and the ImageNet
weights are unavailable offline, so the reference class cannot be constructed; what is pinned is
the arithmetic of the published VGG19 configuration with name-seeded weights.  The input gradient is checked twice:
coarsely against the restatement's own autograd (kink-tolerant bound: 12 ReLU layers + max-pools + |.| each flip local
derivatives inside the forward's rounding band), and STRICTLY (1e-4) against an fp64 evaluation on the kink pattern the
HIP forward actually took (ReLU masks, max-pool argmax, L1 signs: VggLossHIP.preactivations()) - given the pattern the
loss is linear in the image, so nothing is left to tolerate; the pattern itself may differ from the restatement's own
only inside the rounding band."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vgg_spec as V
from test_hip_gridnet import kink_tolerant

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,H,W", [(1, 32, 32), (2, 16, 24)])
def test_vgg_loss_value_and_input_gradient(dev, b, H, W):
    from vlg.vgg_loss import VggLossHIP
    assert sum(int(torch.tensor(s).prod()) for s in V.param_shapes().values()) == 10585152
    p = V.test_params(0)
    net = VggLossHIP(b, H, W, dev)
    assert net.reference_shapes() == {k: tuple(v) for k, v in V.param_shapes().items()}
    net.load_state_dict(p)
    g = torch.Generator().manual_seed(H)
    out = torch.randn(b, 3, H, W, generator=g).requires_grad_(True)
    tgt = torch.randn(b, 3, H, W, generator=g)
    want = V.vgg_loss(p, out, tgt)
    (20.0 * want).backward()                                  # x20: reference src/trainer.py:249
    loss, dimg = net.loss_and_grad(out.detach().to(dev), tgt.to(dev), grad_scale=20.0)
    assert abs(float(loss) - float(want)) <= 1e-4 * abs(float(want)), (float(loss), float(want))
    kink_tolerant(dimg, out.grad, "d vgg / d output")
    loss2, none = net.loss_and_grad(out.detach().to(dev), out.detach().to(dev), want_grad=False)
    assert float(loss2) == 0.0 and none is None


def _windows(x):
    """(b, C, H, W) -> (b, C, H/2, W/2, 4): the four candidates of every 2x2 max-pool window"""
    b, C, H, W = x.shape
    return x.view(b, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, C, H // 2, W // 2, 4)


def _fp64_on_pattern(p, out, pre_hip, scale):
    """fp64 restatement of scale * VggLoss evaluated ON the kink pattern of the HIP forward.  Returns (d/d out, flips): how
    many kink decisions differ from the fp64 forward's own, after asserting each of them lies inside the rounding band."""
    band, flips = 1e-5, 0
    x = out.double().requires_grad_(True)
    h, idx, prev = x, 0, None
    for v in V.CFG:
        if v == "M":
            win_h, win_o = _windows(pre_hip[prev].double()), _windows(h)      # HIP pre-activation picks the argmax
            sel = win_h.argmax(dim=-1, keepdim=True)
            own = _windows(pre_own).argmax(dim=-1, keepdim=True)
            diff = sel != own
            if bool(diff.any()):
                top2 = _windows(pre_own).topk(2, dim=-1).values
                gap = (top2[..., 0] - top2[..., 1])[diff.squeeze(-1)]
                # (a window whose candidates are all negative passes no gradient at all: ReLU' = 0 whichever is picked)
                dead = (top2[..., 0] <= 0)[diff.squeeze(-1)]
                assert bool(((gap <= band * float(pre_own.abs().max())) | dead).all()), "max-pool argmax differs outside the rounding band"
                flips += int(diff.sum())
            h = torch.gather(win_o, -1, sel).squeeze(-1)
            idx += 1
        else:
            key = "features.%d" % idx
            pre_own = F.conv2d(h, p[key + ".weight"].double(), p[key + ".bias"].double(), padding=1)
            mask = pre_hip[key] > 0
            diff = mask != (pre_own.detach() > 0)
            if bool(diff.any()):
                assert float(pre_own.detach()[diff].abs().max()) <= band * float(pre_own.detach().abs().max()), key + ": ReLU branch differs outside the rounding band"
                flips += int(diff.sum())
            h = pre_own * mask.double()
            prev, pre_own = key, pre_own.detach()
            idx += 2
    fo_hip, ft_hip = torch.relu(pre_hip[prev]), torch.relu(pre_hip["target"])
    sign = torch.sign(fo_hip - ft_hip).double()                              # the L1's own kink: taken from the HIP features
    loss = scale * (sign * h).sum() / h.numel()
    loss.backward()
    return x.grad, flips


# (1, 32, 32) and (2, 16, 24): every 128..512-channel layer runs split-K (a handful of row tiles for 256 CUs);
# (2, 256, 256): the reference's frame size - the 128 -> 128 layer at 2 x 128 x 128 pixels has 265 row tiles, so both its
# forward and its data gradient take the tail split (conv_tail_plan), the 64-channel layers the plain one-block-per-tile path
@pytest.mark.parametrize("b,H,W", [(1, 32, 32), (2, 16, 24), (2, 256, 256)])
def test_vgg_gradient_strict_given_pattern(dev, b, H, W):
    from vlg import hip
    from vlg.vgg_loss import VggLossHIP
    lib = hip.load()
    p = V.test_params(0)
    net = VggLossHIP(b, H, W, dev)
    net.load_state_dict(p)
    rows = lambda k: b * ((H >> k) + 2) * ((W >> k) + 2)
    if (b, H, W) == (2, 256, 256):      # the paths this shape is here for (a planning change must not silently drop them)
        assert lib.vlg_conv3x3_fwd_splits(rows(1), 128, 128, 128) == 1 and lib.vlg_conv3x3_fwd_workspace(rows(1), 128, 128, 128) > 0
        assert lib.vlg_conv3x3_dgrad_splits(rows(1), 128, 128) == 1 and lib.vlg_conv3x3_dgrad_workspace(rows(1), 128, 128) > 0
        assert lib.vlg_conv3x3_fwd_splits(rows(3), 512, 512, 512) > 1
    else:
        assert lib.vlg_conv3x3_fwd_splits(rows(2), 256, 256, 256) > 1 and lib.vlg_conv3x3_dgrad_splits(rows(3), 512, 512) > 1
    g = torch.Generator().manual_seed(H + 1)
    out = torch.randn(b, 3, H, W, generator=g)
    tgt = torch.randn(b, 3, H, W, generator=g)
    loss, dimg = net.loss_and_grad(out.to(dev), tgt.to(dev), grad_scale=20.0)
    pre = {k: v.cpu() for k, v in net.preactivations().items()}
    torch.set_num_threads(16)
    want, flips = _fp64_on_pattern(p, out, pre, 20.0)
    scale = float(want.abs().max())
    err = float((dimg.cpu().double() - want).abs().max()) / scale
    assert err <= 1e-4, "d vgg / d output given the pattern: %.3g of max (flips inside the band: %d)" % (err, flips)
