"""GPU: VggLoss (reference src/loss.py:29-49) - value and input gradient against the torch-CPU restatement
oracle/vgg_spec.py.  PARITY UNPINNED against the reference itself for this one term: torchvision and the ImageNet
weights are unavailable offline, so the reference class cannot be constructed; what is pinned is
the arithmetic of the published VGG19 configuration with name-seeded weights.  Gradient comparison uses the
kink-tolerant bound (12 ReLU layers + max-pools: see oracle.gridnet_spec.test_params)."""
import pytest
import torch

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
