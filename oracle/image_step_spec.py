"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's training-step body (pixel model).

Follows reference src/trainer.py:193-258 line by line (cited inline) with torch-CPU ops, on top of
oracle/gridnet_spec.py (bit-identical to the reference's GridNet on the committed fixtures) and torch
restatements of the reference's losses (src/loss.py:20-25, 68-91 - the same expressions; the plain-C
oracle and the HIP kernels are pinned to the reference's own outputs for them in tests/golden).
The HED edge maps are inputs and the VGG term is absent, exactly as in vlg/image_engine.py.
Only tests/ import this file.
"""
import torch
import torch.nn.functional as F

from oracle import gridnet_spec as G


def gradient_loss(a, b):                                    # reference src/loss.py:20-25
    xloss = torch.sum(torch.abs(torch.abs(a[:, :, 1:, :] - a[:, :, :-1, :]) - torch.abs(b[:, :, 1:, :] - b[:, :, :-1, :])))
    yloss = torch.sum(torch.abs(torch.abs(a[:, :, :, 1:] - a[:, :, :, :-1]) - torch.abs(b[:, :, :, 1:] - b[:, :, :, :-1])))
    return (xloss + yloss) / (a.size()[0] * a.size()[1] * a.size()[2] * a.size()[3])


def ssim_loss(x, y):                                        # reference src/loss.py:68-91
    def one(x, y):
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
        sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
        sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
        sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
        n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
        d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
        return torch.clamp((1 - n / d) / 2, 0, 1).mean()
    return sum(one(x[:, i], y[:, i]) for i in range(x.size()[1]))


def step_losses(p, batch, coord, flip=False, vgg_params=None, branches=None):
    img_mean = torch.tensor([0.485, 0.456, 0.406])[None, :, None, None]        # trainer.py:123
    img_std = torch.tensor([0.229, 0.224, 0.225])[None, :, None, None]         # trainer.py:122
    mean_arr = torch.tensor([-0.03, -0.088, -0.188])[None, :, None, None]      # trainer.py:120
    std_arr = torch.tensor([0.448, 0.448, 0.450])[None, :, None, None]         # trainer.py:121
    f1 = (batch["frame1"] - img_mean) / img_std                                 # :193
    f2 = (batch["frame2"] - img_mean) / img_std                                 # :194
    f3 = (batch["frame3"] - img_mean) / img_std                                 # :195
    x = torch.cat([batch["e1"], batch["seg1"], f1, f2, batch["seg2"], batch["e2"]], dim=1)   # :197
    seg3 = batch["seg3"]
    if flip:                                                                    # :200-206
        x, f3, seg3 = torch.flip(x, [3]), torch.flip(f3, [3]), torch.flip(seg3, [2])
    seg, img = G.forward(p, x, coord, branches=branches)                        # :209
    img = (img - mean_arr) / std_arr                                            # :212
    l1 = F.l1_loss(img, f3)                                                     # :248 (x40 below)
    gd, ss = gradient_loss(img, f3), ssim_loss(img, f3)                         # :249 CombinedLoss without VGG
    ce = F.cross_entropy(seg, seg3)                                             # :250
    style = gd + ss
    if vgg_params is not None:                                                  # CombinedLoss = vgg + gd + ssim, loss.py:61-62
        from oracle import vgg_spec
        style = style + vgg_spec.vgg_loss(vgg_params, img, f3)
    return l1, gd, ss, ce, 40 * l1 + 20 * style + 10 * ce                      # :251


def loss_and_grads(p, batch, coord, flip=False, vgg_params=None, branches=None):
    """`branches` (oracle.gridnet_spec.Branches) pins / records the PReLU branch pattern, see there."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    parts = step_losses(q, batch, coord, flip, vgg_params, branches)
    parts[4].backward()
    return [float(v.detach()) for v in parts], {k: v.grad for k, v in q.items()}
