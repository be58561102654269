#!/usr/bin/env python
"""Generates tests/golden/*.npz from the parts of the reference that import on CPU.

Run in the build container only (needs /root/reference; it never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Sources of each fixture (reference file:line):
  image_losses_{16,64}.npz  loss.GradientLoss / loss.SsimLoss imported from reference src/loss.py:16-25,64-91
                            (torchvision stubbed: loss.py:10 imports it, VggLoss needs a weight download
                            and is NOT exercised); nn.CrossEntropyLoss('mean') and nn.L1Loss() constructed
                            exactly as reference src/trainer.py:124,130 does.  Values + input gradients.
  adam_beta05.npz           torch.optim.Adam(lr=2e-4, betas=(0.5, 0.999)) as reference src/trainer.py:83
                            (defaults from src/main.py:139-141), three steps on a 16-element vector.
  prep_input.npz            the tensor expressions of reference src/trainer.py:193-206 (normalise, 10-channel
                            cat, shared flip) evaluated verbatim on small inputs; trainer.py itself cannot be
                            imported here (torchvision / tensorboardX / cv2 absent).
  average_meter.npz         utils.AverageMeter imported from reference src/utils.py:1-16.
  gridnet_keys.json         state-dict key lists of models.GridNet(10) (reference src/models/gridnet.py:7-58),
                            kept for checkpoint-format work on the conv path.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))   # loss.py:10 imports it at top level
    import loss as ref_loss          # noqa: E402
    import models as ref_models      # noqa: E402
    import utils as ref_utils        # noqa: E402
    os.makedirs(OUT, exist_ok=True)

    for hw in (16, 64):
        g = torch.Generator().manual_seed(hw)
        a = torch.rand(2, 3, hw, hw, generator=g).requires_grad_(True)
        b = torch.rand(2, 3, hw, hw, generator=g)
        logits = (torch.randn(2, 20, hw, hw, generator=g) * 2).requires_grad_(True)
        target = torch.randint(0, 20, (2, hw, hw), generator=g)
        out = {"a": a.detach().numpy(), "b": b.numpy(), "logits": logits.detach().numpy(), "target": target.numpy()}
        for name, fn in (("gradient", ref_loss.GradientLoss()), ("ssim", ref_loss.SsimLoss()), ("l1", nn.L1Loss())):
            a.grad = None
            v = fn(a, b)
            v.backward()
            out[name + "_value"] = np.float32(v.item())
            out[name + "_grad"] = a.grad.numpy().copy()
        ce = nn.CrossEntropyLoss(reduction="mean")
        v = ce(input=logits, target=target)
        v.backward()
        out["ce_value"] = np.float32(v.item())
        out["ce_grad"] = logits.grad.numpy().copy()
        np.savez_compressed(os.path.join(OUT, "image_losses_%d.npz" % hw), **out)

    # Adam(beta1=0.5), three steps
    g = torch.Generator().manual_seed(5)
    p = torch.randn(16, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2e-4, betas=(0.5, 0.999))
    rec = {"p0": p.detach().numpy().copy()}
    for s in range(1, 4):
        gr = torch.randn(16, generator=g) * (10.0 ** (s - 2))
        p.grad = gr.clone()
        opt.step()
        rec["g%d" % s] = gr.numpy()
        rec["p%d" % s] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "adam_beta05.npz"), **rec)

    # input preparation, reference src/trainer.py:193-206 evaluated verbatim
    g = torch.Generator().manual_seed(9)
    bsz, H, W = 2, 6, 10
    frame1, frame2, frame3 = (torch.rand(bsz, 3, H, W, generator=g) for _ in range(3))
    seg1, seg2 = (torch.randint(0, 20, (bsz, 1, H, W), generator=g).float() for _ in range(2))
    seg3 = torch.randint(0, 20, (bsz, H, W), generator=g)
    e1, e2 = (torch.rand(bsz, 1, H, W, generator=g) for _ in range(2))
    img_std_arr = torch.tensor([0.229, 0.224, 0.225])[None, :, None, None]      # trainer.py:122
    img_mean_arr = torch.tensor([0.485, 0.456, 0.406])[None, :, None, None]     # trainer.py:123
    rec = {"e1": e1.numpy(), "seg1": seg1.numpy(), "frame1": frame1.numpy(), "frame2": frame2.numpy(),
           "seg2": seg2.numpy(), "e2": e2.numpy(), "frame3": frame3.numpy(), "seg3": seg3.numpy()}
    f1 = (frame1 - img_mean_arr) / img_std_arr                                   # trainer.py:193
    f2 = (frame2 - img_mean_arr) / img_std_arr                                   # trainer.py:194
    f3 = (frame3 - img_mean_arr) / img_std_arr                                   # trainer.py:195
    x = torch.cat([e1, seg1, f1, f2, seg2, e2], dim=1)                           # trainer.py:197
    rec.update(x_noflip=x.numpy(), frame3_noflip=f3.numpy(), seg3_noflip=seg3.numpy())
    rec.update(x_flip=torch.flip(x, [3]).numpy(), frame3_flip=torch.flip(f3, [3]).numpy(),   # trainer.py:202-205
               seg3_flip=torch.flip(seg3, [2]).numpy())                                       # trainer.py:206
    np.savez_compressed(os.path.join(OUT, "prep_input.npz"), **rec)

    # AverageMeter
    am = ref_utils.AverageMeter()
    vals, ns, avgs = [3.5, 1.25, 7.0, 0.5], [4, 4, 2, 1], []
    for v, n in zip(vals, ns):
        am.update(v, n)
        avgs.append(am.avg)
    np.savez_compressed(os.path.join(OUT, "average_meter.npz"), vals=np.array(vals), ns=np.array(ns), avgs=np.array(avgs))

    torch.manual_seed(0)
    gn = ref_models.GridNet(10)
    with open(os.path.join(OUT, "gridnet_keys.json"), "w") as f:
        json.dump({"GridNet": {k: list(v.shape) for k, v in gn.state_dict().items()},
                   "n_params": sum(p.numel() for p in gn.parameters())}, f, indent=0)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
