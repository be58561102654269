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
  gridnet_keys.json         state-dict key lists of models.GridNet(10) (reference src/models/gridnet.py:7-58).
  gridnet_small.npz         models.GridNet(10, filters_level=[8,16,24]) on a (2,10,32,48) input: seg, img, input
                            gradient and EVERY parameter gradient under loss = sum(seg*r1) + sum(img*r2).
  gridnet_full64.npz        models.GridNet(10) (real widths 32/64/96) on (1,10,64,64): outputs, dx, seven parameter
                            gradients, all PReLU-slope gradients, |grad| sums of every tensor.
  hned.npz                  models.HNED() (reference src/models/hned.py:9-105) with oracle.hned_spec.test_params weights
                            (the trained ones are at an author-local path, trainer.py:97) on (1,3,64,64) and (2,3,32,48)
                            inputs: all six outputs (d1..d5, fuse).
  coordgridnet_256.npz      models.CoordGridNet(10, filters_level=[8,16,24]) on (1,10,256,256) (the only size the
                            reference accepts): img, seg crop + checksums, dx crop, lateral_in / up_05 gradients.
                            Parameters in all three come from oracle.gridnet_spec.test_params (name-seeded), so
                            no weights are stored.
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

    # ---- GridNet / CoordGridNet: outputs and parameter gradients of the reference modules themselves,
    # with parameters overwritten by oracle.gridnet_spec.test_params (so fixtures need not store weights)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import gridnet_spec as G

    def run(model, shapes, x, seed, linear=False, want_margin=False):
        params = G.test_params(shapes, seed=seed, linear=linear)
        sd = model.state_dict()
        assert list(sd.keys()) == list(shapes.keys()), (list(sd.keys())[:5], list(shapes.keys())[:5])
        for k, v in params.items():
            assert tuple(sd[k].shape) == tuple(v.shape), k
        model.load_state_dict(params)
        margin = [float("inf")]
        hooks = []
        if want_margin:     # smallest |pre-activation| entering any PReLU (see gridnet_spec.test_params docstring)
            for m in model.modules():
                if isinstance(m, nn.PReLU):
                    hooks.append(m.register_forward_hook(lambda mod, inp, out: margin.__setitem__(0, min(margin[0], float(inp[0].detach().abs().min())))))
        g = torch.Generator().manual_seed(seed + 77)
        xin = x.clone().requires_grad_(True)
        seg, img = model(xin)
        for h in hooks:
            h.remove()
        r_seg, r_img = torch.randn(seg.shape, generator=g), torch.randn(img.shape, generator=g)
        ((seg * r_seg).sum() + (img * r_img).sum()).backward()
        grads = {k: v.grad.detach() for k, v in model.named_parameters()}
        return seg.detach(), img.detach(), grads, xin.grad.detach(), r_seg, r_img, margin[0]

    # A: small channels, non-square, batch 2: everything stored, with the real slopes and with slopes = 1
    # (see oracle.gridnet_spec.test_params for why both exist)
    filt = [8, 16, 24]
    shapes = G.param_shapes(10, filt)
    x = torch.randn(2, 10, 32, 48, generator=torch.Generator().manual_seed(3))
    rec = {"x": x.numpy()}
    for tag, linear in (("", False), ("lin_", True)):
        seg, img, grads, dx, r_seg, r_img, margin = run(ref_models.GridNet(10, filters_level=filt), shapes, x, 1,
                                                       linear=linear, want_margin=True)
        rec.update({tag + "seg": seg.numpy(), tag + "img": img.numpy(), tag + "dx": dx.numpy(),
                    tag + "kink_margin": np.float64(margin)})
        rec.update({tag + "grad:" + k: v.numpy() for k, v in grads.items()})
    rec.update(r_seg=r_seg.numpy(), r_img=r_img.numpy())
    np.savez_compressed(os.path.join(OUT, "gridnet_small.npz"), **rec)

    # B: the real channel widths [32,64,96] at 64x64, twice: real slopes (forward strict, gradients kink-tolerant)
    # and slopes = 1 (everything strict)
    shapes = G.param_shapes(10)
    x = torch.randn(1, 10, 64, 64, generator=torch.Generator().manual_seed(4))
    keep = ["lateral_in.conv.1.weight", "lateral_in.conv2.bias", "down_10.conv.1.weight", "lateral_23.conv.3.weight",
            "up_05.up.2.weight", "lateral_out_img.conv.3.weight", "lateral_out_seg.conv.1.bias"]
    rec = {}
    for tag, linear in (("", False), ("lin_", True)):
        seg, img, grads, dx, r_seg, r_img, _ = run(ref_models.GridNet(10), shapes, x, seed=2, linear=linear)
        rec.update({tag + "seg": seg.numpy(), tag + "img": img.numpy(), tag + "dx": dx.numpy()})
        rec.update({tag + "grad:" + k: grads[k].numpy() for k in keep})
        rec[tag + "prelu_grads"] = np.array([float(v) for k, v in grads.items() if v.numel() == 1], dtype=np.float32)
        rec[tag + "grad_abs_sums"] = np.array([float(v.abs().sum()) for v in grads.values()], dtype=np.float64)
    rec["prelu_names"] = np.array([k for k, v in grads.items() if v.numel() == 1])
    np.savez_compressed(os.path.join(OUT, "gridnet_full64.npz"), **rec)

    # C: CoordGridNet (accepts 256x256 only: AddCoords buffers are hard-coded, modules.py:69-70; it also calls
    # .cuda() in its constructor, so construct it with Tensor.cuda patched to a no-op - SURVEY.md section 8c)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    rec = {}
    try:
        shapes = G.param_shapes(10, filt, coord=True)
        x = torch.randn(1, 10, 256, 256, generator=torch.Generator().manual_seed(5))
        for tag, linear in (("", False), ("lin_", True)):
            seg, img, grads, dx, r_seg, r_img, _ = run(ref_models.CoordGridNet(10, filters_level=filt), shapes, x, seed=3, linear=linear)
            rec.update({tag + "seg_crop": seg[:, :, :24, :24].numpy(), tag + "img_crop": img[:, :, 100:164, 100:164].numpy(),
                        tag + "seg_sum": np.float64(seg.double().sum()), tag + "seg_sq": np.float64((seg.double() ** 2).sum()),
                        tag + "img_sum": np.float64(img.double().sum()), tag + "img_sq": np.float64((img.double() ** 2).sum()),
                        tag + "dx_crop": dx[:, :, -16:, -16:].numpy()})
            rec.update({tag + "grad:" + k: v.numpy() for k, v in grads.items() if k.startswith("lateral_in") or k.startswith("up_05")})
    finally:
        torch.Tensor.cuda = orig_cuda
    np.savez_compressed(os.path.join(OUT, "coordgridnet_256.npz"), **rec)

    # ---- HED edge detector (frozen, forward only): reference module with name-seeded weights
    from oracle import hned_spec as HS
    hed = ref_models.HNED()
    hp = HS.test_params(seed=0)
    assert list(hed.state_dict().keys()) == list(hp.keys())
    hed.load_state_dict(hp)
    rec = {}
    for tag, shape, sd_ in (("a", (1, 3, 64, 64), 6), ("b", (2, 3, 32, 48), 7)):
        x = torch.rand(shape, generator=torch.Generator().manual_seed(sd_))
        with torch.no_grad():
            outs = hed(x)
        rec[tag + "_x"] = x.numpy()
        rec[tag + "_out"] = torch.stack([o[:, 0] for o in outs]).numpy()        # (6, b, H, W)
    np.savez_compressed(os.path.join(OUT, "hned.npz"), **rec)

    torch.manual_seed(0)
    gn = ref_models.GridNet(10)
    with open(os.path.join(OUT, "gridnet_keys.json"), "w") as f:
        json.dump({"GridNet": {k: list(v.shape) for k, v in gn.state_dict().items()},
                   "n_params": sum(p.numel() for p in gn.parameters())}, f, indent=0)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
