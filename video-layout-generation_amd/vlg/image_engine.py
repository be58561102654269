"""The reference's OWN training step (pixel model) on the gfx950 kernels.

Restates the body of Trainer.train() in reference src/trainer.py:184-258 with the repairs of SURVEY.md
Appendix A, every piece a HIP kernel of libvlg_hip.so:

    x = cat[e1, seg1, norm(frame1), norm(frame2), seg2, e2]   (:193-197)   vlg_prep_input (+ shared flip :200-206)
    seg, img = gridnet(x)                                      (:209)      vlg/gridnet.py (conv3x3 MFMA kernels)
    img = (img - mean_arr) / std_arr                           (:212)      vlg_affine_nchw
    loss = 40 L1(img, frame3) + 20 (GradientLoss + SsimLoss)(img, frame3) + 10 CE(seg, seg3)    (:248-251)
    backward; Adam(lr, betas=(beta1, 0.999))                   (:257-258)  analytic gradients, vlg_adam_step

Deviations, stated: (1) the edge maps e1/e2 are inputs, OR - with with_hed=True - the fused output [5] of the
frozen HED net run on frame1 / frame2 under no-grad exactly as trainer.py:190-192 intends (Appendix A-3); that
net (vlg/hned.py) is complete, but its trained weights sit at an author-local path (trainer.py:97, A-2), so
unless a checkpoint is loaded into engine.hed the edges come from whatever weights it was given;
(2) the VGG term of CombinedLoss (loss.py:29-49, 61-62) is included with with_vgg=True (vlg/vgg_loss.py) - its
arithmetic is complete, but torchvision's ImageNet weights are a download (parity unpinned), so
unless real weights are loaded into engine.vgg it is a perceptual loss under whatever frozen weights it was given;
(3) gradients are overwritten each step (A-5).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import hip
from .gridnet import GridNetHIP
from .hip import call, ptr
from .spec import ADAM_BETA1, ADAM_BETA2, ADAM_EPS, ADAM_LR, OUT_MEAN, OUT_STD

IMAGE_KEYS = ("frame1", "seg1", "frame2", "seg2", "frame3", "seg3", "e1", "e2")
W_L1, W_STYLE, W_CE = 40.0, 20.0, 10.0          # reference src/trainer.py:248-250


class ImageEngine:
    def __init__(self, batch: int, H: int, W: int, device, arch: str = "CoordGridNet", lr: float = ADAM_LR,
                 beta1: float = ADAM_BETA1, filters=(32, 64, 96), with_hed: bool = False, with_vgg: bool = False):
        if arch not in ("GridNet", "CoordGridNet"):
            raise ValueError("arch must be GridNet or CoordGridNet (reference src/main.py:101-102)")
        self.device, self.lr, self.beta1 = device, float(lr), float(beta1)
        self.b, self.H, self.W = batch, H, W
        self.net = GridNetHIP(10, batch, H, W, device, coord=(arch == "CoordGridNet"), filters=filters)
        self.vgg = None
        if with_vgg:
            from .vgg_loss import VggLossHIP
            self.vgg = VggLossHIP(batch, H, W, device)
        self.hed = None
        if with_hed:
            from .hned import HNEDHIP
            self.hed = HNEDHIP(batch, H, W, device)
        n = self.net.params.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        self.step_count = 0
        f32 = dict(dtype=torch.float32, device=device)
        self.x10 = torch.empty(batch, 10, H, W, **f32)
        self.f3 = torch.empty(batch, 3, H, W, **f32)
        self.seg3 = torch.empty(batch, H, W, dtype=torch.int64, device=device)
        self.img = torch.empty(batch, 3, H, W, **f32)
        self.dimg = torch.empty(batch, 3, H, W, **f32)
        self.dtmp = torch.empty(batch, 3, H, W, **f32)
        self.dseg = torch.empty(batch, 20, H, W, **f32)
        self.scratch = torch.zeros(hip.load().vlg_image_loss_scratch(), **f32)
        self.losses = torch.zeros(8, **f32)          # {l1, gradient, ssim, ce, vgg, -, -, -}
        arr = ctypes.c_float * 3
        self._mean, self._istd = arr(*OUT_MEAN), arr(*[1.0 / s for s in OUT_STD])
        self._zero = arr(0.0, 0.0, 0.0)

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def load_state_dict(self, sd) -> None:
        self.net.load_state_dict(sd)

    def state_dict(self):
        return self.net.state_dict()

    def forward(self, batch: Dict[str, torch.Tensor], flip: bool = False, want_grads: bool = True) -> torch.Tensor:
        """Forward + losses (+ d loss / d outputs).  Returns the device scalar-array {l1, gd, ssim, ce}."""
        if self.hed is not None and "e1" not in batch:           # trainer.py:190-192, fused map = output [5]
            batch = dict(batch)
            batch["e1"] = self.hed.forward(batch["frame1"])[5].unsqueeze(1).contiguous()
            batch["e2"] = self.hed.forward(batch["frame2"])[5].unsqueeze(1).contiguous()
        for k in IMAGE_KEYS:
            t = batch[k]
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("batch[%r] must be a contiguous HIP tensor" % k)
        b, H, W, s = self.b, self.H, self.W, self._stream()
        call("vlg_prep_input", ptr(batch["e1"]), ptr(batch["seg1"]), ptr(batch["frame1"]), ptr(batch["frame2"]),
             ptr(batch["seg2"]), ptr(batch["e2"]), ptr(batch["frame3"]), ptr(batch["seg3"]), ptr(self.x10), ptr(self.f3),
             ptr(self.seg3), b, H, W, 1 if flip else 0, s)
        self.seg, img_raw = self.net.forward(self.x10)
        call("vlg_affine_nchw", ptr(img_raw), ptr(self.img), b, 3, H * W, self._mean, self._istd, s)
        L, sc = self.losses, ptr(self.scratch)
        g = lambda t: ptr(t) if want_grads else 0
        # every loss kernel writes value and gradient in one pass; weights go in as grad_scale
        call("vlg_l1_mean", ptr(self.img), ptr(self.f3), g(self.dimg), L.data_ptr(), sc, self.img.numel(), W_L1, s)
        call("vlg_gradient_loss", ptr(self.img), ptr(self.f3), g(self.dtmp), L.data_ptr() + 4, sc, b * 3, H, W, W_STYLE, s)
        if want_grads:
            call("vlg_add_rows", ptr(self.dimg), ptr(self.dtmp), self.dimg.numel(), 1, s)
        call("vlg_ssim_loss", ptr(self.img), ptr(self.f3), g(self.dtmp), L.data_ptr() + 8, sc, b, 3, H, W, W_STYLE, s)
        if want_grads:
            call("vlg_add_rows", ptr(self.dimg), ptr(self.dtmp), self.dimg.numel(), 1, s)
        call("vlg_ce_nchw", ptr(self.seg), ptr(self.seg3), g(self.dseg), L.data_ptr() + 12, sc, b, 20, H * W, W_CE, s)
        if self.vgg is not None:                       # CombinedLoss = vgg + gd + ssim, x20 (loss.py:61-62, trainer.py:249)
            lv, dv = self.vgg.loss_and_grad(self.img, self.f3, grad_scale=W_STYLE, want_grad=want_grads)
            L[4:5].copy_(lv)
            if want_grads:
                call("vlg_add_rows", ptr(self.dimg), ptr(dv), self.dimg.numel(), 1, s)
        return self.losses

    def total(self) -> torch.Tensor:
        """40 L1 + 20 (VGG + GD + SSIM) + 10 CE   (reference src/trainer.py:248-251; the VGG slot is 0 without with_vgg)."""
        L = self.losses
        return W_L1 * L[0] + W_STYLE * (L[1] + L[2] + L[4]) + W_CE * L[3]

    def backward(self) -> None:
        b, H, W, s = self.b, self.H, self.W, self._stream()
        call("vlg_affine_nchw", ptr(self.dimg), ptr(self.dtmp), b, 3, H * W, self._zero, self._istd, s)   # d/d img_raw
        self.net.backward(self.dseg, self.dtmp)

    def adam_step(self, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        call("vlg_adam_step", ptr(self.net.params), ptr(self.net.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             self.net.params.numel(), self.step_count, self.lr, self.beta1, ADAM_BETA2, ADAM_EPS, grad_scale, self._stream())

    def train_step(self, batch, flip: bool = False) -> torch.Tensor:
        self.forward(batch, flip)
        self.backward()
        self.adam_step()
        return self.total()


def synthetic_frames(n: int, H: int, W: int, seed: int = 1024) -> Dict[str, torch.Tensor]:
    """SURVEY.md section 8d Spec R inputs: frames U[0,1), seg ids U{0..19} (float maps for frames 1,2, int64 for
    frame 3 - reference src/folder.py:97-104), edge maps U[0,1) standing in for the frozen HED's output."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in ("frame1", "frame2", "frame3"):
        out[k] = torch.rand(n, 3, H, W, generator=g)
    for k in ("seg1", "seg2"):
        out[k] = torch.randint(0, 20, (n, 1, H, W), generator=g).float()
    out["seg3"] = torch.randint(0, 20, (n, H, W), generator=g)
    for k in ("e1", "e2"):
        out[k] = torch.rand(n, 1, H, W, generator=g)
    return out
