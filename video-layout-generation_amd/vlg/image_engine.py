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
from .gridnet import GridNetHIP, reference_param_order
from .hip import call, ptr
from .spec import ADAM_BETA1, ADAM_BETA2, ADAM_EPS, ADAM_LR, IMG_MEAN, IMG_STD, OUT_MEAN, OUT_STD

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
        # {l1, gradient, ssim, ce, vgg, -, -, -}: the 8 floats behind the last gradient, so under data parallelism
        # they ride in the last gradient bucket (vlg/gridnet.py TAIL_EXTRA)
        self.losses = self.net.grads_ext[self.net.n_params_padded:]
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

    # ------------------------------------------------------------------ optimiser state (checkpoints)
    def optimizer_state(self) -> Dict[str, object]:
        """Adam state in torch.optim.Adam.state_dict() form - {'state': {i: {'step','exp_avg','exp_avg_sq'}},
        'param_groups': [...]} with i indexing the reference model's parameters() order and tensors in the reference's
        shapes - i.e. what reference src/trainer.py:91-92 loads into its own optimizer ('optimizer' entry of --ckpt)."""
        order = reference_param_order(self.net.coord)
        state = {}
        if self.step_count > 0:                      # torch keeps no per-parameter state before the first step
            m, v = self.net.unpack(self.exp_avg), self.net.unpack(self.exp_avg_sq)
            state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": m[k], "exp_avg_sq": v[k]}
                     for i, k in enumerate(order)}
        group = {"lr": self.lr, "betas": (self.beta1, ADAM_BETA2), "eps": ADAM_EPS, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer(self, st: Dict[str, object]) -> None:
        """Accepts a torch.optim.Adam state_dict over the reference model's parameters (reference trainer.py:92) - mapped
        parameter by parameter into the kernels' flat layout - or the flat {'exp_avg','exp_avg_sq','step'} form."""
        n = self.net.params.numel()
        if "state" in st and "param_groups" in st:
            order = reference_param_order(self.net.coord)
            ids = [i for g in st["param_groups"] for i in g["params"]]
            if len(ids) != len(order):
                raise ValueError("optimizer state covers %d parameters, %s has %d"
                                 % (len(ids), "CoordGridNet" if self.net.coord else "GridNet", len(order)))
            state = st["state"]
            if len(state) == 0:
                self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.step_count = 0
                return
            missing = [i for i in ids if i not in state]
            if missing:
                raise ValueError("optimizer state lacks entries for parameter ids %s" % missing[:4])
            steps = {int(float(state[i]["step"])) for i in ids}
            if len(steps) != 1:
                raise ValueError("per-parameter step counts differ (%s): one flat Adam step cannot represent that" % sorted(steps)[:4])
            self.net.pack({k: state[i]["exp_avg"] for i, k in zip(ids, order)}, self.exp_avg)
            self.net.pack({k: state[i]["exp_avg_sq"] for i, k in zip(ids, order)}, self.exp_avg_sq)
            self.step_count = steps.pop()
            return
        for k in ("exp_avg", "exp_avg_sq", "step"):
            if k not in st:
                raise ValueError("optimizer state is neither a torch.optim.Adam state_dict nor the flat form (no %r)" % k)
        if st["exp_avg"].numel() != n or st["exp_avg_sq"].numel() != n:
            raise ValueError("optimizer state has %d elements, model has %d" % (st["exp_avg"].numel(), n))
        self.exp_avg.copy_(st["exp_avg"]); self.exp_avg_sq.copy_(st["exp_avg_sq"]); self.step_count = int(st["step"])

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

    def backward(self, reducer=None) -> None:
        b, H, W, s = self.b, self.H, self.W, self._stream()
        call("vlg_affine_nchw", ptr(self.dimg), ptr(self.dtmp), b, 3, H * W, self._zero, self._istd, s)   # d/d img_raw
        self.net.backward(self.dseg, self.dtmp, reducer)

    def adam_step(self, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        call("vlg_adam_step", ptr(self.net.params), ptr(self.net.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             self.net.params.numel(), self.step_count, self.lr, self.beta1, ADAM_BETA2, ADAM_EPS, grad_scale, self._stream())

    def train_step(self, batch, flip: bool = False, reducer=None) -> torch.Tensor:
        """forward -> losses -> backward (+ bucketed gradient all-reduce overlapped with it) -> Adam.  Returns the total
        loss (summed over ranks when a reducer is attached: the loss floats travel in the last bucket)."""
        self.forward(batch, flip)
        self.backward(reducer)
        if reducer is not None:
            reducer.wait()
            self.adam_step(reducer.grad_scale)                     # 1/world: DDP's gradient mean, trainer.py:113
        else:
            self.adam_step()
        return self.total()


class FrameRollout:
    """Autoregressive prediction of the next frames, reference src/trainer.py:453-476 (generate_sequence), forward only.

    Starting from two ImageNet-normalised frames and their segmentation-id maps (what eval_generate_sequence hands
    over, trainer.py:440-450), `steps` (8 in the reference) times:
        x = cat[e(-2), seg[-2], img[-2], img[-1], seg[-1], e(-1)]            trainer.py:461
        seg_next, img_next = gridnet(x)                                      :464
        img_next = (img_next - mean_arr) / std_arr                           :466
        seg_next = argmax(seg_next, dim=1) as float                          :467
    and the lists grow by the prediction (:468-469).  Returns (p, q) = cat(img, dim=1), cat(seg, dim=1) - the two
    arrays the reference saves to ../predict (:470-476): (b, 3*(steps+2), H, W) and (b, steps+2, H, W).

    Repair of SURVEY.md Appendix A-10, stated: the reference concatenates 8 channels for a network built with
    n_channels=10 (trainer.py:82) and calls an undefined self.netG.  The two missing channels are the edge maps the
    network was trained with (trainer.py:190-197): e = hed(frame)[5] on the frame in [0,1], i.e. on
    img * img_std + img_mean exactly as trainer.py:214-216 feeds its own prediction back to the edge net; channel
    order is the training input's.  The nets are forward-only twins of the training nets (shared weights) sized for
    this batch."""

    def __init__(self, engine: "ImageEngine", hed, batch: int, H: int, W: int):
        from .hned import HNEDHIP
        self.device, self.b, self.H, self.W = engine.device, batch, H, W
        src = engine.net
        self.net = GridNetHIP(10, batch, H, W, self.device, coord=src.coord, filters=src.filters, params_from=src)
        self.hed = HNEDHIP(batch, H, W, self.device, params_from=hed)
        arr = ctypes.c_float * 3
        self._mean, self._istd = arr(*OUT_MEAN), arr(*[1.0 / s for s in OUT_STD])
        # frame in [0,1] = img * img_std + img_mean  (trainer.py:215) as (img - shift) * scale
        self._unshift, self._unscale = arr(*[-m / s for m, s in zip(IMG_MEAN, IMG_STD)]), arr(*IMG_STD)
        self.x10 = torch.empty(batch, 10, H, W, dtype=torch.float32, device=self.device)
        self.frame01 = torch.empty(batch, 3, H, W, dtype=torch.float32, device=self.device)

    def _edges(self, img: torch.Tensor) -> torch.Tensor:
        s = torch.cuda.current_stream().cuda_stream
        call("vlg_affine_nchw", ptr(img), ptr(self.frame01), self.b, 3, self.H * self.W, self._unshift, self._unscale, s)
        return self.hed.forward(self.frame01)[5].contiguous()          # fused map (hned.py:105), (b,H,W)

    def run(self, img1, img2, seg1, seg2, steps: int = 8):
        b, H, W, dev = self.b, self.H, self.W, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        want = {"img": (b, 3, H, W), "seg": (b, 1, H, W)}
        for name, t, kind in (("img1", img1, "img"), ("img2", img2, "img"), ("seg1", seg1, "seg"), ("seg2", seg2, "seg")):
            if tuple(t.shape) != want[kind]:
                raise ValueError("%s must have shape %s, got %s" % (name, want[kind], tuple(t.shape)))
        img = [img1.to(**f32).contiguous(), img2.to(**f32).contiguous()]             # trainer.py:455-458
        seg = [seg1.to(**f32).contiguous(), seg2.to(**f32).contiguous()]
        edge = [self._edges(img[0]), self._edges(img[1])]
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(steps):                                                       # trainer.py:460
            call("vlg_rollout_input", ptr(edge[-2]), ptr(seg[-2]), ptr(img[-2]), ptr(img[-1]), ptr(seg[-1]), ptr(edge[-1]),
                 ptr(self.x10), b, H * W, s)
            seg_logits, img_raw = self.net.forward(self.x10)
            nxt = torch.empty(b, 3, H, W, **f32)
            call("vlg_affine_nchw", ptr(img_raw), ptr(nxt), b, 3, H * W, self._mean, self._istd, s)
            ids = torch.empty(b, 1, H, W, **f32)
            call("vlg_argmax_nchw", ptr(seg_logits), ptr(ids), b, seg_logits.shape[1], H * W, s)
            img.append(nxt)
            seg.append(ids)
            edge.append(self._edges(nxt))
        return torch.cat(img, dim=1), torch.cat(seg, dim=1)                          # trainer.py:470-471


def random_state(shapes, seed: int) -> Dict[str, torch.Tensor]:
    """torch's default initialisers for a state_dict of the given shapes - Conv2d weight and bias U(+-1/sqrt(fan_in)), PReLU
    slope 0.25 - from one seeded generator: what a net without a checkpoint starts from (benchmarks; Trainer draws the
    same distributions from the shared seed, reference src/main.py:57-60)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        if shp == (1,) and not k.endswith("bias"):
            out[k] = torch.full(shp, 0.25)
            continue
        wshape = shp if len(shp) == 4 else tuple(shapes.get(k[:-len("bias")] + "weight", (1, 64)))
        fan_in = 1
        for v in wshape[1:]:
            fan_in *= v
        out[k] = (torch.rand(shp, generator=g) * 2 - 1) / float(max(fan_in, 1)) ** 0.5
    return out


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
