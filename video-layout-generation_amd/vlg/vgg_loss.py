"""VggLoss (reference src/loss.py:29-49) on gfx950: L1 between frozen VGG19 relu4_4 features of two images.

    model = torchvision.models.vgg19(pretrained=True); self.features = Sequential(*children[:-10])   (loss.py:33-38)
    loss  = (features(output) - features(target)).abs().mean()                                         (loss.py:43-47)

i.e. features[0..26]: conv3x3+ReLU x2 @64, pool, x2 @128, pool, x4 @256, pool, x4 @512 (10 585 152 frozen
parameters).  Forward runs on vlg_conv3x3_fwd (ReLU applied by the consumer on load, max-pool on the
pre-activation since ReLU commutes with max), the input gradient on vlg_conv3x3_dgrad (ReLU' fused in its epilogue)
and vlg_maxpool2x2_bwd; no weight gradients exist (the trunk is frozen, loss.py:40-41).

PARITY UNPINNED against the reference for this term: torchvision is absent here and the ImageNet weights are a
download, so neither the reference class nor its weights can be run.  State-dict keys and
shapes are torchvision's (features.0.weight ...), so real weights load unchanged; tests pin the arithmetic against
a torch-CPU restatement with name-seeded weights.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import hip
from .gridnet import _Geo, _PT
from .hip import CEPI_CIN4, CEPI_DPRELU, call, ptr

CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512)     # torchvision vgg19 cfg 'E', first 27 modules


def conv_keys():
    """[(features index, cin, cout)] of the 12 convolutions inside features[:27]."""
    out, idx, cin = [], 0, 3
    for v in CFG:
        if v == "M":
            idx += 1
        else:
            out.append((idx, cin, v))
            cin = v
            idx += 2
    return out


class VggLossHIP:
    def __init__(self, batch: int, H: int, W: int, device):
        if H % 8 or W % 8:
            raise ValueError("H and W must be divisible by 8 (three 2x2 max-pools)")
        hip.load()
        if device.type != "cuda":
            raise hip.HipError("VggLossHIP needs a HIP device; there is no CPU path")
        self.device, self.b, self.H, self.W = device, batch, H, W
        self.geo = [_Geo(batch, H >> k, W >> k, device) for k in range(4)]
        self.x = _PT(self.geo[0], 3, device)
        self.ops: List[tuple] = []            # ("conv", key, tin, tout, cin, cout, relu_on_load) | ("pool", tin, tout)
        t, level, cin, idx = self.x, 0, 3, 0
        first = True
        for v in CFG:
            if v == "M":
                level += 1
                o = _PT(self.geo[level], cin, device)
                self.ops.append(("pool", t, o))
                t = o
                idx += 1
            else:
                o = _PT(self.geo[level], v, device)
                self.ops.append(("conv", "features.%d" % idx, t, o, cin, v, not first))
                t, cin, first = o, v, False
                idx += 2
        self.feat = t
        self.feat_tgt = _PT(self.feat.geo, self.feat.C, device)
        for op in self.ops:                   # gradient buffers (one consumer per tensor: no accumulation)
            tin = op[2] if op[0] == "conv" else op[1]
            tin.grad = _PT(tin.geo, tin.C, device)
        self.feat.grad = _PT(self.feat.geo, self.feat.C, device)
        off = 0
        self.off: Dict[str, int] = {}
        for op in self.ops:
            if op[0] == "conv":
                _, key, tin, tout, ci, co, _ = op
                self.off[key + ".weight"] = off
                off += tout.cp * 9 * tin.cp
                self.off[key + ".bias"] = off
                off += tout.cp
        self.off["_zero"] = off
        off += 4
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.scratch = torch.zeros(4096, dtype=torch.float32, device=device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)
        # split-K workspace (coarse levels: every tile; elsewhere the tiles beyond the last full round of 256 - csrc/conv.hip)
        lib = hip.load()
        need = max([lib.vlg_conv3x3_fwd_workspace(op[3].geo.rows, op[2].cp, op[5], op[3].cp) for op in self.ops if op[0] == "conv"] +
                   [lib.vlg_conv3x3_dgrad_workspace(op[2].geo.rows, op[2].cp, op[3].cp) for op in self.ops if op[0] == "conv"] + [0])
        self.ws = torch.empty(need, dtype=torch.float32, device=device) if need else None
        self.ws_n = need

    def reference_shapes(self):
        s = {}
        for op in self.ops:
            if op[0] == "conv":
                s[op[1] + ".weight"], s[op[1] + ".bias"] = (op[5], op[4], 3, 3), (op[5],)
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        self.params.zero_()
        for op in self.ops:
            if op[0] != "conv":
                continue
            _, key, tin, tout, ci, co, _ = op
            w = sd[key + ".weight"].to(torch.float32)
            if tuple(w.shape) != (co, ci, 3, 3):
                raise ValueError("%s.weight has shape %s, expected %s" % (key, tuple(w.shape), (co, ci, 3, 3)))
            wp = torch.zeros(tout.cp, 9, tin.cp)
            wp[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
            o = self.off[key + ".weight"]
            self.params[o:o + wp.numel()].copy_(wp.flatten())
            o = self.off[key + ".bias"]
            self.params[o:o + co].copy_(sd[key + ".bias"].to(torch.float32))

    def _pp(self, key: str) -> int:
        return self.params.data_ptr() + 4 * self.off[key]

    def _features(self, img: torch.Tensor, s: int) -> None:
        b, H, W = self.b, self.H, self.W
        img = img.contiguous()
        call("vlg_nchw_to_padded", ptr(img), self.x.ptr, b, 3, H, W, self.x.cp, -1, s)
        for op in self.ops:
            if op[0] == "pool":
                _, tin, tout = op
                call("vlg_maxpool2x2", tin.ptr, tout.ptr, b, tout.geo.H, tout.geo.W, tin.cp, s)
            else:
                _, key, tin, tout, ci, co, relu = op
                g = tout.geo
                # (the image layer: 3 channels in a 32-channel padded tensor - contraction over (tap, 4 channels))
                call("vlg_conv3x3_fwd", tin.ptr, self._pp(key + ".weight"), self._pp(key + ".bias"), tout.ptr, 0, ptr(g.mask),
                     self._pp("_zero") if relu else 0, 0, g.rows, tin.cp, co, tout.cp, g.wp, tin.cp,
                     CEPI_CIN4 if (ci <= 4 and not relu) else 0, ptr(self.ws), self.ws_n, s)

    def preactivations(self) -> Dict[str, torch.Tensor]:
        """Every convolution's output BEFORE its ReLU, as (b, C, h, w) tensors keyed by the torchvision module name
        ("features.<i>"), for the image the trunk ran LAST (loss_and_grad: `output`), plus "target" = the kept
        pre-activation features of the target.  These fix every kink decision the backward took - ReLU' = [pre > 0], the
        2x2 max-pool's argmax (taken on the pre-activation: ReLU commutes with max) and the sign of the final L1 - so a
        test can evaluate a smooth fp64 restatement ON that pattern (tests/test_hip_vgg.py::test_vgg_gradient_strict_given_pattern)."""
        s = torch.cuda.current_stream().cuda_stream
        out = {}

        def nchw(t: _PT) -> torch.Tensor:
            o = torch.empty(self.b, t.C, t.geo.H, t.geo.W, dtype=torch.float32, device=self.device)
            call("vlg_padded_to_nchw", t.ptr, ptr(o), self.b, t.C, t.geo.H, t.geo.W, t.cp, s)
            return o
        for op in self.ops:
            if op[0] == "conv":
                out[op[1]] = nchw(op[3])
        out["target"] = nchw(self.feat_tgt)
        return out

    def loss_and_grad(self, output: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0, want_grad: bool = True):
        """Returns (loss[1] device tensor, d(grad_scale * loss)/d output as (b,3,H,W) or None)."""
        b, H, W = self.b, self.H, self.W
        s = torch.cuda.current_stream().cuda_stream
        self._features(target, s)
        call("vlg_add_rows", self.feat_tgt.ptr, self.feat.ptr, self.feat.n, 0, s)          # keep the target's features
        self._features(output, s)
        f = self.feat
        count = b * f.C * f.geo.H * f.geo.W
        call("vlg_l1_relu_padded", f.ptr, self.feat_tgt.ptr, f.grad.ptr if want_grad else 0, ptr(self.loss), ptr(self.scratch),
             f.geo.rows, f.cp, count, grad_scale, s)
        if not want_grad:
            return self.loss, None
        for op in reversed(self.ops):
            if op[0] == "pool":
                _, tin, tout = op
                call("vlg_maxpool2x2_bwd", tin.ptr, tout.grad.ptr, tin.grad.ptr, b, tout.geo.H, tout.geo.W, tin.cp, s)
            else:
                _, key, tin, tout, ci, co, relu = op
                g = tin.geo
                call("vlg_conv3x3_dgrad", tout.grad.ptr, self._pp(key + ".weight"), tin.grad.ptr, tin.ptr, ptr(g.mask),
                     self._pp("_zero") if relu else 0, 0, 0, 0, g.rows, tin.cp, tout.cp, g.wp, tin.cp,
                     CEPI_DPRELU if relu else 0, ptr(self.ws), self.ws_n, 0, s)
        dimg = torch.empty(b, 3, H, W, dtype=torch.float32, device=self.device)
        call("vlg_padded_to_nchw", self.x.grad.ptr, ptr(dimg), b, 3, H, W, self.x.cp, s)
        return self.loss, dimg
