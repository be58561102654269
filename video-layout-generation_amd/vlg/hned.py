"""The reference's frozen HED edge detector (reference src/models/hned.py:9-105), forward only, on gfx950.

VGG16-style trunk of 13 ReLU-conv3x3 in five stages with 2x2 max-pools between them, five 1x1 score
convolutions, bilinear resize of every score map to the input size (align_corners=False), sigmoids and a
1x1 fusion (hned.py:73-105).  The 3x3 convolutions are vlg_conv3x3_fwd launches (csrc/conv.hip) with the ReLU
applied by the CONSUMER on load (PReLU slope 0), pools / score convs / the fused head are small HBM-bound
kernels (csrc/gridnet_ops.hip).  The reference calls this net under no_grad three times per step
(src/trainer.py:190-192,214-216) and uses the 6th output (the fused map) as the edge channel (Appendix A-3).

State-dict keys and shapes are the reference's (moduleVggOne.0.weight ... moduleCombine.0.bias), so the
authors' trained checkpoint (trainer.py:97-99, key 'generator') would load unchanged; it is not in the
repository, so tests pin structure and arithmetic with name-seeded weights.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List

import torch

from . import hip
from .gridnet import _Geo, _PT
from .hip import CEPI_CIN4, call, ptr

STAGES = (("moduleVggOne", 3, 64, (0, 2)), ("moduleVggTwo", 64, 128, (1, 3)), ("moduleVggThr", 128, 256, (1, 3, 5)),
          ("moduleVggFou", 256, 512, (1, 3, 5)), ("moduleVggFiv", 512, 512, (1, 3, 5)))
SCORES = ("moduleScoreOne", "moduleScoreTwo", "moduleScoreThr", "moduleScoreFou", "moduleScoreFiv")
BGR_MEAN = (104.00698793, 116.66876762, 122.67891434)      # hned.py:74-76 (applied to channels 0,1,2 as given)


class HNEDHIP:
    def __init__(self, batch: int, H: int, W: int, device, params_from: "HNEDHIP" = None):
        """params_from: twin for another batch / image size reading that instance's weights (no copy)."""
        if H % 16 or W % 16:
            raise ValueError("H and W must be divisible by 16 (four 2x2 max-pools)")
        hip.load()
        if device.type != "cuda":
            raise hip.HipError("HNEDHIP needs a HIP device; there is no CPU path")
        self.device, self.b, self.H, self.W = device, batch, H, W
        self.geo = [_Geo(batch, H >> k, W >> k, device) for k in range(5)]
        self.pre = torch.empty(batch, 3, H, W, dtype=torch.float32, device=device)
        self.x = _PT(self.geo[0], 3, device)
        self.convs: List[tuple] = []          # (key, in tensor, out tensor, cin, cout, relu_on_load)
        self.pools: Dict[int, tuple] = {}
        self.feats: List[_PT] = []
        t = self.x
        for si, (name, cin, cout, idx) in enumerate(STAGES):
            if si > 0:
                pooled = _PT(self.geo[si], cin, device)
                self.pools[si] = (t, pooled)
                t = pooled
            c = cin
            for j, i in enumerate(idx):
                o = _PT(self.geo[si], cout, device)
                self.convs.append(("%s.%d" % (name, i), t, o, c, cout, not (si == 0 and j == 0), si))
                t, c = o, cout
            self.feats.append(t)
        off = 0
        self.off: Dict[str, int] = {}
        for key, tin, tout, cin, cout, _, _ in self.convs:
            self.off[key + ".weight"] = off
            off += tout.cp * 9 * tin.cp
            self.off[key + ".bias"] = off
            off += tout.cp
        for name, (_, _, cout, _) in zip(SCORES, STAGES):
            self.off[name + ".weight"] = off
            off += cout
            self.off[name + ".bias"] = off
            off += 4
        self.off["moduleCombine.0.weight"] = off
        off += 8
        self.off["moduleCombine.0.bias"] = off
        off += 4
        self.off["_zero"] = off                       # ReLU = PReLU with slope 0
        off += 4
        self.params = torch.zeros(off, dtype=torch.float32, device=device) if params_from is None else params_from.params
        self.score = [torch.empty(batch, H >> k, W >> k, dtype=torch.float32, device=device) for k in range(5)]
        # split-K workspace (coarse levels: every tile; elsewhere the tiles beyond the last full round of 256 - csrc/conv.hip)
        lib = hip.load()
        need = max([lib.vlg_conv3x3_fwd_workspace(tout.geo.rows, tin.cp, cout, tout.cp) for _, tin, tout, _, cout, _, _ in self.convs] + [0])
        self.ws = torch.empty(need, dtype=torch.float32, device=device) if need else None
        self.ws_n = need
        arr = ctypes.c_float * 3
        self._shift, self._scale = arr(*[m / 255.0 for m in BGR_MEAN]), arr(255.0, 255.0, 255.0)

    def reference_shapes(self):
        s = {}
        for key, tin, tout, cin, cout, _, _ in self.convs:
            s[key + ".weight"], s[key + ".bias"] = (cout, cin, 3, 3), (cout,)
        for name, (_, _, cout, _) in zip(SCORES, STAGES):
            s[name + ".weight"], s[name + ".bias"] = (1, cout, 1, 1), (1,)
        s["moduleCombine.0.weight"], s["moduleCombine.0.bias"] = (1, 5, 1, 1), (1,)
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        for k, shp in self.reference_shapes().items():
            if k not in sd or tuple(sd[k].shape) != tuple(shp):
                raise ValueError("state_dict entry %s missing or of shape %s (want %s)" % (k, tuple(sd[k].shape) if k in sd else None, shp))
        self.params.zero_()
        for key, tin, tout, cin, cout, _, _ in self.convs:
            wp = torch.zeros(tout.cp, 9, tin.cp)
            wp[:cout, :, :cin] = sd[key + ".weight"].to(torch.float32).permute(0, 2, 3, 1).reshape(cout, 9, cin)
            o = self.off[key + ".weight"]
            self.params[o:o + wp.numel()].copy_(wp.flatten())
            o = self.off[key + ".bias"]
            self.params[o:o + cout].copy_(sd[key + ".bias"].to(torch.float32))
        for name in SCORES + ("moduleCombine.0",):
            for part in (".weight", ".bias"):
                v = sd[name + part].to(torch.float32).flatten()
                o = self.off[name + part]
                self.params[o:o + v.numel()].copy_(v)

    def _pp(self, key: str) -> int:
        return self.params.data_ptr() + 4 * self.off[key]

    def forward(self, frames: torch.Tensor) -> torch.Tensor:
        """frames (b,3,H,W) in [0,1] -> (6,b,H,W): d1..d5 and the fused edge map (hned.py:105)."""
        b, H, W = self.b, self.H, self.W
        if tuple(frames.shape) != (b, 3, H, W) or not frames.is_cuda or frames.dtype != torch.float32:
            raise ValueError("frames must be a float32 HIP tensor of shape %s" % ((b, 3, H, W),))
        s = torch.cuda.current_stream().cuda_stream
        frames = frames.contiguous()
        call("vlg_affine_nchw", ptr(frames), ptr(self.pre), b, 3, H * W, self._shift, self._scale, s)     # hned.py:74-78
        call("vlg_nchw_to_padded", ptr(self.pre), self.x.ptr, b, 3, H, W, self.x.cp, -1, s)
        done_pool = set()
        for key, tin, tout, cin, cout, relu, si in self.convs:
            if si in self.pools and si not in done_pool:
                src, dst = self.pools[si]
                call("vlg_maxpool2x2", src.ptr, dst.ptr, b, dst.geo.H, dst.geo.W, src.cp, s)
                done_pool.add(si)
            g = tout.geo
            call("vlg_conv3x3_fwd", tin.ptr, self._pp(key + ".weight"), self._pp(key + ".bias"), tout.ptr, 0, ptr(g.mask),
                 self._pp("_zero") if relu else 0, 0, g.rows, tin.cp, cout, tout.cp, g.wp, tin.cp,
                 CEPI_CIN4 if (cin <= 4 and not relu) else 0, ptr(self.ws), self.ws_n, s)
        for k, (name, f) in enumerate(zip(SCORES, self.feats)):
            call("vlg_score1x1_relu", f.ptr, self._pp(name + ".weight"), self._pp(name + ".bias"), ptr(self.score[k]), b,
                 f.geo.H, f.geo.W, f.C, f.cp, s)
        out = torch.empty(6, b, H, W, dtype=torch.float32, device=self.device)
        call("vlg_hed_head", *[ptr(m) for m in self.score], self._pp("moduleCombine.0.weight"), self._pp("moduleCombine.0.bias"),
             ptr(out), b, H, W, s)
        return out
