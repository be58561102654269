"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's GridNet / CoordGridNet.

Functional torch-CPU restatement (F.conv2d / F.prelu / F.interpolate) of
    reference src/models/gridnet.py:7-58   (GridNet)      and :63-114 (CoordGridNet)
    reference src/models/modules.py:5-25   (LateralBlock),  :29-42 (DownSamplingBlock),
                                   :44-58  (UpSamplingBlock), :65-96 (AddCoords),
                                   :99-110 (CoordConv),      :115-135 (CoordLateralBlock)
operating on a plain {reference state_dict key: tensor} dict, so it can run where the reference
cannot travel (the GPU box).  PINNED: tests/test_oracle_golden.py checks it against outputs and
parameter gradients captured from the reference modules themselves (tests/golden/gridnet_*.npz,
written by oracle/make_golden.py).  Only tests/ import this file.

Also holds the deterministic parameter generator shared by make_golden.py and the tests, so the
fixtures need not store weights.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- structure
def block_list(n_col: int = 6) -> List[Tuple[str, str, int, int]]:
    """(name, kind, in_level, out_level) of every block, in the order the reference constructors create
    them is irrelevant here - this is the set; levels index filters_level.  Kinds: lateral | down | up."""
    out = [("down_00", "down", 0, 1), ("down_10", "down", 1, 2)]
    half = n_col // 2
    for i in range(1, half):
        out += [("lateral_0%d" % (i - 1), "lateral", 0, 0), ("down_0%d" % i, "down", 0, 1),
                ("down_1%d" % i, "down", 1, 2), ("lateral_1%d" % (i - 1), "lateral", 1, 1),
                ("lateral_2%d" % (i - 1), "lateral", 2, 2)]
    for i in range(half, n_col):
        out += [("lateral_2%d" % (i - 1), "lateral", 2, 2), ("lateral_1%d" % (i - 1), "lateral", 1, 1),
                ("lateral_0%d" % (i - 1), "lateral", 0, 0), ("up_1%d" % i, "up", 2, 1), ("up_0%d" % i, "up", 1, 0)]
    return out


def param_shapes(n_channels: int = 10, filters=(32, 64, 96), seg_out: int = 20, img_out: int = 3,
                 coord: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """Reference state_dict keys and shapes (checked against tests/golden/gridnet_keys.json)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(key, cin, cout):
        s[key + ".weight"] = (cout, cin, 3, 3)
        s[key + ".bias"] = (cout,)

    f = filters
    if coord:       # CoordLateralBlock, modules.py:115-135 (no leading PReLU, CoordConv adds 2 channels)
        conv("lateral_in.conv.0.conv", n_channels + 2, f[0])
        s["lateral_in.conv.1.weight"] = (1,)
        conv("lateral_in.conv.2.conv", f[0] + 2, f[0])
        conv("lateral_in.conv2.conv", n_channels + 2, f[0])
    else:           # LateralBlock with shortcut conv, modules.py:5-25
        s["lateral_in.conv.0.weight"] = (1,)
        conv("lateral_in.conv.1", n_channels, f[0])
        s["lateral_in.conv.2.weight"] = (1,)
        conv("lateral_in.conv.3", f[0], f[0])
        conv("lateral_in.conv2", n_channels, f[0])
    for name, cout in (("lateral_out_seg", seg_out), ("lateral_out_img", img_out)):
        s[name + ".conv.0.weight"] = (1,)
        conv(name + ".conv.1", f[0], cout)
        s[name + ".conv.2.weight"] = (1,)
        conv(name + ".conv.3", cout, cout)
    for name, kind, li, lo in block_list():
        if kind == "up":        # [Upsample, PReLU, Conv, PReLU, Conv], modules.py:49-55
            s[name + ".up.1.weight"] = (1,)
            conv(name + ".up.2", f[li], f[lo])
            s[name + ".up.3.weight"] = (1,)
            conv(name + ".up.4", f[lo], f[lo])
        else:                   # [PReLU, Conv, PReLU, Conv], modules.py:12-17 / 34-39
            s[name + ".conv.0.weight"] = (1,)
            conv(name + ".conv.1", f[li], f[lo])
            s[name + ".conv.2.weight"] = (1,)
            conv(name + ".conv.3", f[lo], f[lo])
    return s


def test_params(shapes: "Dict[str, Tuple[int, ...]]", seed: int = 0, linear: bool = False) -> Dict[str, torch.Tensor]:
    """Deterministic, name-keyed values: conv weights/biases ~ U(+-1/sqrt(fan_in)), every PReLU slope a
    different value in (0.1, 0.4) so a mixed-up slope cannot go unnoticed.

    linear=True sets every slope to exactly 1 (PReLU == identity).  Why that mode exists: the gradient of a
    PReLU network is DISCONTINUOUS in its pre-activations, so two fp32 implementations whose forward values
    differ in the last bits (here: 4e-7 relative) disagree by O(1) on every element whose pre-activation lies
    inside that rounding band - a handful out of the ~5 M activations of a 64x64 full-width GridNet - and
    every gradient downstream moves by ~1e-3.  With slopes = 1 the network is smooth, so ALL gradients
    (including d/d slope, whose integrand x*[x<=0] is continuous) can be compared at 1e-4; with real slopes the
    forward is compared at 1e-4 and gradients with a kink-tolerant bound.  (Choosing inputs that stay clear of
    the kinks is not possible at useful sizes: the smallest |pre-activation| over a 2x32x48 small-width net is
    already ~1e-6 for every seed tried.)"""
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7fffffff)
        if len(shape) == 4:
            bound = 1.0 / math.sqrt(shape[1] * 9)
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif shape == (1,):
            out[name] = torch.ones(shape) if linear else torch.rand(shape, generator=g) * 0.3 + 0.1
        else:
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
    return out


# ------------------------------------------------------------------------------- forward
def add_coords(x: torch.Tensor) -> torch.Tensor:
    """AddCoords (modules.py:65-96): two channels, the first varying along H, the second along W, values
    k/(dim-1)*2-1 (the reference hard-codes 256; identical there)."""
    b, _, H, W = x.shape
    yy = (torch.arange(H, dtype=torch.float32) / (H - 1)) * 2 - 1
    xx = (torch.arange(W, dtype=torch.float32) / (W - 1)) * 2 - 1
    return torch.cat([x, yy[None, None, :, None].expand(b, 1, H, W), xx[None, None, None, :].expand(b, 1, H, W)], dim=1)


def _conv(p, key, x, stride=1):
    return F.conv2d(x, p[key + ".weight"], p[key + ".bias"], stride=stride, padding=1)


class Branches:
    """Optional record / override of which side of its kink every PReLU input fell on.

    `record` (dict) receives {PReLU key: the input tensor}.  `positive` ({PReLU key: bool tensor}) REPLACES the
    test x > 0 of that PReLU: y = x where positive, slope * x elsewhere, so autograd's derivatives follow the
    given pattern.  Used to compare backward passes of two fp32 implementations exactly: given the SAME branch
    pattern a PReLU network's backward is a smooth function of everything else, so the 1e-4 bound applies to every
    gradient; the pattern itself is compared separately (it may differ only where |x| sits inside the rounding band
    of the forward pass).  With neither, F.prelu runs untouched (the path pinned bit for bit to the reference)."""

    def __init__(self, positive=None, record=None):
        self.positive, self.record = positive, record


def _prelu(p, key, x, br):
    if br is not None and br.record is not None:
        br.record[key] = x.detach()
    if br is None or br.positive is None:
        return F.prelu(x, p[key])
    return torch.where(br.positive[key], x, p[key] * x)


def _block(p, name, kind, x, br=None):
    if kind == "up":
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)         # modules.py:50
        x = _conv(p, name + ".up.2", _prelu(p, name + ".up.1.weight", x, br))
        return _conv(p, name + ".up.4", _prelu(p, name + ".up.3.weight", x, br))
    x = _conv(p, name + ".conv.1", _prelu(p, name + ".conv.0.weight", x, br), stride=2 if kind == "down" else 1)
    return _conv(p, name + ".conv.3", _prelu(p, name + ".conv.2.weight", x, br))


def forward(p: Dict[str, torch.Tensor], x: torch.Tensor, coord: bool = False, n_col: int = 6, branches: "Branches" = None):
    """GridNet.forward (gridnet.py:43-58) / CoordGridNet.forward (:99-114).  Returns (seg, img)."""
    br = branches
    if coord:
        xc = add_coords(x)
        h = _prelu(p, "lateral_in.conv.1.weight", _conv(p, "lateral_in.conv.0.conv", xc), br)
        x0 = _conv(p, "lateral_in.conv.2.conv", add_coords(h)) + _conv(p, "lateral_in.conv2.conv", xc)
    else:
        x0 = _block(p, "lateral_in", "lateral", x, br) + _conv(p, "lateral_in.conv2", x)
    x1 = _block(p, "down_00", "down", x0, br)
    x2 = _block(p, "down_10", "down", x1, br)
    for i in range(1, n_col):
        if i < n_col / 2:
            x0 = _block(p, "lateral_0%d" % (i - 1), "lateral", x0, br)
            x1 = _block(p, "down_0%d" % i, "down", x0, br) + _block(p, "lateral_1%d" % (i - 1), "lateral", x1, br)
            x2 = _block(p, "down_1%d" % i, "down", x1, br) + _block(p, "lateral_2%d" % (i - 1), "lateral", x2, br)
        else:
            x2 = _block(p, "lateral_2%d" % (i - 1), "lateral", x2, br)
            x1 = _block(p, "up_1%d" % i, "up", x2, br) + _block(p, "lateral_1%d" % (i - 1), "lateral", x1, br)
            x0 = _block(p, "up_0%d" % i, "up", x1, br) + _block(p, "lateral_0%d" % (i - 1), "lateral", x0, br)
    return _block(p, "lateral_out_seg", "lateral", x0, br), _block(p, "lateral_out_img", "lateral", x0, br)


def forward_backward(p, x, r_seg, r_img, coord=False, branches: "Branches" = None):
    """loss = sum(seg * r_seg) + sum(img * r_img); returns (seg, img, {key: grad}, dx)."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    xin = x.detach().clone().requires_grad_(True)
    seg, img = forward(q, xin, coord, branches=branches)
    ((seg * r_seg).sum() + (img * r_img).sum()).backward()
    return seg.detach(), img.detach(), {k: v.grad for k, v in q.items()}, xin.grad
