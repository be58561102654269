"""GridNet / CoordGridNet (the reference's trainable model) on the gfx950 convolution kernels.

Restates the structure of reference src/models/gridnet.py:7-58 (GridNet) and :63-114 (CoordGridNet) -
3 rows x 6 columns of PReLU->conv3x3->PReLU->conv3x3 blocks (reference src/models/modules.py:5-58),
stride-2 convs going down, bilinear x2 going up, two heads - as a static tape of launches of
vlg_conv3x3_{fwd,dgrad,wgrad}, vlg_upsample2x_{fwd,bwd} over halo-padded channels-last tensors
(csrc/conv.hip).  Forward and backward are explicit; there is no autograd.

Parameters keep the reference's state_dict names and shapes at the boundary (load_state_dict /
state_dict convert to and from the kernels' [cout_p][9][cin_p] layout), so a reference checkpoint's
'gridnet' entry loads unchanged (reference src/trainer.py:85-92).

    net = GridNetHIP(n_channels=10, batch=4, H=256, W=256, device=dev, coord=True)
    seg, img = net.forward(x)                 # x (b,10,H,W) NCHW -> (b,20,H,W), (b,3,H,W)
    net.backward(dseg, dimg)                  # fills net.grads; net.named_grads() in reference shapes
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from . import hip
from .hip import CEPI_ACCUM, CEPI_DPRELU, CEPI_RESID, call, ptr


def _ceil32(c: int) -> int:
    return (c + 31) // 32 * 32


def reference_param_order(coord: bool) -> List[str]:
    """Keys in the order the reference constructors register their parameters (reference src/models/gridnet.py:19-39 /
    :75-95, src/models/modules.py:10-25,34-39,49-55,118-135) = module.parameters() order = the index space of the
    torch.optim.Adam state_dict the reference's get_gridnet builds and loads (src/trainer.py:83,91-92)."""
    def blk(name: str, kind: str = "lateral") -> List[str]:
        seq, i = ("up", (1, 2, 3, 4)) if kind == "up" else ("conv", (0, 1, 2, 3))
        return ["%s.%s.%d.weight" % (name, seq, i[0]), "%s.%s.%d.weight" % (name, seq, i[1]), "%s.%s.%d.bias" % (name, seq, i[1]),
                "%s.%s.%d.weight" % (name, seq, i[2]), "%s.%s.%d.weight" % (name, seq, i[3]), "%s.%s.%d.bias" % (name, seq, i[3])]
    if coord:
        out = ["lateral_in.conv.0.conv.weight", "lateral_in.conv.0.conv.bias", "lateral_in.conv.1.weight",
               "lateral_in.conv.2.conv.weight", "lateral_in.conv.2.conv.bias", "lateral_in.conv2.conv.weight",
               "lateral_in.conv2.conv.bias"]
    else:
        out = blk("lateral_in") + ["lateral_in.conv2.weight", "lateral_in.conv2.bias"]
    out += blk("lateral_out_seg") + blk("lateral_out_img") + blk("down_00") + blk("down_10")
    for i in (1, 2):
        out += blk("lateral_0%d" % (i - 1)) + blk("down_0%d" % i) + blk("down_1%d" % i) + blk("lateral_1%d" % (i - 1)) + \
            blk("lateral_2%d" % (i - 1))
    for i in (3, 4, 5):
        out += blk("lateral_2%d" % (i - 1)) + blk("lateral_1%d" % (i - 1)) + blk("lateral_0%d" % (i - 1)) + \
            blk("up_1%d" % i, "up") + blk("up_0%d" % i, "up")
    return out


class _Geo:
    """One resolution level: padded geometry, interior mask, stride-2 tables to the next coarser level."""

    def __init__(self, b: int, H: int, W: int, device):
        self.b, self.H, self.W = b, H, W
        self.wp = W + 2
        self.rows = b * (H + 2) * (W + 2)
        self.guard = self.wp + 40                    # >= 32 rows of K-tile overrun + one row shift + 1
        m = torch.zeros(b, H + 2, W + 2)
        m[:, 1:H + 1, 1:W + 1] = 1.0
        self.mask = torch.cat([m.flatten(), torch.zeros(64)]).to(device)
        self.down_rowtab = None                      # rows of the coarser level -> centre rows here
        self.down_taptabs = None                     # [9][rows here] -> rows of the coarser level (or -1)

    def link_coarser(self, coarse: "_Geo", device) -> None:
        b, H, W, h, w = self.b, self.H, self.W, coarse.H, coarse.W
        n = torch.arange(b).view(b, 1, 1)
        # forward / weight gradient: coarse interior (y',x') reads the fine window centred at (2y'-1, 2x'-1)
        yc = torch.arange(h + 2).view(1, h + 2, 1)
        xc = torch.arange(w + 2).view(1, 1, w + 2)
        centre = (n * (H + 2) + (2 * yc - 1)) * (W + 2) + (2 * xc - 1)
        interior = (yc >= 1) & (yc <= h) & (xc >= 1) & (xc <= w)
        tab = torch.where(interior, centre, torch.zeros_like(centre)).flatten()
        self.down_rowtab = torch.cat([tab, torch.zeros(64, dtype=tab.dtype)]).to(torch.int32).to(device)
        # data gradient: fine interior pixel (Y',X') gets tap (ky,kx) from coarse (y,x) with 2y+ky-1 = Y'-1
        yf = torch.arange(H + 2).view(1, H + 2, 1)
        xf = torch.arange(W + 2).view(1, 1, W + 2)
        fin = (yf >= 1) & (yf <= H) & (xf >= 1) & (xf <= W)
        tabs = []
        for ky in range(3):
            for kx in range(3):
                ny, nx = yf - ky, xf - kx             # = 2y, 2x  (Y'-1+1-ky)
                ok = fin & (ny % 2 == 0) & (nx % 2 == 0) & (ny >= 0) & (nx >= 0) & (ny // 2 < h) & (nx // 2 < w)
                row = (n * (h + 2) + (ny // 2 + 1)) * (w + 2) + (nx // 2 + 1)
                tabs.append(torch.where(ok, row, torch.full_like(row, -1)).flatten())
        self.down_taptabs = torch.stack(tabs).to(torch.int32).contiguous().to(device)


class _PT:
    """Padded channels-last activation (or its gradient): zero halo, zero guard rows, Cp = ceil32(C)."""

    def __init__(self, geo: _Geo, C: int, device, coord: bool = False):
        self.geo, self.C = geo, C
        self.cp = _ceil32(C + (2 if coord else 0))
        self.coord_c0 = C if coord else -1
        self.buf = torch.zeros((geo.rows + 2 * geo.guard) * self.cp, dtype=torch.float32, device=device)
        self.ptr = self.buf.data_ptr() + 4 * geo.guard * self.cp
        self.n = geo.rows * self.cp
        self.grad: Optional["_PT"] = None
        self.grad_written = False


class _Conv:
    def __init__(self, key, x, out, cin, cout, stride, prelu, resid, act_ch):
        self.key, self.x, self.out, self.cin, self.cout = key, x, out, cin, cout
        self.stride, self.prelu, self.resid, self.act_ch = stride, prelu, resid, act_ch
        self.w_off = self.b_off = 0


class GridNetHIP:
    TAIL_EXTRA = 8

    def __init__(self, n_channels: int, batch: int, H: int, W: int, device, coord: bool = False,
                 filters=(32, 64, 96), seg_out: int = 20, img_out: int = 3, need_input_grad: bool = False,
                 params_from: Optional["GridNetHIP"] = None):
        """params_from: build a FORWARD-ONLY twin of another instance (other batch / image size) that reads that
        instance's parameter buffer - the rollout and validation-time shapes share the training weights, no copy."""
        if H % 4 or W % 4:
            raise ValueError("H and W must be divisible by 4 (two stride-2 levels)")
        hip.load()
        if device.type != "cuda":
            raise hip.HipError("GridNetHIP needs a HIP device; there is no CPU path")
        self.device, self.coord, self.filters = device, coord, tuple(filters)
        self.n_channels, self.seg_out, self.img_out = n_channels, seg_out, img_out
        self.need_input_grad = need_input_grad
        self.geo = [_Geo(batch, H >> l, W >> l, device) for l in range(3)]
        self.geo[0].link_coarser(self.geo[1], device)
        self.geo[1].link_coarser(self.geo[2], device)
        self.tape: List[object] = []
        self.tensors: List[_PT] = []
        self.prelu_keys: List[str] = []
        f = self.filters
        # ---- build the static graph in the reference's forward order (gridnet.py:43-58 / :99-114)
        # `_group` tags every convolution with the gradient bucket it belongs to: backward finishes the heads first, then
        # grid columns 5 .. 1, then the input blocks - the order data-parallel buckets are handed to the reducer
        self._group = "in"
        self.x = self._tensor(0, n_channels, coord=coord)
        if coord:      # CoordLateralBlock: [CoordConv, PReLU, CoordConv] + CoordConv shortcut (modules.py:115-135)
            t = self._conv("lateral_in.conv.0.conv", self.x, f[0], out_coord=True)
            s = self._conv("lateral_in.conv2.conv", self.x, f[0])
            x0 = self._conv("lateral_in.conv.2.conv", t, f[0], prelu="lateral_in.conv.1.weight", resid=s, act_ch=f[0])
        else:          # LateralBlock with shortcut conv (modules.py:5-25)
            s = self._conv("lateral_in.conv2", self.x, f[0])
            x0 = self._block("lateral_in", "lateral", self.x, f[0], resid=s)
        x1 = self._block("down_00", "down", x0, f[1])
        x2 = self._block("down_10", "down", x1, f[2])
        for i in range(1, 6):
            self._group = "col%d" % i
            if i < 3:
                x0 = self._block("lateral_0%d" % (i - 1), "lateral", x0, f[0])
                d = self._block("down_0%d" % i, "down", x0, f[1])
                x1 = self._block("lateral_1%d" % (i - 1), "lateral", x1, f[1], resid=d)
                d = self._block("down_1%d" % i, "down", x1, f[2])
                x2 = self._block("lateral_2%d" % (i - 1), "lateral", x2, f[2], resid=d)
            else:
                x2 = self._block("lateral_2%d" % (i - 1), "lateral", x2, f[2])
                u = self._block("up_1%d" % i, "up", x2, f[1])
                x1 = self._block("lateral_1%d" % (i - 1), "lateral", x1, f[1], resid=u)
                u = self._block("up_0%d" % i, "up", x1, f[0])
                x0 = self._block("lateral_0%d" % (i - 1), "lateral", x0, f[0], resid=u)
        self._group = "head"
        self.seg = self._block("lateral_out_seg", "lateral", x0, seg_out)
        self.img = self._block("lateral_out_img", "lateral", x0, img_out)
        self._alloc_params(params_from)
        lib = hip.load()
        # workspace of the forward convolutions (tail split: tiles beyond the last full round of 256 CUs; split-K of wide
        # convolutions on small grids, csrc/conv.hip) - sized over EVERY convolution forward() hands it to
        self.ws_n = max([lib.vlg_conv3x3_fwd_workspace(c.out.geo.rows, c.x.cp, c.cout, c.out.cp)
                         for c in self.tape if isinstance(c, _Conv)] + [0])
        self.ws = torch.empty(self.ws_n, dtype=torch.float32, device=device) if self.ws_n else None
        self.forward_only = params_from is not None
        if self.forward_only:
            return
        # Backward scratch: every convolution owns a region of the slab arena (weight-gradient partials) and, if a PReLU
        # precedes it, of the slope-gradient arena; two table-driven launches at the end of backward() reduce them all
        # (vlg_reduce_slabs_table / vlg_sum_partials_table) instead of two tiny launches per convolution.
        convs = [op for op in self.tape if isinstance(op, _Conv)]
        off = da_off = 0
        for c in convs:
            c.n_slabs = lib.vlg_conv3x3_wgrad_slabs(c.out.geo.rows, c.x.cp, c.out.cp)
            c.slab_stride = c.out.cp * 9 * c.x.cp + c.out.cp
            c.slab_off = off
            off += (c.n_slabs * c.slab_stride + 3) // 4 * 4
            c.da_n = lib.vlg_conv3x3_dgrad_slabs(c.x.geo.rows, c.x.cp) if c.prelu else 0
            c.da_off = da_off
            da_off += (c.da_n + 3) // 4 * 4
        self.slabs = torch.empty(off, dtype=torch.float32, device=device)
        self.da_part = torch.zeros(da_off + 8, dtype=torch.float32, device=device)
        rows = [[self.slabs.data_ptr() + 4 * c.slab_off, c.slab_stride, c.n_slabs, self.grads.data_ptr() + 4 * c.w_off,
                 c.slab_stride] for c in convs]
        self.reduce_table = torch.tensor(rows, dtype=torch.int64).to(device)
        drows = [[self.da_part.data_ptr() + 4 * c.da_off, c.da_n, self.grads.data_ptr() + 4 * self.p_off[c.prelu]]
                 for c in convs if c.prelu]
        self.da_table = torch.tensor(drows, dtype=torch.int64).to(device) if drows else None
        self.n_da = len(drows)
        self.n_convs = len(convs)
        # gradient buckets: (tag, first conv, one past the last conv, first float, one past the last float), convs of a
        # group are consecutive on the tape and so are their parameters
        self.groups: List[Tuple[str, int, int, int, int]] = []
        for i, c in enumerate(convs):
            c.index = i
            if self.groups and self.groups[-1][0] == c.group:
                tag, lo, _, start, _ = self.groups[-1]
                self.groups[-1] = (tag, lo, i + 1, start, c.b_off + c.out.cp)
            else:
                self.groups.append((c.group, i, i + 1, c.w_off, c.b_off + c.out.cp))
        self._group_of_first = {lo: g for g in self.groups for lo in (g[1],)}

    # ------------------------------------------------------------------ graph construction
    def _tensor(self, level: int, C: int, coord: bool = False) -> _PT:
        t = _PT(self.geo[level], C, self.device, coord)
        if coord:    # AddCoords channels are constants of the buffer (modules.py:65-96)
            g = t.geo
            call("vlg_fill_coords", t.ptr, g.b, g.H, g.W, t.cp, t.coord_c0, self._stream())
        t.level = level
        self.tensors.append(t)
        return t

    def _conv(self, key, x: _PT, cout, stride=1, prelu=None, resid=None, act_ch=None, out_coord=False) -> _PT:
        out = self._tensor(x.level + (1 if stride == 2 else 0), cout, coord=out_coord)
        cin = x.C + (2 if x.coord_c0 >= 0 else 0)
        op = _Conv(key, x, out, cin, cout, stride, prelu, resid, act_ch if act_ch is not None else x.cp)
        op.group = self._group
        if prelu is not None:
            self.prelu_keys.append(prelu)
        self.tape.append(op)
        return out

    def _block(self, name, kind, x: _PT, cout, resid=None) -> _PT:
        if kind == "up":     # [Upsample, PReLU, Conv, PReLU, Conv]  (modules.py:49-55)
            u = self._tensor(x.level - 1, x.C)
            self.tape.append(("up", x, u))
            t = self._conv(name + ".up.2", u, cout, prelu=name + ".up.1.weight")
            return self._conv(name + ".up.4", t, cout, prelu=name + ".up.3.weight", resid=resid)
        t = self._conv(name + ".conv.1", x, cout, stride=2 if kind == "down" else 1, prelu=name + ".conv.0.weight")
        return self._conv(name + ".conv.3", t, cout, prelu=name + ".conv.2.weight", resid=resid)

    def _alloc_params(self, params_from: Optional["GridNetHIP"] = None) -> None:
        off = 0
        self.p_off: Dict[str, int] = {}
        for op in self.tape:
            if isinstance(op, _Conv):
                op.w_off = off
                off += op.out.cp * 9 * op.x.cp
                op.b_off = off
                off += op.out.cp
        for k in self.prelu_keys:
            self.p_off[k] = off
            off += 4
        self.n_params_padded = off
        if params_from is not None:
            if (params_from.coord, params_from.filters, params_from.n_channels, params_from.n_params_padded) != \
                    (self.coord, self.filters, self.n_channels, off):
                raise ValueError("params_from must be the same architecture")
            self.params, self.grads = params_from.params, None
            return
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        # TAIL_EXTRA floats after the last parameter carry the step's loss scalars, so they travel inside the last
        # data-parallel gradient bucket instead of a collective of their own (reference src/trainer.py:256 spent one)
        self.grads_ext = torch.zeros(off + self.TAIL_EXTRA, dtype=torch.float32, device=self.device)
        self.grads = self.grads_ext[:off]

    # --------------------------------------------------------------------------- parameters
    def reference_shapes(self) -> "OrderedDict[str, Tuple[int, ...]]":
        """Reference state_dict keys -> shapes (as a set; the reference's own key order is the order of its
        constructors, which nothing here depends on)."""
        s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        for op in self.tape:
            if isinstance(op, _Conv):
                s[op.key + ".weight"] = (op.cout, op.cin, 3, 3)
                s[op.key + ".bias"] = (op.cout,)
        for k in self.prelu_keys:
            s[k] = (1,)
        return s

    def pack(self, sd: Dict[str, torch.Tensor], flat: torch.Tensor) -> None:
        """Reference-format tensors ([cout,cin,3,3] weights, [cout] biases, [1] slopes) -> a flat buffer in the kernel
        layout (padded lanes zero).  Used for the parameters and for per-parameter optimiser state alike."""
        want = self.reference_shapes()
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError("state_dict lacks %s" % missing[:4])
        if flat.numel() != self.n_params_padded:
            raise ValueError("flat buffer has %d floats, layout needs %d" % (flat.numel(), self.n_params_padded))
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError("%s has shape %s, expected %s" % (k, tuple(sd[k].shape), tuple(shp)))
        flat.zero_()
        for op in self.tape:
            if isinstance(op, _Conv):
                w = sd[op.key + ".weight"].to(torch.float32)
                wp = torch.zeros(op.out.cp, 9, op.x.cp)
                wp[:op.cout, :, :op.cin] = w.permute(0, 2, 3, 1).reshape(op.cout, 9, op.cin)
                flat[op.w_off:op.w_off + wp.numel()].copy_(wp.flatten())
                flat[op.b_off:op.b_off + op.cout].copy_(sd[op.key + ".bias"].to(torch.float32))
        for k in self.prelu_keys:
            flat[self.p_off[k]:self.p_off[k] + 1].copy_(sd[k].to(torch.float32).flatten())

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Reference-format state_dict (reference src/trainer.py:89-90 'gridnet') -> the parameter buffer."""
        self.pack(sd, self.params)

    def unpack(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Inverse of pack(): flat kernel-layout buffer -> {reference key: tensor} on the CPU."""
        return self._export(flat)

    def _export(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        f = flat.detach().cpu()
        for op in self.tape:
            if isinstance(op, _Conv):
                wp = f[op.w_off:op.w_off + op.out.cp * 9 * op.x.cp].view(op.out.cp, 3, 3, op.x.cp)
                out[op.key + ".weight"] = wp[:op.cout, :, :, :op.cin].permute(0, 3, 1, 2).contiguous()
                out[op.key + ".bias"] = f[op.b_off:op.b_off + op.cout].clone()
        for k in self.prelu_keys:
            out[k] = f[self.p_off[k]:self.p_off[k] + 1].clone()
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self._export(self.params)

    def named_grads(self) -> Dict[str, torch.Tensor]:
        return self._export(self.grads)

    def prelu_inputs(self) -> Dict[str, torch.Tensor]:
        """{PReLU key: its input of the last forward, (b,C,H,W)} - diagnostics / tests: which side of the kink
        every pre-activation fell on (the activation is applied by the consuming conv on load, so the stored
        tensor IS the pre-activation)."""
        out, s = {}, self._stream()
        for op in self.tape:
            if isinstance(op, _Conv) and op.prelu:
                g, C = op.x.geo, op.x.C
                t = torch.empty(g.b, C, g.H, g.W, dtype=torch.float32, device=self.device)
                call("vlg_padded_to_nchw", op.x.ptr, ptr(t), g.b, C, g.H, g.W, op.x.cp, s)
                out[op.prelu] = t
        return out

    # ------------------------------------------------------------------------------ forward
    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def _pp(self, off: int) -> int:
        return self.params.data_ptr() + 4 * off

    def forward(self, x: torch.Tensor):
        g0 = self.geo[0]
        if tuple(x.shape) != (g0.b, self.n_channels, g0.H, g0.W) or not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("input must be a float32 HIP tensor of shape %s" % ((g0.b, self.n_channels, g0.H, g0.W),))
        s = self._stream()
        x = x.contiguous()
        call("vlg_nchw_to_padded", ptr(x), self.x.ptr, g0.b, self.n_channels, g0.H, g0.W, self.x.cp, -1, s)
        for op in self.tape:
            if isinstance(op, _Conv):
                gx, go = op.x.geo, op.out.geo
                rowtab = ptr(gx.down_rowtab) if op.stride == 2 else 0
                call("vlg_conv3x3_fwd", op.x.ptr, self._pp(op.w_off), self._pp(op.b_off), op.out.ptr,
                     op.resid.ptr if op.resid is not None else 0, ptr(go.mask),
                     self._pp(self.p_off[op.prelu]) if op.prelu else 0, rowtab, go.rows, op.x.cp, op.cout, op.out.cp,
                     gx.wp, op.act_ch, CEPI_RESID if op.resid is not None else 0, ptr(self.ws), self.ws_n, s)
            else:
                _, src, dst = op
                call("vlg_upsample2x_fwd", src.ptr, dst.ptr, src.geo.b, src.geo.H, src.geo.W, src.cp, s)
        seg = torch.empty(g0.b, self.seg_out, g0.H, g0.W, dtype=torch.float32, device=self.device)
        img = torch.empty(g0.b, self.img_out, g0.H, g0.W, dtype=torch.float32, device=self.device)
        call("vlg_padded_to_nchw", self.seg.ptr, ptr(seg), g0.b, self.seg_out, g0.H, g0.W, self.seg.cp, s)
        call("vlg_padded_to_nchw", self.img.ptr, ptr(img), g0.b, self.img_out, g0.H, g0.W, self.img.cp, s)
        return seg, img

    # ----------------------------------------------------------------------------- backward
    def _grad_of(self, t: _PT) -> _PT:
        if t.grad is None:
            t.grad = _PT(t.geo, t.C + (2 if t.coord_c0 >= 0 else 0), self.device)
            assert t.grad.cp == t.cp
        return t.grad

    def bucket_ranges(self) -> List[Tuple[str, int, int]]:
        """[(tag, start, end)] over grads_ext in the order backward completes them (vlg.dp.GradReducer): heads, grid
        columns 5..1, the input blocks, and a tail bucket = every PReLU slope + the TAIL_EXTRA loss floats."""
        out = [(tag, start, end) for tag, _, _, start, end in reversed(self.groups)]
        out.append(("tail", self.groups[-1][4], self.n_params_padded + self.TAIL_EXTRA))
        return out

    def backward(self, dseg: torch.Tensor, dimg: torch.Tensor, reducer=None) -> Optional[torch.Tensor]:
        """Parameter gradients of sum(seg*dseg) + sum(img*dimg) into self.grads (overwritten).  Returns the
        input gradient (NCHW) when built with need_input_grad.  With a `reducer` (vlg.dp.GradReducer over grads_ext,
        buckets = bucket_ranges()) each bucket's partial sums are reduced as soon as its last convolution is done and
        the bucket is handed over, so its all-reduce overlaps the rest of backward (DDP's overlap, reference
        src/trainer.py:113); without one, a single table-driven launch reduces everything at the end."""
        if self.forward_only:
            raise RuntimeError("this GridNetHIP was built forward-only (params_from)")
        lib = hip.load()
        s = self._stream()
        g0 = self.geo[0]
        for t in self.tensors:
            t.grad_written = False
        for t, d, C in ((self.seg, dseg, self.seg_out), (self.img, dimg, self.img_out)):
            gt = self._grad_of(t)
            d = d.contiguous()
            call("vlg_nchw_to_padded", ptr(d), gt.ptr, g0.b, C, g0.H, g0.W, gt.cp, -1, s)
            t.grad_written = True
        for op in reversed(self.tape):
            if isinstance(op, _Conv):
                gx, go = op.x.geo, op.out.geo
                dout = op.out.grad
                if dout is None or not op.out.grad_written:
                    raise RuntimeError("no gradient reached the output of %s" % op.key)
                slope = self._pp(self.p_off[op.prelu]) if op.prelu else 0
                # weight + bias gradient
                call("vlg_conv3x3_wgrad", dout.ptr, op.x.ptr, self.slabs.data_ptr() + 4 * op.slab_off, op.slab_stride,
                     self.slabs.numel() - op.slab_off,
                     ptr(gx.down_rowtab) if op.stride == 2 else 0, slope, go.rows, op.x.cp, op.out.cp, gx.wp, op.act_ch, s)
                # data gradient (skipped for the network input unless asked for)
                if op.x is not self.x or self.need_input_grad or op.prelu:
                    gin = self._grad_of(op.x)
                    epi = (CEPI_ACCUM if op.x.grad_written else 0) | (CEPI_DPRELU if op.prelu else 0)
                    taps = ptr(gx.down_taptabs) if op.stride == 2 else 0
                    call("vlg_conv3x3_dgrad", dout.ptr, self._pp(op.w_off), gin.ptr, op.x.ptr, ptr(gx.mask), slope,
                         self.da_part.data_ptr() + 4 * op.da_off if op.prelu else 0, taps, gx.rows if op.stride == 2 else 0, gx.rows, op.x.cp,
                         op.out.cp, gx.wp, op.act_ch, epi, 0, 0, op.da_n, s)
                    op.x.grad_written = True
                # the other branch of the residual sum receives the same gradient (gridnet.py:51-56)
                if op.resid is not None:
                    gr = self._grad_of(op.resid)
                    call("vlg_add_rows", gr.ptr, dout.ptr, dout.n, 1 if op.resid.grad_written else 0, s)
                    op.resid.grad_written = True
                if reducer is not None and op.index in self._group_of_first:       # bucket complete
                    tag, lo, hi, _, _ = self._group_of_first[op.index]
                    call("vlg_reduce_slabs_table", self.reduce_table.data_ptr() + 40 * lo, hi - lo, 64, s)
                    reducer.ready(tag)
            else:
                _, src, dst = op
                gs = self._grad_of(src)
                call("vlg_upsample2x_bwd", dst.grad.ptr, gs.ptr, src.geo.b, src.geo.H, src.geo.W, src.cp,
                     1 if src.grad_written else 0, s)
                src.grad_written = True
        # every weight / bias gradient (unless reduced bucket by bucket above) and every PReLU slope gradient
        if reducer is None:
            call("vlg_reduce_slabs_table", ptr(self.reduce_table), self.n_convs, 64, s)
        if self.da_table is not None:
            call("vlg_sum_partials_table", ptr(self.da_table), self.n_da, s)
        if reducer is not None:
            reducer.ready("tail")
        if self.need_input_grad:
            dx = torch.empty(g0.b, self.n_channels, g0.H, g0.W, dtype=torch.float32, device=self.device)
            call("vlg_padded_to_nchw", self.x.grad.ptr, ptr(dx), g0.b, self.n_channels, g0.H, g0.W, self.x.cp, s)
            return dx
        return None
