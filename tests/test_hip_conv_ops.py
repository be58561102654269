"""GPU: ONE convolution at a time through the C ABI - vlg_conv3x3_fwd / _dgrad / _wgrad (csrc/conv.hip) against
PReLU -> Conv2d(k=3, padding=1[, stride=2]) exactly as the reference composes them (reference
src/models/modules.py:12-17 lateral, :34-39 down-sampling; nn.PReLU() = one shared slope) evaluated by torch-CPU in
fp64 with autograd.

Every input element is kept >= 1e-3 away from the PReLU kink, so both implementations take the same branch at every
element and there is no rounding-band excuse: outputs, dx, dW, db AND the slope gradient must agree to 1e-4 of the
tensor's scale with REAL slopes (0.25, a negative one, 0 = the ReLU the frozen trunks use), stride 1 and 2, channel
counts 3 / 10 / 32 / 64 / 96 / 128, activation limited to the first act_ch channels (AddCoords' constant channels
stay linear and pass no gradient, modules.py:65-135), the residual-sum epilogue, weight gradients split over several
row ranges, ragged widths, and the split-K forward / data gradient of the frozen trunks' coarse levels."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _away_from_kink(shape, g, margin=1e-3):
    v = torch.randn(shape, generator=g)
    return torch.where(v >= 0, v + margin, v - margin)


def _rel(got, want, what, tol=TOL):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(float(want.abs().max()), 1e-12)
    err = float((got - want).abs().max())
    assert err <= tol * scale, "%s: max |err| %.3e vs %.1e x scale %.3e" % (what, err, tol, scale)


class _Harness:
    """Padded-NHWC plumbing around one convolution (the layout GridNetHIP uses, vlg/gridnet.py)."""

    def __init__(self, dev, b, H, W, cin, cout, stride):
        from vlg.gridnet import _Geo, _PT
        self.dev, self.b, self.H, self.W, self.cin, self.cout, self.stride = dev, b, H, W, cin, cout, stride
        self.gi = _Geo(b, H, W, dev)
        self.go = self.gi if stride == 1 else _Geo(b, H // 2, W // 2, dev)
        if stride == 2:
            self.gi.link_coarser(self.go, dev)
        self.PT = _PT
        self.x, self.dx = _PT(self.gi, cin, dev), _PT(self.gi, cin, dev)
        self.y, self.dy, self.res = _PT(self.go, cout, dev), _PT(self.go, cout, dev), _PT(self.go, cout, dev)
        self.cin_p, self.cout_p = self.x.cp, self.y.cp
        self.S = torch.cuda.current_stream().cuda_stream

    def put(self, t, pt, C, geo):
        from vlg import hip
        t = t.to(self.dev).float().contiguous()
        hip.call("vlg_nchw_to_padded", t.data_ptr(), pt.ptr, geo.b, C, geo.H, geo.W, pt.cp, -1, self.S)

    def get(self, pt, C, geo):
        from vlg import hip
        out = torch.empty(geo.b, C, geo.H, geo.W, device=self.dev)
        hip.call("vlg_padded_to_nchw", pt.ptr, out.data_ptr(), geo.b, C, geo.H, geo.W, pt.cp, self.S)
        return out.cpu()

    def pack_weight(self, w):
        wp = torch.zeros(self.cout_p, 9, self.cin_p)
        wp[:self.cout, :, :self.cin] = w.float().permute(0, 2, 3, 1).reshape(self.cout, 9, self.cin)
        return wp.flatten().to(self.dev)

    def unpack_weight(self, flat):
        wp = flat.cpu()[:self.cout_p * 9 * self.cin_p].view(self.cout_p, 3, 3, self.cin_p)
        return wp[:self.cout, :, :, :self.cin].permute(0, 3, 1, 2).contiguous()


def _reference(x, w, bias, slope, act_ch, stride, resid, r):
    """fp64 torch: y = conv(prelu on channels < act_ch)(+resid); gradients of sum(y * r)."""
    x = x.double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    bias = bias.double().requires_grad_(True)
    a = None if slope is None else torch.tensor([slope], dtype=torch.float64, requires_grad=True)
    if a is None:
        xa = x
    else:
        xa = torch.cat([F.prelu(x[:, :act_ch], a), x[:, act_ch:]], dim=1)
    if a is not None:
        xa.retain_grad()
    y = F.conv2d(xa, w, bias, stride=stride, padding=1)
    if resid is not None:
        y = y + resid.double()
    (y * r.double()).sum().backward()
    dx = x.grad.clone()
    dx[:, act_ch:] = 0                    # constant (AddCoords) channels pass no gradient on
    if a is None:
        return y.detach(), dx, w.grad, bias.grad, None, 0.0
    # the slope gradient is ONE number, sum over the negative side of x * dL/d prelu(x): its natural scale is the
    # sum of the magnitudes of those terms
    neg = (x.detach()[:, :act_ch] < 0)
    da_scale = float((x.detach()[:, :act_ch] * xa.grad[:, :act_ch]).abs()[neg].sum())
    return y.detach(), dx, w.grad, bias.grad, a.grad, da_scale


CASES = [
    # b, H,  W, cin, cout, stride, act_ch, resid, slope
    (2, 12, 20, 32, 32, 1, None, False, 0.25),
    (2, 12, 20, 10, 32, 1, None, False, 0.25),         # network input width
    (1, 16, 16, 3, 64, 1, None, True, -0.3),           # negative slope, residual sum
    (2, 12, 20, 96, 96, 1, None, True, 0.17),
    (1, 8, 12, 64, 96, 1, None, False, 0.25),
    (2, 16, 24, 32, 64, 2, None, False, 0.25),         # down-sampling blocks (modules.py:34-39)
    (1, 16, 16, 64, 96, 2, None, True, -0.2),
    (3, 12, 12, 12, 32, 1, 10, False, 0.3),            # CoordConv: 10 data + 2 coordinate channels stay linear
    (1, 12, 12, 34, 32, 1, 32, True, 0.3),             # CoordLateralBlock's second conv (32 + 2)
    (2, 40, 44, 64, 64, 1, None, False, 0.25),         # 3864 rows: weight gradient in several row ranges
    (2, 40, 44, 32, 20, 1, None, False, 0.1),          # head widths (20 classes)
    (1, 24, 28, 20, 3, 1, None, False, 0.4),           # head widths (3 colours), ragged width
    (1, 16, 20, 32, 32, 1, None, False, 0.0),          # ReLU (frozen trunks)
    (1, 16, 20, 32, 64, 1, None, False, None),         # no activation (shortcut conv, modules.py:22-25)
    (4, 64, 64, 32, 32, 1, None, True, 0.25),          # full tiles + many blocks
    (4, 64, 64, 96, 64, 1, None, False, 0.3),          # up-sampling block widths 96 -> 64
]


@pytest.mark.parametrize("b,H,W,cin,cout,stride,act_ch,resid,slope", CASES)
def test_conv3x3_fwd_dgrad_wgrad(dev, b, H, W, cin, cout, stride, act_ch, resid, slope):
    from vlg import hip
    from vlg.hip import CEPI_DPRELU, CEPI_RESID
    lib = hip.load()
    g = torch.Generator().manual_seed(1000 * cin + 10 * cout + stride)
    h = _Harness(dev, b, H, W, cin, cout, stride)
    act = cin if act_ch is None else act_ch
    x = _away_from_kink((b, cin, H, W), g)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    bias = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    Ho, Wo = H // stride, W // stride
    r = torch.randn(b, cout, Ho, Wo, generator=g)
    rs = torch.randn(b, cout, Ho, Wo, generator=g) if resid else None
    y_w, dx_w, dw_w, db_w, da_w, da_scale = _reference(x, w, bias, slope, act, stride, rs, r)

    S = h.S
    h.put(x, h.x, cin, h.gi)
    h.put(r, h.dy, cout, h.go)
    if resid:
        h.put(rs, h.res, cout, h.go)
    wdev = h.pack_weight(w)
    bdev = torch.zeros(h.cout_p, device=dev)
    bdev[:cout] = bias.to(dev)
    sl = None if slope is None else torch.tensor([slope, 0, 0, 0], dtype=torch.float32, device=dev)
    slp = 0 if sl is None else sl.data_ptr()
    act_arg = h.cin_p if act_ch is None else act_ch          # GridNetHIP passes x.cp when every channel is activated
    rowtab = h.gi.down_rowtab.data_ptr() if stride == 2 else 0
    # ---- forward
    hip.call("vlg_conv3x3_fwd", h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, h.res.ptr if resid else 0,
             h.go.mask.data_ptr(), slp, rowtab, h.go.rows, h.cin_p, cout, h.cout_p, h.gi.wp, act_arg,
             CEPI_RESID if resid else 0, 0, 0, S)
    _rel(h.get(h.y, cout, h.go), y_w, "forward")
    # the halo of the output stays exactly zero (the next conv's zero padding)
    yp = h.y.buf[h.go.guard * h.cout_p:(h.go.guard + h.go.rows) * h.cout_p].view(b, Ho + 2, Wo + 2, h.cout_p)
    assert float(yp[:, 0].abs().max()) == 0 and float(yp[:, :, 0].abs().max()) == 0
    assert float(yp[:, -1].abs().max()) == 0 and float(yp[:, :, -1].abs().max()) == 0
    # ---- data gradient (+ PReLU' and the slope-gradient partials)
    n_da = lib.vlg_conv3x3_dgrad_slabs(h.gi.rows, h.cin_p)
    da_part = torch.zeros(n_da + 8, device=dev)
    taps = h.gi.down_taptabs.data_ptr() if stride == 2 else 0
    hip.call("vlg_conv3x3_dgrad", h.dy.ptr, wdev.data_ptr(), h.dx.ptr, h.x.ptr, h.gi.mask.data_ptr(), slp,
             da_part.data_ptr() if sl is not None else 0, taps, h.gi.rows if stride == 2 else 0, h.gi.rows, h.cin_p,
             h.cout_p, h.gi.wp, act_arg, CEPI_DPRELU if sl is not None else 0, 0, 0, n_da, S)
    _rel(h.get(h.dx, cin, h.gi), dx_w, "dx")
    if sl is not None:
        da = torch.zeros(4, device=dev)
        hip.call("vlg_sum_partials", da_part.data_ptr(), n_da, da.data_ptr(), 0, S)
        assert abs(float(da[0]) - float(da_w)) <= TOL * max(abs(float(da_w)), 1e-2 * da_scale), (float(da[0]), float(da_w), da_scale)
    # ---- weight + bias gradient
    n_slabs = lib.vlg_conv3x3_wgrad_slabs(h.go.rows, h.cin_p, h.cout_p)
    stride_f = h.cout_p * 9 * h.cin_p + h.cout_p
    slabs = torch.empty(n_slabs * stride_f, device=dev)
    hip.call("vlg_conv3x3_wgrad", h.dy.ptr, h.x.ptr, slabs.data_ptr(), stride_f, slabs.numel(), rowtab, slp, h.go.rows,
             h.cin_p, h.cout_p, h.gi.wp, act_arg, S)
    gw = torch.empty(stride_f, device=dev)
    hip.call("vlg_reduce_slabs", slabs.data_ptr(), stride_f, n_slabs, gw.data_ptr(), stride_f, S)
    torch.cuda.synchronize()
    if (b, H, W) == (2, 40, 44):
        assert n_slabs > 1, "this case is meant to split the weight gradient over row ranges"
    _rel(h.unpack_weight(gw), dw_w, "dW")
    _rel(gw.cpu()[h.cout_p * 9 * h.cin_p:][:cout], db_w, "db")
    # padded weight lanes receive exactly zero (they must stay zero under Adam)
    full = gw.cpu()[:h.cout_p * 9 * h.cin_p].view(h.cout_p, 9, h.cin_p)
    assert float(full[cout:].abs().max() if cout < h.cout_p else 0.0) == 0.0
    if act_ch is None:
        assert float(full[:, :, cin:].abs().max() if cin < h.cin_p else 0.0) == 0.0


@pytest.mark.parametrize("b,H,W,cin,cout", [(1, 16, 16, 128, 128), (2, 8, 8, 256, 256), (1, 16, 16, 128, 256)])
def test_split_k_trunk_convs(dev, b, H, W, cin, cout):
    """Coarse levels of the frozen VGG19 / HED trunks (loss.py:29-49, hned.py:9-58): ReLU -> conv with the contraction
    cut into K ranges through a workspace (forward) and the ReLU' data gradient the VGG term back-propagates."""
    from vlg import hip
    from vlg.hip import CEPI_DPRELU
    lib = hip.load()
    g = torch.Generator().manual_seed(cin + cout)
    h = _Harness(dev, b, H, W, cin, cout, 1)
    x = _away_from_kink((b, cin, H, W), g)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    bias = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    r = torch.randn(b, cout, H, W, generator=g)
    y_w, dx_w, _, _, _, _ = _reference(x, w, bias, 0.0, cin, 1, None, r)
    S = h.S
    h.put(x, h.x, cin, h.gi)
    h.put(r, h.dy, cout, h.go)
    wdev, bdev = h.pack_weight(w), bias.to(dev).contiguous()
    zero = torch.zeros(4, device=dev)
    splits = lib.vlg_conv3x3_fwd_splits(h.go.rows, h.cin_p, cout, h.cout_p)
    assert splits > 1, "shape is meant to take the split-K path"
    ws = torch.empty(splits * h.go.rows * h.cout_p, device=dev)
    hip.call("vlg_conv3x3_fwd", h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, 0, h.go.mask.data_ptr(),
             zero.data_ptr(), 0, h.go.rows, h.cin_p, cout, h.cout_p, h.gi.wp, h.cin_p, 0, ws.data_ptr(), ws.numel(), S)
    _rel(h.get(h.y, cout, h.go), y_w, "split-K forward")
    dsplits = lib.vlg_conv3x3_dgrad_splits(h.gi.rows, h.cin_p, h.cout_p)
    assert dsplits > 1
    ws2 = torch.empty(dsplits * h.gi.rows * h.cin_p, device=dev)
    hip.call("vlg_conv3x3_dgrad", h.dy.ptr, wdev.data_ptr(), h.dx.ptr, h.x.ptr, h.gi.mask.data_ptr(), zero.data_ptr(),
             0, 0, 0, h.gi.rows, h.cin_p, h.cout_p, h.gi.wp, h.cin_p, CEPI_DPRELU, ws2.data_ptr(), ws2.numel(), 0, S)
    _rel(h.get(h.dx, cin, h.gi), dx_w, "split-K dx")
    # a short workspace is refused on the host, before any launch
    rc = lib.vlg_conv3x3_fwd(h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, 0, h.go.mask.data_ptr(), zero.data_ptr(),
                             0, h.go.rows, h.cin_p, cout, h.cout_p, h.gi.wp, h.cin_p, 0, ws.data_ptr(), ws.numel() - 1, S)
    assert rc == 1001


@pytest.mark.parametrize("b,H,W,cin,cout,resid", [(2, 128, 128, 128, 128, False), (2, 128, 128, 64, 128, True)])
def test_tail_split_convs(dev, b, H, W, cin, cout, resid):
    """265 / 529 row tiles on 256 CUs: with a workspace the tiles beyond the last full round are cut along K inside the same
    launch and summed by the finish kernel (csrc/conv.hip, conv_tail_plan) - forward with bias / residual / halo mask and
    the ReLU' data gradient (accumulating), against the same fp64 reference as the one-block-per-tile path."""
    from vlg import hip
    from vlg.hip import CEPI_ACCUM, CEPI_DPRELU, CEPI_RESID
    lib = hip.load()
    g = torch.Generator().manual_seed(3 * cin + cout)
    h = _Harness(dev, b, H, W, cin, cout, 1)
    x = _away_from_kink((b, cin, H, W), g)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    bias = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    r = torch.randn(b, cout, H, W, generator=g)
    rs = torch.randn(b, cout, H, W, generator=g) if resid else None
    prior = torch.randn(b, cin, H, W, generator=g)
    y_w, dx_w, _, _, _, _ = _reference(x, w, bias, 0.0, cin, 1, rs, r)
    S = h.S
    h.put(x, h.x, cin, h.gi)
    h.put(r, h.dy, cout, h.go)
    h.put(prior, h.dx, cin, h.gi)
    if resid:
        h.put(rs, h.res, cout, h.go)
    wdev, bdev = h.pack_weight(w), bias.to(dev).contiguous()
    zero = torch.zeros(4, device=dev)
    need = lib.vlg_conv3x3_fwd_workspace(h.go.rows, h.cin_p, cout, h.cout_p)
    assert lib.vlg_conv3x3_fwd_splits(h.go.rows, h.cin_p, cout, h.cout_p) == 1 and 0 < need < h.go.rows * h.cout_p, need
    ws = torch.full((need,), float("nan"), device=dev)
    hip.call("vlg_conv3x3_fwd", h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, h.res.ptr if resid else 0, h.go.mask.data_ptr(),
             zero.data_ptr(), 0, h.go.rows, h.cin_p, cout, h.cout_p, h.gi.wp, h.cin_p, CEPI_RESID if resid else 0,
             ws.data_ptr(), ws.numel(), S)
    _rel(h.get(h.y, cout, h.go), y_w, "tail-split forward")
    assert not bool(torch.isnan(ws).all()), "the workspace was not used: the tail plan did not run"
    yp = h.y.buf[h.go.guard * h.cout_p:(h.go.guard + h.go.rows) * h.cout_p].view(b, H + 2, W + 2, h.cout_p)
    assert float(yp[:, -1].abs().max()) == 0 and float(yp[:, :, -1].abs().max()) == 0      # halo rows of the tail stay zero
    # one float short: the launch falls back to one block per tile, same result
    h.y.buf.zero_()
    hip.call("vlg_conv3x3_fwd", h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, h.res.ptr if resid else 0, h.go.mask.data_ptr(),
             zero.data_ptr(), 0, h.go.rows, h.cin_p, cout, h.cout_p, h.gi.wp, h.cin_p, CEPI_RESID if resid else 0,
             ws.data_ptr(), ws.numel() - 1, S)
    _rel(h.get(h.y, cout, h.go), y_w, "forward, workspace too small for the tail plan")
    dneed = lib.vlg_conv3x3_dgrad_workspace(h.gi.rows, h.cin_p, h.cout_p)
    assert dneed > 0
    ws2 = torch.full((dneed,), float("nan"), device=dev)
    hip.call("vlg_conv3x3_dgrad", h.dy.ptr, wdev.data_ptr(), h.dx.ptr, h.x.ptr, h.gi.mask.data_ptr(), zero.data_ptr(),
             0, 0, 0, h.gi.rows, h.cin_p, h.cout_p, h.gi.wp, h.cin_p, CEPI_DPRELU | CEPI_ACCUM, ws2.data_ptr(), ws2.numel(), 0, S)
    _rel(h.get(h.dx, cin, h.gi), dx_w + prior.double(), "tail-split dx (accumulated)")
    assert not bool(torch.isnan(ws2).all())


@pytest.mark.parametrize("b,H,W,cin,cout", [(1, 16, 16, 3, 64), (2, 40, 44, 3, 64), (1, 24, 28, 4, 20)])
def test_image_layer_conv_cin4(dev, b, H, W, cin, cout):
    """First layers of the frozen VGG19 / HED trunks (loss.py:29-49, hned.py:9-58: Conv2d(3, 64, 3, padding=1) on the
    image, no activation in front): VLG_CEPI_CIN4 contracts over (tap, 4 channels) instead of 9 x 32 padded channels -
    the same convolution, against the same fp64 reference; halo stays zero; a partial row tile and a narrow head."""
    from vlg import hip
    from vlg.hip import CEPI_CIN4
    lib = hip.load()
    g = torch.Generator().manual_seed(31 * cin + cout + H)
    h = _Harness(dev, b, H, W, cin, cout, 1)
    x = torch.randn(b, cin, H, W, generator=g)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    bias = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    y_w = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    h.put(x, h.x, cin, h.gi)
    wdev = h.pack_weight(w)
    bdev = torch.zeros(h.cout_p, device=dev)
    bdev[:cout] = bias.to(dev)
    h.y.buf.fill_(7.0)                                   # stale contents everywhere, guard band included
    hip.call("vlg_conv3x3_fwd", h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, 0, h.go.mask.data_ptr(), 0, 0, h.go.rows,
             h.cin_p, cout, h.cout_p, h.gi.wp, h.cin_p, CEPI_CIN4, 0, 0, h.S)
    _rel(h.get(h.y, cout, h.go), y_w, "image-layer forward")
    yp = h.y.buf[h.go.guard * h.cout_p:(h.go.guard + h.go.rows) * h.cout_p].view(b, H + 2, W + 2, h.cout_p)
    assert float(yp[:, 0, :, :cout].abs().max()) == 0 and float(yp[:, :, 0, :cout].abs().max()) == 0
    assert float(yp[:, -1, :, :cout].abs().max()) == 0 and float(yp[:, :, -1, :cout].abs().max()) == 0
    # combinations the image-layer kernel does not implement are refused on the host
    zero = torch.zeros(4, device=dev)
    rc = lib.vlg_conv3x3_fwd(h.x.ptr, wdev.data_ptr(), bdev.data_ptr(), h.y.ptr, 0, h.go.mask.data_ptr(), zero.data_ptr(), 0,
                             h.go.rows, h.cin_p, cout, h.cout_p, h.gi.wp, h.cin_p, CEPI_CIN4, 0, 0, h.S)
    assert rc == 1001


def test_dgrad_accumulates_into_shared_input(dev):
    """A tensor consumed by two blocks (gridnet.py:51-56) collects both data gradients: VLG_CEPI_ACCUM adds."""
    from vlg import hip
    from vlg.hip import CEPI_ACCUM, CEPI_DPRELU
    lib = hip.load()
    b, H, W, cin, cout = 2, 12, 16, 32, 64
    g = torch.Generator().manual_seed(7)
    h = _Harness(dev, b, H, W, cin, cout, 1)
    x = _away_from_kink((b, cin, H, W), g)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    r = torch.randn(b, cout, H, W, generator=g)
    prior = torch.randn(b, cin, H, W, generator=g)
    _, dx_w, _, _, _, _ = _reference(x, w, torch.zeros(cout), 0.25, cin, 1, None, r)
    h.put(x, h.x, cin, h.gi)
    h.put(r, h.dy, cout, h.go)
    h.put(prior, h.dx, cin, h.gi)
    sl = torch.tensor([0.25, 0, 0, 0], device=dev)
    n_da = lib.vlg_conv3x3_dgrad_slabs(h.gi.rows, h.cin_p)
    da_part = torch.zeros(n_da + 8, device=dev)
    hip.call("vlg_conv3x3_dgrad", h.dy.ptr, h.pack_weight(w).data_ptr(), h.dx.ptr, h.x.ptr, h.gi.mask.data_ptr(),
             sl.data_ptr(), da_part.data_ptr(), 0, 0, h.gi.rows, h.cin_p, h.cout_p, h.gi.wp, h.cin_p,
             CEPI_DPRELU | CEPI_ACCUM, 0, 0, n_da, h.S)
    _rel(h.get(h.dx, cin, h.gi), dx_w + prior.double(), "accumulated dx")
