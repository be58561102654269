#!/usr/bin/env python
"""Development tool: per-shape throughput of vlg_conv3x3_{fwd,dgrad,wgrad} on the shapes the reference's step uses
(GridNet rows 32/64/96 channels, VGG19[:27] / HED VGG16 trunks).  Algorithmic FLOP = 2 * pixels * cin * cout * 9.

    python tools/conv_bench.py [batch] [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
from vlg.hip import call
from vlg.gridnet import _Geo, _PT

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ONLY = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None      # indices into SHAPES
NOACT = os.environ.get("CONV_BENCH_NOACT") == "1"                                   # no activation on load (slope pointer NULL)
dev = torch.device("cuda:0")
lib = hip.load()
ptr = lambda t: t.data_ptr()
stream = torch.cuda.current_stream().cuda_stream

SHAPES = [(S, 32, 32), (S // 2, 64, 64), (S // 4, 96, 96),                       # GridNet rows
          (S, 3, 64), (S, 64, 64), (S // 2, 64, 128), (S // 2, 128, 128), (S // 4, 128, 256), (S // 4, 256, 256),
          (S // 8, 256, 512), (S // 8, 512, 512), (S // 16, 512, 512)]          # VGG / HED trunks


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("b=%d  %-18s %10s %10s %10s   (TFLOP/s algorithmic; fp32 MFMA peak 157.3)" % (b, "HxW cin->cout", "fwd", "dgrad", "wgrad"))
for si, (hw, cin, cout) in enumerate(SHAPES):
    if ONLY is not None and si not in ONLY:
        continue
    geo = _Geo(b, hw, hw, dev)
    x, y = _PT(geo, cin, dev, False), _PT(geo, cout, dev, False)
    x.buf.normal_(); y.buf.normal_()
    dx = _PT(geo, cin, dev, False)
    w = torch.randn(y.cp * 9 * x.cp, device=dev) * 0.05
    bias = torch.zeros(y.cp, device=dev)
    zero = torch.zeros(1, device=dev)
    n_slab = lib.vlg_conv3x3_wgrad_slabs(geo.rows, x.cp, y.cp)
    slab_stride = y.cp * 9 * x.cp + y.cp
    slabs = torch.empty(n_slab * slab_stride, device=dev)
    da = torch.zeros(lib.vlg_conv3x3_dgrad_slabs(geo.rows, x.cp) + 8, device=dev)
    flop = 2.0 * b * hw * hw * cin * cout * 9
    nsp = lib.vlg_conv3x3_fwd_splits(geo.rows, x.cp, cout, y.cp)
    wsn = lib.vlg_conv3x3_fwd_workspace(geo.rows, x.cp, cout, y.cp)       # all tiles split (coarse levels) or the tail plan
    ws = torch.empty(wsn, device=dev) if wsn else None
    f = timeit(lambda: call("vlg_conv3x3_fwd", x.ptr, ptr(w), ptr(bias), y.ptr, 0, ptr(geo.mask), 0 if NOACT else ptr(zero), 0, geo.rows,
                            x.cp, cout, y.cp, geo.wp, x.cp, 0, hip.ptr(ws), ws.numel() if ws is not None else 0, stream))
    dsp = lib.vlg_conv3x3_dgrad_splits(geo.rows, x.cp, y.cp)
    dwsn = lib.vlg_conv3x3_dgrad_workspace(geo.rows, x.cp, y.cp)
    dws = torch.empty(dwsn, device=dev) if dwsn else None
    # da = NULL when a split path exists (frozen trunks ask for no slope gradient), else the GridNet form
    d = timeit(lambda: call("vlg_conv3x3_dgrad", y.ptr, ptr(w), dx.ptr, x.ptr, ptr(geo.mask), ptr(zero), 0 if dwsn else ptr(da),
                            0, 0, geo.rows, x.cp, y.cp, geo.wp, x.cp, 8, hip.ptr(dws), dws.numel() if dws is not None else 0, da.numel(), stream))
    g = timeit(lambda: call("vlg_conv3x3_wgrad", y.ptr, x.ptr, ptr(slabs), slab_stride, slabs.numel(), 0, 0 if NOACT else ptr(zero), geo.rows, x.cp, y.cp,
                            geo.wp, x.cp, stream))
    print("     %4dx%-4d %3d->%-3d %7.1f us %5.1f  %7.1f us %5.1f  %7.1f us %5.1f  (%d slabs%s)" % (
        hw, hw, cin, cout, f * 1e6, flop / f / 1e12, d * 1e6, flop / d / 1e12, g * 1e6, flop / g / 1e12, n_slab,
        (", fwd split-K %d" % nsp if nsp > 1 else (", fwd tail split" if wsn else "")) +
        (", dgrad split-K %d" % dsp if dsp > 1 else (", dgrad tail split" if dwsn else ""))))
