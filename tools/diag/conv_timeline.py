#!/usr/bin/env python
"""Diagnostic: per-CU timeline of one 3x3 convolution launch (needs the -DVLG_TIMELINE build of conv.hip):

    cd video-layout-generation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DVLG_TIMELINE -c conv.hip -o /tmp/conv_tl.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libvlg_tl.so /tmp/conv_tl.o $(ls *.o | grep -v '^conv.o')
    VLG_HIP_LIB=$PWD/tools/ab/libvlg_tl.so python tools/diag/conv_timeline.py [hw cin cout [batch [fwd|dgrad|wgrad]]]

Every block stamps s_memrealtime (100 MHz) at entry, main-loop start, main-loop end and exit (stores drained) with
HW_ID / XCC_ID: block lifetime, prologue / loop / epilogue split, blocks per CU over time, idle gaps per CU."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
from vlg.hip import call
from vlg.gridnet import _Geo, _PT

hw, cin, cout = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 32, 32)
b = int(sys.argv[4]) if len(sys.argv) > 4 else 4
mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"          # fwd | dgrad | wgrad
dev = torch.device("cuda:0")
lib = hip.require_diag()
raw = ctypes.CDLL(hip.LIB_PATH)
raw.vlg_debug_set_conv_probe.argtypes = [ctypes.c_void_p]
ptr = lambda t: t.data_ptr()
S = torch.cuda.current_stream().cuda_stream
geo = _Geo(b, hw, hw, dev)
x, y = _PT(geo, cin, dev, False), _PT(geo, cout, dev, False)
x.buf.normal_()
w = torch.randn(y.cp * 9 * x.cp, device=dev) * 0.05
bias = torch.zeros(y.cp, device=dev)
zero = torch.zeros(4, device=dev)
y.buf.normal_()
dx = _PT(geo, cin, dev, False)
da = torch.zeros(lib.vlg_conv3x3_dgrad_slabs(geo.rows, x.cp) + 8, device=dev)
n_slab = lib.vlg_conv3x3_wgrad_slabs(geo.rows, x.cp, y.cp)
slab_stride = y.cp * 9 * x.cp + y.cp
slabs = torch.empty(n_slab * slab_stride, device=dev)
run = {"fwd": lambda: call("vlg_conv3x3_fwd", x.ptr, ptr(w), ptr(bias), y.ptr, 0, ptr(geo.mask), ptr(zero), 0, geo.rows, x.cp, cout, y.cp,
                           geo.wp, x.cp, 0, 0, 0, S),
       "dgrad": lambda: call("vlg_conv3x3_dgrad", y.ptr, ptr(w), dx.ptr, x.ptr, ptr(geo.mask), ptr(zero), ptr(da), 0, 0, geo.rows, x.cp, y.cp,
                             geo.wp, x.cp, 8, 0, 0, da.numel(), S),
       "wgrad": lambda: call("vlg_conv3x3_wgrad", y.ptr, x.ptr, ptr(slabs), slab_stride, slabs.numel(), 0, ptr(zero), geo.rows, x.cp, y.cp,
                             geo.wp, x.cp, S)}[mode]
nblk = 1 << 16
probe = torch.zeros(8 * nblk, dtype=torch.int64, device=dev)
for _ in range(3000):
    run()
raw.vlg_debug_set_conv_probe(probe.data_ptr())
run()
torch.cuda.synchronize()
raw.vlg_debug_set_conv_probe(None)
p = probe.cpu().view(-1, 8)
p = p[p[:, 3] != 0]
t = (p[:, :4] - p[:, 0].min()).double() / 100.0          # us since the first block's entry
hwid, xcc = p[:, 4], p[:, 5] & 0xf
cu = ((xcc << 8) | ((hwid >> 8) & 0xff)).tolist()
total = float(t[:, 3].max())
life, loop, pro, epi = t[:, 3] - t[:, 0], t[:, 2] - t[:, 1], t[:, 1] - t[:, 0], t[:, 3] - t[:, 2]
print("%d blocks on %d CUs, launch %.1f us; per block: life %.1f (%.1f-%.1f)  prologue %.2f  loop %.1f (%.1f-%.1f)  epilogue %.2f (%.1f-%.1f)" % (
    len(p), len(set(cu)), total, life.mean(), life.min(), life.max(), pro.mean(), loop.mean(), loop.min(), loop.max(), epi.mean(), epi.min(), epi.max()))
pk = p[:, 7]
lc, vm, br = (pk >> 40).double(), ((pk >> 20) & 0xfffff).double(), (pk & 0xfffff).double()
print("wave 0 of a block, shader clocks: main loop %.0f, waiting for tile loads %.0f (%.0f%%), at the barrier %.0f (%.0f%%)" % (
    lc.mean(), vm.mean(), 100 * vm.sum() / lc.sum(), br.mean(), 100 * br.sum() / lc.sum()))
# blocks resident over time (whole chip) and per-CU residency
import collections
ev = sorted([(float(a), 1) for a in t[:, 0]] + [(float(e), -1) for e in t[:, 3]])
res, last, area = 0, 0.0, collections.Counter()
for tt, dlt in ev:
    area[res] += tt - last
    last, res = tt, res + dlt
print("chip-wide resident blocks (time share): " + "  ".join("%d-%d: %.0f%%" % (lo, lo + 255, 100 * sum(v for k, v in area.items() if lo <= k < lo + 256) / total) for lo in range(0, 1537, 256)))
bycu = collections.defaultdict(list)
for i, c in enumerate(cu):
    bycu[c].append(i)
cnt = sorted(len(v) for v in bycu.values())
print("blocks per CU: min %d median %d max %d" % (cnt[0], cnt[len(cnt) // 2], cnt[-1]))
# per-CU: time with k blocks inside their main loop
share = collections.Counter()
for c, idx in bycu.items():
    ev = sorted([(float(t[i, 1]), 1) for i in idx] + [(float(t[i, 2]), -1) for i in idx])
    res, last = 0, 0.0
    for tt, dlt in ev:
        share[res] += tt - last
        last, res = tt, res + dlt
    share[0] += total - last
tot = sum(share.values())
print("per CU, blocks inside the main loop (time share): " + "  ".join("%d: %.0f%%" % (k, 100 * v / tot) for k, v in sorted(share.items())))
first = sorted(bycu.items(), key=lambda kv: len(kv[1]))[-1]
print("busiest CU %x:" % first[0])
for i in sorted(first[1], key=lambda i: float(t[i, 0])):
    print("   tile %5d  entry %6.1f  loop %6.1f .. %6.1f  exit %6.1f" % (int(p[i, 6]), *[float(v) for v in t[i]]))
