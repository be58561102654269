#!/usr/bin/env python
"""Diagnostic: per-CU timeline of one fp32 GEMM launch (needs the -DVLG_TIMELINE build: tools/ab/libvlg_tl.so).

Every block stamps s_memrealtime (100 MHz) at entry, main-loop start, main-loop end and exit (after its stores have
drained) together with HW_ID / XCC_ID.  Grouped by CU this shows whether the co-resident blocks of a CU run their
main loops and epilogues in lockstep, and how long no block of a CU has matrix work.
    cd video-layout-generation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Xclang -target-feature -Xclang -load-store-opt \
        -DVLG_TIMELINE -c gemm.hip -o /tmp/gemm_tl.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libvlg_tl.so \
        /tmp/gemm_tl.o $(ls *.o | grep -v '^gemm.o')
    VLG_HIP_LIB=$PWD/tools/ab/libvlg_tl.so python tools/diag/gemm_timeline.py [ff1|dgelu|mul|dplain|qkv|proj|wgrad ...] [sweep]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
from vlg.hip import EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_NONE, EPI_RESID
lib = hip.require_diag()
dev = torch.device("cuda:0")
M, d = 32768, 256
ff = 4 * d
S = torch.cuda.current_stream().cuda_stream
r = lambda *s: torch.randn(*s, device=dev)
P = lambda t: t.data_ptr()
x_d, x_3d, x_ff, x_ff2 = r(M, d), r(M, 3 * d), r(M, ff), r(M, ff)
y_d, y_3d, y_ff = r(M, d), r(M, 3 * d), r(M, ff)
w_qkv, w_proj, w_ff1, w_ff2 = r(3 * d, d), r(d, d), r(ff, d), r(d, ff)
bias = r(ff)
slabs = torch.empty(160 * (ff * d + ff), device=dev)
CASES = {
    "ff1": (2048, lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_ff1), d, P(bias), P(y_ff), ff, 0, P(x_ff2), M, ff, d, EPI_BIAS | EPI_GELU, S)),
    "qkv": (1536, lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_qkv), d, P(bias), P(y_3d), 3 * d, 0, 0, M, 3 * d, d, EPI_BIAS, S)),
    "proj": (512, lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_proj), d, P(bias), P(y_d), d, P(x_d), 0, M, d, d, EPI_BIAS | EPI_RESID, S)),
    "ff2": (512, lambda: hip.call("vlg_linear_fwd", P(x_ff), ff, P(w_ff2), ff, P(bias), P(y_d), d, P(x_d), 0, M, d, ff, EPI_BIAS | EPI_RESID, S)),
    "dgelu": (2048, lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(w_ff2), ff, P(y_ff), ff, P(x_ff2), M, d, ff, EPI_DGELU, S)),
    "mul": (2048, lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(w_ff2), ff, P(y_ff), ff, P(x_ff2), M, d, ff, 2048, S)),
    "dplain": (2048, lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(w_ff2), ff, P(y_ff), ff, 0, M, d, ff, EPI_NONE, S)),
    "dqkv": (512, lambda: hip.call("vlg_linear_dgrad", P(x_3d), 3 * d, P(w_qkv), d, P(y_d), d, 0, M, 3 * d, d, EPI_NONE, S)),
    "wgrad": (512, lambda: hip.call("vlg_linear_wgrad", P(x_ff), ff, P(x_d), d, P(slabs), ff * d + ff, slabs.numel(), M, ff, d, 0, S)),
}
which = [a for a in sys.argv[1:] if a != "sweep"] or ([] if "sweep" in sys.argv else ["ff1", "qkv", "proj", "wgrad"])
for name in which:
    nblk, run = CASES[name]
    probe = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
    for _ in range(8000):                   # ~1 s of back-to-back launches: the clock has settled (no sync before the stamped launch)
        run()
    lib.vlg_debug_set_clock_probe(probe.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.vlg_debug_set_clock_probe(None)
    p = probe.cpu().view(-1, 8)[:nblk]
    t = (p[:, :4] - p[:, 0].min()).double() / 100.0          # us since the first block's entry
    hw, xcc = p[:, 4], p[:, 5] & 0xf
    cu = ((xcc << 8) | ((hw >> 8) & 0xff)).tolist()          # (xcc, se, sh, cu) key
    total = float(t[:, 3].max())
    life, loop, pro, epi = t[:, 3] - t[:, 0], t[:, 2] - t[:, 1], t[:, 1] - t[:, 0], t[:, 3] - t[:, 2]
    print("== %s: %d blocks, launch %.1f us; per block: life %.1f (%.1f-%.1f)  prologue %.2f  loop %.1f (%.1f-%.1f)  epilogue %.1f (%.1f-%.1f)" % (
        name, nblk, total, life.mean(), life.min(), life.max(), pro.mean(), loop.mean(), loop.min(), loop.max(), epi.mean(), epi.min(), epi.max()))
    bycu = {}
    for b, c in enumerate(cu):
        bycu.setdefault(c, []).append(b)
    print("   CUs seen: %d, blocks per CU min/max %d/%d" % (len(bycu), min(len(v) for v in bycu.values()), max(len(v) for v in bycu.values())))
    # per CU: time with no block inside its main loop (matrix pipes certainly idle), and time with >= 1 / >= 2 in the loop
    res = 0.05
    nstep = int(total / res) + 1
    idle_frac, one_frac = [], []
    for c, bl in bycu.items():
        cnt = torch.zeros(nstep)
        for b in bl:
            a, e = int(t[b, 1] / res), int(t[b, 2] / res)
            cnt[a:e] += 1
        idle_frac.append(float((cnt == 0).float().mean()))
        one_frac.append(float((cnt == 1).float().mean()))
    idle = torch.tensor(idle_frac)
    print("   share of the launch with NO block of the CU in its main loop: mean %.3f  min %.3f  max %.3f;  exactly one: mean %.3f" % (
        idle.mean(), idle.min(), idle.max(), torch.tensor(one_frac).mean()))
    # chip-wide: blocks inside the epilogue as a function of time (coarse histogram, 16 bins)
    bins = 16
    hist_e, hist_l = [0] * bins, [0] * bins
    for i in range(bins):
        tt = (i + 0.5) * total / bins
        hist_l[i] = int(((t[:, 1] <= tt) & (tt < t[:, 2])).sum())
        hist_e[i] = int(((t[:, 2] <= tt) & (tt < t[:, 3])).sum())
    print("   blocks in main loop over time:", hist_l)
    print("   blocks in epilogue  over time:", hist_e)
    # three sample CUs
    for c in list(bycu)[:3]:
        bl = sorted(bycu[c], key=lambda b: float(t[b, 0]))
        print("   CU %04x:" % c, "  ".join("[%5.1f %5.1f | %5.1f %5.1f]" % tuple(t[b].tolist()) for b in bl))


def occupancy_sweep():
    """main-loop rate with 1 / 2 / 3 blocks per CU and a long contraction (no epilogue anywhere while the loops run)"""
    K = 4096
    for per_cu in (1, 2, 3, 4):
        Mx = 128 * 32 * per_cu
        a, w, y = r(Mx, K), r(ff, K), torch.empty(Mx, ff, device=dev)
        fn = lambda: hip.call("vlg_linear_fwd", P(a), K, P(w), K, P(bias), P(y), ff, 0, 0, Mx, ff, K, EPI_BIAS, S)
        probe = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
        for _ in range(int(2500 / per_cu)):
            fn()
        lib.vlg_debug_set_clock_probe(probe.data_ptr())
        fn()
        torch.cuda.synchronize()
        lib.vlg_debug_set_clock_probe(None)
        nb = 256 * per_cu
        p = probe.cpu().view(-1, 8)[:nb]
        t = (p[:, :4] - p[:, 0].min()).double() / 100.0
        loop = t[:, 2] - t[:, 1]
        flop_blk = 2.0 * 128 * 128 * K
        print("blocks/CU %d (BK=%s): launch %.1f us, loop mean %.1f (min %.1f max %.1f) -> %.1f TFLOP/s over the mean loop, %.1f over the launch; first entry spread %.1f us" % (
            per_cu, os.environ.get("VLG_GEMM_BK", "32"), float(t[:, 3].max()), loop.mean(), loop.min(), loop.max(),
            nb * flop_blk / float(loop.mean()) * 1e-6, nb * flop_blk / float(t[:, 3].max()) * 1e-6, float(t[:, 0].max())))


if "sweep" in sys.argv:
    occupancy_sweep()
