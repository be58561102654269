#!/usr/bin/env python
"""A/B of fp32 GEMM builds in ONE process, interleaved rounds (guide rule 24: separate invocations differ by several per
cent on one box).  Every argument is `tag=path/to/libvlg.so[:BK[:RUN]]` (BK = forced contraction depth 16 | 32 | 0 through
vlg_debug_set_gemm_bk, RUN = argument of vlg_debug_set_gemm_run, e.g. 0x1ffff = chained tiles as ping-pong pairs, 0 = no
chaining; when the build exports them); `cur` = the in-tree build.
    python tools/ab/gemm_ab.py base=tools/ab/libvlg_base.so cur [--only gelu] [--rounds 7]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch  # noqa: E402

P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
SIG = {
    "vlg_linear_fwd": [P, I, P, I, P, P, I, P, P, L, I, I, I, P],
    "vlg_linear_dgrad": [P, I, P, I, P, I, P, L, I, I, I, P],
    "vlg_linear_wgrad": [P, I, P, I, P, L, L, L, I, I, I, P],
    "vlg_linear_wgrad_slabs_for": [L, I, I, I],
}
EPI_NONE, EPI_BIAS, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_GELU_GRAD, EPI_MUL = 0, 1, 2, 4, 8, 1024, 2048


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 7
    torch.zeros(1, device="cuda")          # PyTorch's HIP runtime first, as vlg/hip.py does
    libs = []
    for a in args:
        tag, _, rest = a.partition("=")
        path, _, bk = rest.partition(":")
        bk, _, run = bk.partition(":")
        path = path or os.path.join(ROOT, "video-layout-generation_amd", "libvlg_hip.so")
        lib = ctypes.CDLL(os.path.abspath(path))
        for n, sig in SIG.items():
            getattr(lib, n).argtypes = sig
            getattr(lib, n).restype = I
        libs.append((tag + (":" + bk if bk else ""), lib, int(bk) if bk else 0, int(run, 0) if run else -1))
    dev = torch.device("cuda:0")
    S = torch.cuda.current_stream().cuda_stream
    M, d = 32768, 256
    ff = 4 * d
    r = lambda *s: torch.randn(*s, device=dev)
    x_d, x_3d, x_ff, x_ff2 = r(M, d), r(M, 3 * d), r(M, ff), r(M, ff)
    y_d, y_3d, y_ff = r(M, d), r(M, 3 * d), r(M, ff)
    w_qkv, w_proj, w_ff1, w_ff2 = r(3 * d, d), r(d, d), r(ff, d), r(d, ff)
    bias = r(ff)
    slabs = torch.empty(200 * (ff * d + ff), device=dev)
    p = lambda t: t.data_ptr()
    fl = lambda n, k: 2.0 * M * n * k
    cases = [
        ("fwd qkv", fl(3 * d, d), lambda l: l.vlg_linear_fwd(p(x_d), d, p(w_qkv), d, p(bias), p(y_3d), 3 * d, 0, 0, M, 3 * d, d, EPI_BIAS, S)),
        ("fwd proj+resid", fl(d, d), lambda l: l.vlg_linear_fwd(p(x_d), d, p(w_proj), d, p(bias), p(y_d), d, p(x_d), 0, M, d, d, EPI_BIAS | EPI_RESID, S)),
        ("fwd ff1 gelu", fl(ff, d), lambda l: l.vlg_linear_fwd(p(x_d), d, p(w_ff1), d, p(bias), p(y_ff), ff, 0, p(x_ff2), M, ff, d, EPI_BIAS | EPI_GELU, S)),
        ("fwd ff1 gelu+grad", fl(ff, d), lambda l: l.vlg_linear_fwd(p(x_d), d, p(w_ff1), d, p(bias), p(y_ff), ff, 0, p(x_ff2), M, ff, d, EPI_BIAS | EPI_GELU | EPI_GELU_GRAD, S)),
        ("fwd ff1 u-only", fl(ff, d), lambda l: l.vlg_linear_fwd(p(x_d), d, p(w_ff1), d, p(bias), p(y_ff), ff, 0, 0, M, ff, d, EPI_BIAS, S)),
        ("fwd ff2+resid", fl(d, ff), lambda l: l.vlg_linear_fwd(p(x_ff), ff, p(w_ff2), ff, p(bias), p(y_d), d, p(x_d), 0, M, d, ff, EPI_BIAS | EPI_RESID, S)),
        ("dgrad qkv", fl(3 * d, d), lambda l: l.vlg_linear_dgrad(p(x_3d), 3 * d, p(w_qkv), d, p(y_d), d, 0, M, 3 * d, d, EPI_NONE, S)),
        ("dgrad proj", fl(d, d), lambda l: l.vlg_linear_dgrad(p(x_d), d, p(w_proj), d, p(y_d), d, 0, M, d, d, EPI_NONE, S)),
        ("dgrad ff1", fl(ff, d), lambda l: l.vlg_linear_dgrad(p(x_ff), ff, p(w_ff1), d, p(y_d), d, 0, M, ff, d, EPI_NONE, S)),
        ("dgrad ff2 dgelu", fl(d, ff), lambda l: l.vlg_linear_dgrad(p(x_d), d, p(w_ff2), ff, p(y_ff), ff, p(x_ff2), M, d, ff, EPI_DGELU, S)),
        ("dgrad ff2 mul", fl(d, ff), lambda l: l.vlg_linear_dgrad(p(x_d), d, p(w_ff2), ff, p(y_ff), ff, p(x_ff2), M, d, ff, EPI_MUL, S)),
    ]
    for nm, n, k, dy, xx in (("qkv", 3 * d, d, x_3d, x_d), ("proj", d, d, y_d, x_d), ("ff1", ff, d, x_ff, x_d), ("ff2", d, ff, x_d, x_ff)):
        cases.append(("wgrad " + nm, fl(n, k), lambda l, n=n, k=k, dy=dy, xx=xx: l.vlg_linear_wgrad(p(dy), n, p(xx), k, p(slabs), n * k + n, slabs.numel(), M, n, k, 0, S)))
    cases = [c for c in cases if only in c[0]]
    for tag, lib, bk, run in libs:                      # every build must accept every case (and size its slabs within the buffer)
        for nm, n, k in (("qkv", 3 * d, d), ("proj", d, d), ("ff1", ff, d), ("ff2", d, ff)):
            assert lib.vlg_linear_wgrad_slabs_for(M, n, k, 0) * (n * k + n) <= slabs.numel()
    times = {(c[0], t[0]): [] for c in cases for t in libs}
    for _ in range(300):                           # settle the clock
        cases[0][2](libs[0][1])
    for rnd in range(rounds + 1):
        for name, work, fn in cases:
            for tag, lib, bk, run in libs:
                if hasattr(lib, "vlg_debug_set_gemm_bk"):
                    lib.vlg_debug_set_gemm_bk(bk)
                if hasattr(lib, "vlg_debug_set_gemm_run"):
                    lib.vlg_debug_set_gemm_run(run)
                if fn(lib) != 0:                    # this build does not know the epilogue
                    continue
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn(lib)
                e.record()
                torch.cuda.synchronize()
                if rnd:
                    times[(name, tag)].append(s.elapsed_time(e) / 10)
    print("%-18s" % "case" + "".join("%22s" % t[0] for t in libs))
    tot = [0.0] * len(libs)
    for name, work, fn in cases:
        row = "%-18s" % name
        for i, (tag, lib, bk, run) in enumerate(libs):
            t = sorted(times[(name, tag)])
            if not t:
                row += "%22s" % "n/a"
                continue
            med = t[len(t) // 2]
            tot[i] += med
            row += "%9.1f us %5.1f TF/s" % (med * 1e3, work / med * 1e-9)
        print(row)
    print("%-18s" % "sum" + "".join("%9.1f us %10s" % (t * 1e3, "") for t in tot))


if __name__ == "__main__":
    main()
