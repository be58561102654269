#!/usr/bin/env python
"""Diagnostic: main-loop ticks of each fp32 GEMM shape of the step (clock probe in gemm_f32_kernel) next to its MFMA
cycles: separates loop efficiency from prologue / epilogue / launch overhead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
lib = hip.require_diag()
dev = torch.device("cuda:0")
M, d = 32768, 256
ff = 4 * d
S = torch.cuda.current_stream().cuda_stream
r = lambda *s: torch.randn(*s, device=dev)
P = lambda t: t.data_ptr()
x_d, x_3d, x_ff, y_d, y_3d, y_ff, u = r(M, d), r(M, 3 * d), r(M, ff), r(M, d), r(M, 3 * d), r(M, ff), r(M, ff)
w_qkv, w_proj, w_ff1, w_ff2, bias = r(3 * d, d), r(d, d), r(ff, d), r(d, ff), r(ff)
slabs = torch.empty(160 * (ff * d + ff), device=dev)
cases = {
    "fwd qkv": (lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_qkv), d, P(bias), P(y_3d), 3 * d, 0, 0, M, 3 * d, d, 1, S), 3 * d, d, d // 32),
    "fwd ff2": (lambda: hip.call("vlg_linear_fwd", P(x_ff), ff, P(w_ff2), ff, P(bias), P(y_d), d, P(x_d), 0, M, d, ff, 5, S), d, ff, ff // 32),
    "dgrad qkv": (lambda: hip.call("vlg_linear_dgrad", P(x_3d), 3 * d, P(w_qkv), d, P(y_d), d, 0, M, 3 * d, d, 0, S), 3 * d, d, 3 * d // 32),
}
for nm, n, k, dy, xx in (("wgrad qkv", 3 * d, d, x_3d, x_d), ("wgrad ff1", ff, d, x_ff, x_d), ("wgrad proj", d, d, y_d, x_d)):
    ns = lib.vlg_linear_wgrad_slabs(M, n, k)
    per = (M + ns - 1) // ns
    per = (per + 31) // 32 * 32
    cases[nm] = ((lambda n=n, k=k, dy=dy, xx=xx: hip.call("vlg_linear_wgrad", P(dy), n, P(xx), k, P(slabs), n * k + n, slabs.numel(), M, n, k, 0, S)), n, k, per // 32)
for name, (run, n, k, nk) in cases.items():
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    probe = torch.zeros(2 * 4096, dtype=torch.int64, device=dev)
    lib.vlg_debug_set_clock_probe(probe.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.vlg_debug_set_clock_probe(None)
    p = probe.cpu().view(-1, 2).double()
    p = p[p[:, 1] > 0]
    clk = (p[:, 0] / p[:, 1] * 0.1).median()
    print("%-10s %6.1f us  %5.1f TFLOP/s | blocks %4d  k-tiles %2d  loop ticks median %7.0f max %7.0f = %.3f / %.3f of 2 x MFMA cycles (%d)  loop time %.1f us (max %.1f)  clock %.2f GHz"
          % (name, us, 2.0 * M * n * k / us / 1e6, len(p), nk, p[:, 0].median(), p[:, 0].max(), 2 * nk * 4096 / p[:, 0].median(),
             2 * nk * 4096 / p[:, 0].max(), 2 * nk * 4096, p[:, 1].median() / 100, p[:, 1].max() / 100, clk))
