#!/usr/bin/env python
"""Diagnostic: which blocks of the fp32 weight-gradient GEMM are slow?  Per-block main-loop time grouped by XCD
(blockIdx & 7), by K split and by output tile."""
import os, sys
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
x_d, x_ff = r(M, d), r(M, ff)
slabs = torch.empty(160 * (ff * d + ff), device=dev)
n, k = ff, d
run = lambda: hip.call("vlg_linear_wgrad", P(x_ff), n, P(x_d), k, P(slabs), n * k + n, slabs.numel(), M, n, k, 0, S)
for _ in range(300):
    run()
torch.cuda.synchronize()
for rep in range(2):
    probe = torch.zeros(2 * 512, dtype=torch.int64, device=dev)
    lib.vlg_debug_set_clock_probe(probe.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.vlg_debug_set_clock_probe(None)
    p = probe.cpu().view(-1, 2).double()
    t = p[:, 1] / 100.0                      # us
    bid = torch.arange(512)
    nwg, q = 512, 64
    swz = (bid & 7) * q + (bid >> 3)         # rr = 0
    split, tile = swz // 16, swz % 16
    print("rep %d: loop us  min %.1f  median %.1f  max %.1f" % (rep, t.min(), t.median(), t.max()))
    print("  by XCD   :", [round(float(t[(bid & 7) == x].mean()), 1) for x in range(8)])
    print("  by split :", [round(float(t[split == s].mean()), 1) for s in range(0, 32, 4)])
    print("  by tile  :", [round(float(t[tile == s].mean()), 1) for s in range(16)])
    srt = torch.argsort(t, descending=True)[:12]
    print("  slowest  :", [(int(b), int(b) & 7, int(split[b]), int(tile[b]), round(float(t[b]), 1)) for b in srt])
