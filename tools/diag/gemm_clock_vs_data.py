#!/usr/bin/env python
"""Diagnostic: is the fp32 GEMM's clock data-dependent?  Same launch (ff2 forward shape) on random-normal, all-zero and
constant operands: in-kernel clock (s_memtime / s_memrealtime) and achieved TFLOP/s after 1.5 s of back-to-back launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
lib = hip.require_diag()
dev = torch.device("cuda:0")
M, N, K = 32768, 256, 1024
S = torch.cuda.current_stream().cuda_stream
for tag, mk in (("randn", lambda *s: torch.randn(*s, device=dev)), ("zeros", lambda *s: torch.zeros(*s, device=dev)),
                ("ones", lambda *s: torch.ones(*s, device=dev)), ("randn", lambda *s: torch.randn(*s, device=dev))):
    a, w, b, r = mk(M, K), mk(N, K), mk(N), mk(M, N)
    c = torch.empty(M, N, device=dev)
    run = lambda: hip.call("vlg_linear_fwd", a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), c.data_ptr(), N, r.data_ptr(), 0, M, N, K, 5, S)
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    probe = torch.zeros(2 * 512, dtype=torch.int64, device=dev)
    lib.vlg_debug_set_clock_probe(probe.data_ptr())
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    lib.vlg_debug_set_clock_probe(None)
    p = probe.cpu().view(-1, 2).double()
    clk = (p[:, 0] / p[:, 1] * 0.1)
    print("%-6s %.1f us  %.1f TFLOP/s  clock %.3f GHz  main-loop ticks %.0f (MFMA cycles 2 x %d)" % (
        tag, us, 2.0 * M * N * K / us / 1e6, clk.median(), p[:, 0].median(), K // 32 * 4096))
