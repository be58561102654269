#!/usr/bin/env python
"""Diagnostic: in-kernel shader clock of the fp32 GEMM main loop (guide 'DVFS give-back' item 6):
clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, median over blocks, after 2 s of back-to-back launches."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
lib = hip.require_diag()
dev = torch.device("cuda:0")
M, N, K = 32768, 256, 1024
a, w, b, c, r = (torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev),
                 torch.empty(M, N, device=dev), torch.randn(M, N, device=dev))
S = torch.cuda.current_stream().cuda_stream
run = lambda: hip.call("vlg_linear_fwd", a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), c.data_ptr(), N, r.data_ptr(), 0, M, N, K, 5, S)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
probe = torch.zeros(2 * 512, dtype=torch.int64, device=dev)
lib.vlg_debug_set_clock_probe(probe.data_ptr())
for _ in range(20):
    run()
torch.cuda.synchronize()
lib.vlg_debug_set_clock_probe(None)
p = probe.cpu().view(-1, 2).double()
clk = (p[:, 0] / p[:, 1] * 0.1)
print("blocks %d  main-loop ticks median %.0f  realtime median %.1f us  clock GHz: median %.3f min %.3f max %.3f"
      % (len(clk), p[:, 0].median(), p[:, 1].median() / 100.0, clk.median(), clk.min(), clk.max()))
