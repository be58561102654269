#!/usr/bin/env python
"""Development probe: what would co-running a layer's data gradient and weight gradient (independent given dY) buy at the
strong-scaling shard (M = 4 096 tokens)?  Two HIP streams, no events between them, vs the same launches back to back.
    python tools/micro/pair_overlap.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
from vlg.hip import EPI_NONE, EPI_MUL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
lib = hip.load()
d, ff, M = 256, 1024, B * 16 * 64
r = lambda *s: torch.randn(*s, device=dev)
P = lambda t: t.data_ptr()
x_d, x_ff, g_d, g_ff, aux = r(M, d), r(M, ff), r(M, d), r(M, ff), r(M, ff)
w = {"qkv": r(3 * d, d), "proj": r(d, d), "ff1": r(ff, d), "ff2": r(d, ff)}
x3 = r(M, 3 * d)
need = max(lib.vlg_linear_wgrad_slabs_for(M, n, k, 0) * (n * k + n) for (n, k) in ((3 * d, d), (d, d), (ff, d), (d, ff)))
slabs = torch.empty(need, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cases = {  # name: (dgrad launch, wgrad launch)   dY is [M, n], W [n, k]
    "ff2": (lambda S: hip.call("vlg_linear_dgrad", P(g_d), d, P(w["ff2"]), ff, P(g_ff), ff, P(aux), M, d, ff, EPI_MUL, S),
            lambda S: hip.call("vlg_linear_wgrad", P(g_d), d, P(x_ff), ff, P(slabs), d * ff + d, slabs.numel(), M, d, ff, 0, S), 4.0 * M * d * ff),
    "ff1": (lambda S: hip.call("vlg_linear_dgrad", P(g_ff), ff, P(w["ff1"]), d, P(g_d), d, 0, M, ff, d, EPI_NONE, S),
            lambda S: hip.call("vlg_linear_wgrad", P(g_ff), ff, P(x_d), d, P(slabs), ff * d + ff, slabs.numel(), M, ff, d, 0, S), 4.0 * M * d * ff),
    "proj": (lambda S: hip.call("vlg_linear_dgrad", P(g_d), d, P(w["proj"]), d, P(x_d), d, 0, M, d, d, EPI_NONE, S),
             lambda S: hip.call("vlg_linear_wgrad", P(g_d), d, P(x_d), d, P(slabs), d * d + d, slabs.numel(), M, d, d, 0, S), 4.0 * M * d * d),
    "qkv": (lambda S: hip.call("vlg_linear_dgrad", P(x3), 3 * d, P(w["qkv"]), d, P(g_d), d, 0, M, 3 * d, d, EPI_NONE, S),
            lambda S: hip.call("vlg_linear_wgrad", P(x3), 3 * d, P(x_d), d, P(slabs), 3 * d * d + 3 * d, slabs.numel(), M, 3 * d, d, 0, S), 12.0 * M * d * d),
}
n = 200
for name, (dg, wg, flops) in cases.items():
    def run(fa, fb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s1):
            for _ in range(n):
                fa(s1.cuda_stream)
        with torch.cuda.stream(s2):
            for _ in range(n):
                fb(s2.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    nop = lambda S: None
    for _ in range(2):
        a, b_, both = run(dg, nop), run(nop, wg), run(dg, wg)
    print("%-5s dgrad %.1f us, wgrad %.1f us, serial %.1f, two streams %.1f us  (%.0f TFLOP/s -> %.0f)" % (name, a, b_, a + b_, both, flops / (a + b_) / 1e6, flops / both / 1e6))
