#!/usr/bin/env python
"""Development probe: do the bandwidth-bound backward kernels (layer-norm backward, attention backward, slab reduction)
co-run with a weight-gradient GEMM when both are simply in flight on two HIP streams (no events between them)?
    python tools/micro/rider_overlap.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
lib = hip.load()
T, N, d, ff = 16, 64, 256, 1024
M = B * T * N
r = lambda *s: torch.randn(*s, device=dev)
P = lambda t: t.data_ptr()
x_d, y_d, g_d, g_ff, x_ff = r(M, d), r(M, d), r(M, d), r(M, ff), r(M, ff)
qkv, dqkv = r(M, 3 * d), r(M, 3 * d)
stats, gam = r(2, M).abs() + 0.5, r(d)
need = max(lib.vlg_linear_wgrad_slabs_for(M, n, k, 0) * (n * k + n) for (n, k) in ((3 * d, d), (d, d), (ff, d), (d, ff)))
slabs = torch.empty(need, device=dev)
lslabs = torch.empty(lib.vlg_layernorm_bwd_slabs(M) * 2 * d, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
gemms = {
    "wgrad ff1": lambda S: hip.call("vlg_linear_wgrad", P(g_ff), ff, P(x_d), d, P(slabs), ff * d + ff, slabs.numel(), M, ff, d, 0, S),
    "wgrad proj": lambda S: hip.call("vlg_linear_wgrad", P(g_d), d, P(x_d), d, P(slabs), d * d + d, slabs.numel(), M, d, d, 0, S),
    "dgrad ff1": lambda S: hip.call("vlg_linear_dgrad", P(g_ff), ff, P(r(ff, d)), d, P(y_d), d, 0, M, ff, d, 0, S),
}
wdg = r(ff, d)
gemms["dgrad ff1"] = lambda S: hip.call("vlg_linear_dgrad", P(g_ff), ff, P(wdg), d, P(y_d), d, 0, M, ff, d, 0, S)
riders = {
    "ln_bwd": lambda S: hip.call("vlg_layernorm_bwd", P(g_d), P(x_d), P(stats[0]), P(stats[1]), P(gam), P(x_d), P(y_d), P(lslabs), 2 * d, lslabs.numel(), M, d, S),
    "attn_bwd": lambda S: hip.call("vlg_attention_bwd", P(qkv), P(g_d), P(dqkv), B * N, T, d, S),
}
n = 100


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
for gn, gf in gemms.items():
    for rn, rf in riders.items():
        for _ in range(2):
            a, b_, both, both2 = run(gf, nop), run(nop, rf), run(gf, rf), run(rf, gf)
        print("%-10s %6.1f us | %-8s %6.1f us | serial %6.1f | two streams %6.1f (gemm first) %6.1f (rider first)" % (gn, a, rn, b_, a + b_, both, both2))
