#!/usr/bin/env python
"""Development probe (bf16 mode): a projection's data gradient and weight gradient on two HIP streams vs back to back.
    python tools/micro/pair_overlap_bf16.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip
from vlg.hip import EPI_A_BF16 as AB, EPI_B_BF16 as BB, EPI_OUT_BF16 as OB, EPI_BF16 as FL, EPI_MUL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
lib = hip.load()
d, ff, M = 256, 1024, B * 16 * 64
r = lambda *s: torch.randn(*s, device=dev)
bf = lambda t: t.to(torch.bfloat16)
P = lambda t: t.data_ptr()
x_d, g_d = r(M, d), r(M, d)                                   # fp32 residual-stream gradient
h_d, h_ff, h_3d, o_d, o_ff, aux = bf(r(M, d)), bf(r(M, ff)), bf(r(M, 3 * d)), bf(r(M, d)), bf(r(M, ff)), bf(r(M, ff))
w = {"qkv": bf(r(3 * d, d)), "proj": bf(r(d, d)), "ff1": bf(r(ff, d)), "ff2": bf(r(d, ff))}
need = max(lib.vlg_linear_wgrad_slabs_for(M, n, k, FL) * (n * k + n) for (n, k) in ((3 * d, d), (d, d), (ff, d), (d, ff)))
slabs = torch.empty(need, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cases = {
    "ff2": (lambda S: hip.call("vlg_linear_dgrad", P(g_d), d, P(w["ff2"]), ff, P(o_ff), ff, P(aux), M, d, ff, EPI_MUL | FL | BB | OB, S),
            lambda S: hip.call("vlg_linear_wgrad", P(g_d), d, P(h_ff), ff, P(slabs), d * ff + d, slabs.numel(), M, d, ff, FL | BB, S)),
    "ff1": (lambda S: hip.call("vlg_linear_dgrad", P(h_ff), ff, P(w["ff1"]), d, P(o_d), d, 0, M, ff, d, FL | AB | BB | OB, S),
            lambda S: hip.call("vlg_linear_wgrad", P(h_ff), ff, P(h_d), d, P(slabs), ff * d + ff, slabs.numel(), M, ff, d, FL | AB | BB, S)),
    "proj": (lambda S: hip.call("vlg_linear_dgrad", P(g_d), d, P(w["proj"]), d, P(o_d), d, 0, M, d, d, FL | BB | OB, S),
             lambda S: hip.call("vlg_linear_wgrad", P(g_d), d, P(h_d), d, P(slabs), d * d + d, slabs.numel(), M, d, d, FL | BB, S)),
    "qkv": (lambda S: hip.call("vlg_linear_dgrad", P(h_3d), 3 * d, P(w["qkv"]), d, P(o_d), d, 0, M, 3 * d, d, FL | AB | BB | OB, S),
            lambda S: hip.call("vlg_linear_wgrad", P(h_3d), 3 * d, P(h_d), d, P(slabs), 3 * d * d + 3 * d, slabs.numel(), M, 3 * d, d, FL | AB | BB, S)),
}
n = 200


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
tot_s = tot_b = 0.0
for name, (dg, wg) in cases.items():
    for _ in range(2):
        a, b_, both = run(dg, nop), run(nop, wg), run(dg, wg)
    tot_s += a + b_
    tot_b += both
    print("%-5s dgrad %.1f us, wgrad %.1f us, serial %.1f, two streams %.1f us" % (name, a, b_, a + b_, both))
print("per layer: serial %.1f us, two streams %.1f us" % (tot_s, tot_b))
