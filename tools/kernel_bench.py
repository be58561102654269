#!/usr/bin/env python
"""Per-kernel micro-benchmark on one MI355X (development tool; not part of the product path).

Times every kernel of the step at the metric shape with HIP events, interleaved rounds in one
process (guide rule 24), and prints achieved TFLOP/s or GB/s against the algorithmic work.
    python tools/kernel_bench.py [--rounds 5] [--only gemm]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch  # noqa: E402

from vlg import hip  # noqa: E402
from vlg.hip import EPI_ACT_GELU, EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_NONE, EPI_RESID  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--N", type=int, default=64)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--bf16store", action="store_true", help="the bf16 mode's kernels with its HBM element types (GB/s shown)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = hip.load()
    S = torch.cuda.current_stream().cuda_stream
    B, T, N, d = a.B, a.T, a.N, a.d
    M, ff = B * T * N, 4 * d
    r = lambda *s: torch.randn(*s, device=dev)
    x_d, x_3d, x_ff, x_ff2 = r(M, d), r(M, 3 * d), r(M, ff), r(M, ff)
    y_d, y_3d, y_ff = r(M, d), r(M, 3 * d), r(M, ff)
    w_qkv, w_proj, w_ff1, w_ff2 = r(3 * d, d), r(d, d), r(ff, d), r(d, ff)
    bias = r(ff)
    # sized from the library's own split plan (never a guess: a short buffer here is an out-of-bounds GPU write)
    need = max(lib.vlg_linear_wgrad_slabs_for(M, n, k, fl_) * (n * k + n) for (n, k) in ((3 * d, d), (d, d), (ff, d), (d, ff))
               for fl_ in (0, 16, 256))
    slabs = torch.empty(need + 1024, device=dev)
    red_dst = torch.empty(ff * d + ff, device=dev)            # reduce target: [w | b] of the largest projection
    stats = r(2, M)
    g = r(d)
    FLAGS = 16 if a.bf16 else 0
    cases = []

    def add(name, work, unit, fn):
        if a.only in name:
            cases.append((name, work, unit, fn))

    P = lambda t: t.data_ptr()
    fl = lambda n, k: 2.0 * M * n * k
    add("gemm fwd qkv  (bias)", fl(3 * d, d), "F", lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_qkv), d, P(bias), P(y_3d), 3 * d, 0, 0, M, 3 * d, d, EPI_BIAS | FLAGS, S))
    add("gemm fwd proj (bias+resid)", fl(d, d), "F", lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_proj), d, P(bias), P(y_d), d, P(x_d), 0, M, d, d, EPI_BIAS | EPI_RESID | FLAGS, S))
    add("gemm fwd ff1  (bias+gelu)", fl(ff, d), "F", lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_ff1), d, P(bias), P(y_ff), ff, 0, P(x_ff2), M, ff, d, EPI_BIAS | EPI_GELU | FLAGS, S))
    add("gemm fwd ff2  (bias+resid)", fl(d, ff), "F", lambda: hip.call("vlg_linear_fwd", P(x_ff), ff, P(w_ff2), ff, P(bias), P(y_d), d, P(x_d), 0, M, d, ff, EPI_BIAS | EPI_RESID | FLAGS, S))
    if not a.bf16:   # native fp32 step: gelu(u) is never stored (VLG_EPI_ACT_GELU)
        add("gemm fwd ff1  (bias, u only)", fl(ff, d), "F", lambda: hip.call("vlg_linear_fwd", P(x_d), d, P(w_ff1), d, P(bias), P(y_ff), ff, 0, 0, M, ff, d, EPI_BIAS, S))
        add("gemm fwd ff2  (gelu on load)", fl(d, ff), "F", lambda: hip.call("vlg_linear_fwd", P(x_ff), ff, P(w_ff2), ff, P(bias), P(y_d), d, P(x_d), 0, M, d, ff, EPI_BIAS | EPI_RESID | EPI_ACT_GELU, S))
    # asymptote of the main loop: the same tile with a long contraction (prologue / epilogue amortised over 128 K tiles)
    x_long, w_long = r(M // 4, 4096), r(ff, 4096)
    add("gemm fwd asymptote K=4096 (M/4)", 2.0 * (M // 4) * ff * 4096, "F", lambda: hip.call("vlg_linear_fwd", P(x_long), 4096, P(w_long), 4096, P(bias), P(y_ff), ff, 0, 0, M // 4, ff, 4096, EPI_BIAS | FLAGS, S))
    add("gemm dgrad qkv", fl(3 * d, d), "F", lambda: hip.call("vlg_linear_dgrad", P(x_3d), 3 * d, P(w_qkv), d, P(y_d), d, 0, M, 3 * d, d, EPI_NONE | FLAGS, S))
    add("gemm dgrad proj", fl(d, d), "F", lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(w_proj), d, P(y_d), d, 0, M, d, d, EPI_NONE | FLAGS, S))
    add("gemm dgrad ff1", fl(ff, d), "F", lambda: hip.call("vlg_linear_dgrad", P(x_ff), ff, P(w_ff1), d, P(y_d), d, 0, M, ff, d, EPI_NONE | FLAGS, S))
    add("gemm dgrad ff2 (dgelu)", fl(d, ff), "F", lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(w_ff2), ff, P(y_ff), ff, P(x_ff2), M, d, ff, EPI_DGELU | FLAGS, S))
    for nm, n, k, dy, xx in (("qkv", 3 * d, d, x_3d, x_d), ("proj", d, d, y_d, x_d), ("ff1", ff, d, x_ff, x_d), ("ff2", d, ff, x_d, x_ff)):
        ns = lib.vlg_linear_wgrad_slabs_for(M, n, k, FLAGS)
        add("gemm wgrad %-4s (%d slabs)" % (nm, ns), fl(n, k), "F", lambda n=n, k=k, dy=dy, xx=xx: hip.call("vlg_linear_wgrad", P(dy), n, P(xx), k, P(slabs), n * k + n, slabs.numel(), M, n, k, FLAGS, S))
        add("reduce wgrad %-4s" % nm, 4.0 * (n * k + n) * (ns + 1), "B", lambda n=n, k=k, ns=ns: hip.call("vlg_reduce_slabs", P(slabs), n * k + n, ns, P(red_dst), n * k + n, S))
    if not a.bf16:
        ns = lib.vlg_linear_wgrad_slabs_for(M, d, ff, 0)
        add("gemm wgrad ff2 (gelu on load)", fl(d, ff), "F", lambda: hip.call("vlg_linear_wgrad", P(x_d), d, P(x_ff), ff, P(slabs), d * ff + d, slabs.numel(), M, d, ff, EPI_ACT_GELU, S))
    add("attention fwd", 16.0 * M * d, "B", lambda: hip.call("vlg_attention_fwd", P(x_3d), P(y_d), B * N, T, d, S))
    add("attention bwd", 28.0 * M * d, "B", lambda: hip.call("vlg_attention_bwd", P(x_3d), P(x_d), P(y_3d), B * N, T, d, S))
    vocab = 21
    ids = torch.randint(0, vocab, (B, T, N), device=dev)
    boxes = torch.rand(B, T, N, 4, device=dev)
    tabs = r(vocab * d + d * 4 + d + T * d)
    L_emb = vocab * d + d * 4 + d + T * d
    eslabs = torch.empty(lib.vlg_embed_bwd_slabs_for(B, T, N, d, vocab) * L_emb, device=dev)
    add("embed fwd", 4.0 * M * d + 24.0 * M, "B", lambda: hip.call("vlg_embed_fwd", P(ids), P(boxes), P(tabs), tabs.data_ptr() + 4 * vocab * d, tabs.data_ptr() + 4 * (vocab * d + 4 * d),
                                                                     tabs.data_ptr() + 4 * (vocab * d + 5 * d), P(y_d), B, T, N, d, vocab, S))
    add("embed bwd", 4.0 * M * d + 24.0 * M, "B", lambda: hip.call("vlg_embed_bwd", P(x_d), P(ids), P(boxes), P(eslabs), L_emb, eslabs.numel(), B, T, N, d, vocab, S))
    add("embed bwd reduce", 4.0 * L_emb * (lib.vlg_embed_bwd_slabs_for(B, T, N, d, vocab) + 1), "B", lambda: hip.call("vlg_reduce_slabs", P(eslabs), L_emb, lib.vlg_embed_bwd_slabs_for(B, T, N, d, vocab), P(red_dst), L_emb, S))
    hout, hdout = r(M, 24), torch.empty(M, 24, device=dev)
    tcls, tbox, tvalid = torch.randint(0, 20, (B, T, N), device=dev), torch.rand(B, T, N, 4, device=dev) * 0.5 + 0.25, torch.ones(B, T, N, device=dev)
    lscr, lout = torch.zeros(lib.vlg_layout_loss_scratch(), device=dev), torch.zeros(4, device=dev)
    add("layout loss (value + gradient)", (2.0 * 96 + 28) * M, "B", lambda: hip.call("vlg_layout_loss", P(hout), 24, P(tcls), P(tbox), P(tvalid), P(hdout), P(lout), P(lscr),
                                                                                B, T, N, 20, 0.1, 1e-7, 40.0, 20.0, 10.0, S))
    add("layernorm fwd", 8.0 * M * d, "B", lambda: hip.call("vlg_layernorm_fwd", P(x_d), P(g), P(g), P(y_d), P(stats[0]), P(stats[1]), M, d, 1e-5, S))
    add("layernorm bwd", 16.0 * M * d, "B", lambda: hip.call("vlg_layernorm_bwd", P(x_d), P(y_d), P(stats[0]), P(stats[1]), P(g), P(x_d), P(y_d), P(slabs), 2 * d, slabs.numel(), M, d, S))

    if a.bf16store:
        from vlg.hip import EPI_A_BF16 as AB, EPI_B_BF16 as BB, EPI_OUT_BF16 as OB, EPI_BF16 as FL
        cases.clear()
        bf = lambda t: t.to(torch.bfloat16)
        hd, h3, hf, hf2 = bf(x_d), bf(x_3d), bf(x_ff), bf(x_ff2)          # bf16 activations
        od, o3, of = bf(y_d), bf(y_3d), bf(y_ff)
        by = lambda *terms: float(sum(terms))
        E = M * d
        # the storage combinations the bf16 step launches (csrc/gemm_bf16.hip instantiates exactly these): bf16 activations,
        # the bf16 weight shadow, fp32 residual stream / its gradient
        wq16, wp16, wf16, wf216 = bf(w_qkv), bf(w_proj), bf(w_ff1), bf(w_ff2)
        add("fwd qkv  bf16,bf16W->bf16", by(2 * E, 6 * E), "B", lambda: hip.call("vlg_linear_fwd", P(hd), d, P(wq16), d, P(bias), P(o3), 3 * d, 0, 0, M, 3 * d, d, EPI_BIAS | FL | AB | BB | OB, S))
        add("fwd proj bf16,bf16W->f32 +resid", by(2 * E, 4 * E, 4 * E), "B", lambda: hip.call("vlg_linear_fwd", P(hd), d, P(wp16), d, P(bias), P(y_d), d, P(x_d), 0, M, d, d, EPI_BIAS | EPI_RESID | FL | AB | BB, S))
        add("fwd ff1  bf16,bf16W->2xbf16 gelu", by(2 * E, 16 * E), "B", lambda: hip.call("vlg_linear_fwd", P(hd), d, P(wf16), d, P(bias), P(of), ff, 0, P(hf2), M, ff, d, EPI_BIAS | EPI_GELU | FL | AB | BB | OB, S))
        add("fwd ff2  bf16,bf16W->f32 +resid", by(8 * E, 4 * E, 4 * E), "B", lambda: hip.call("vlg_linear_fwd", P(hf), ff, P(wf216), ff, P(bias), P(y_d), d, P(x_d), 0, M, d, ff, EPI_BIAS | EPI_RESID | FL | AB | BB, S))
        add("dgrad qkv  bf16,bf16W->bf16", by(6 * E, 2 * E), "B", lambda: hip.call("vlg_linear_dgrad", P(h3), 3 * d, P(wq16), d, P(od), d, 0, M, 3 * d, d, FL | AB | BB | OB, S))
        add("dgrad proj f32,bf16W->bf16", by(4 * E, 2 * E), "B", lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(wp16), d, P(od), d, 0, M, d, d, FL | BB | OB, S))
        add("dgrad ff1  bf16,bf16W->bf16", by(8 * E, 2 * E), "B", lambda: hip.call("vlg_linear_dgrad", P(hf), ff, P(wf16), d, P(od), d, 0, M, ff, d, FL | AB | BB | OB, S))
        add("dgrad ff2  f32,bf16W->bf16 dgelu", by(4 * E, 8 * E, 8 * E), "B", lambda: hip.call("vlg_linear_dgrad", P(x_d), d, P(wf216), ff, P(of), ff, P(hf2), M, d, ff, EPI_DGELU | FL | BB | OB, S))
        for nm, n, k, dy, xx, bits, nb in (("qkv  bf16,bf16", 3 * d, d, h3, hd, AB | BB, by(6 * E, 2 * E)), ("proj f32,bf16", d, d, y_d, hd, BB, by(4 * E, 2 * E)),
                                           ("ff1  bf16,bf16", ff, d, hf, hd, AB | BB, by(8 * E, 2 * E)), ("ff2  f32,bf16", d, ff, x_d, hf, BB, by(4 * E, 8 * E))):
            ns = lib.vlg_linear_wgrad_slabs_for(M, n, k, FL)
            add("wgrad %s (%d slabs)" % (nm, ns), nb + 4.0 * ns * (n * k + n), "B", lambda n=n, k=k, dy=dy, xx=xx, bits=bits: hip.call("vlg_linear_wgrad", P(dy), n, P(xx), k, P(slabs), n * k + n, slabs.numel(), M, n, k, FL | bits, S))
        add("attention fwd bf16", 8.0 * M * d, "B", lambda: hip.call("vlg_attention_fwd_bf16", P(h3), P(od), B * N, T, d, S))
        add("attention bwd bf16", 14.0 * M * d, "B", lambda: hip.call("vlg_attention_bwd_bf16", P(h3), P(hd), P(o3), B * N, T, d, S))
        add("layernorm fwd ->bf16", 6.0 * M * d, "B", lambda: hip.call("vlg_layernorm_fwd_bf16", P(x_d), P(g), P(g), P(od), P(stats[0]), P(stats[1]), M, d, 1e-5, S))
        add("layernorm bwd bf16 dy", 14.0 * M * d, "B", lambda: hip.call("vlg_layernorm_bwd_bf16", P(hd), P(y_d), P(stats[0]), P(stats[1]), P(g), P(x_d), P(x_d), P(slabs), 2 * d, slabs.numel(), M, d, S))

    times = {c[0]: [] for c in cases}
    for rnd in range(a.rounds + 1):
        for name, work, unit, fn in cases:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            if rnd:
                times[name].append(s.elapsed_time(e) / a.iters)
    for name, work, unit, fn in cases:
        t = sorted(times[name])
        med = t[len(t) // 2] * 1e-3
        rate = work / med
        print("%-32s %8.1f us (min %7.1f)  %s" % (name, med * 1e6, t[0] * 1e3,
              "%6.1f TFLOP/s (%4.1f%% of 157.3)" % (rate / 1e12, rate / 1.573e12) if unit == "F"
              else "%6.0f GB/s (%4.1f%% of 8000)" % (rate / 1e9, rate / 8e10)))


if __name__ == "__main__":
    main()
