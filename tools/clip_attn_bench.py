#!/usr/bin/env python
"""Development tool: the per-clip attention kernels (csrc/attention_clip.hip) alone at the metric shape.
    python tools/clip_attn_bench.py [B] [T] [N] [d] [--valid]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg import hip

a = [x for x in sys.argv[1:] if not x.startswith("--")]
B, T, N, d = (int(a[i]) if len(a) > i else v for i, v in enumerate((32, 16, 64, 256)))
dev = torch.device("cuda:0")
hip.load()
M, S, heads = B * T * N, T * N, d // 64
qkv, dout = torch.randn(M, 3 * d, device=dev) * 0.7, torch.randn(M, d, device=dev)
out, dqkv = torch.empty(M, d, device=dev), torch.empty(M, 3 * d, device=dev)
lse, delta = torch.empty(B * heads * S, device=dev), torch.empty(B * heads * S, device=dev)
valid = torch.ones(B, T, N, device=dev) if "--valid" in sys.argv else None
P = lambda t: 0 if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
fwd = lambda: hip.call("vlg_attention_clip_fwd", P(qkv), P(valid), P(out), P(lse), B, T, N, d, st)
bwd = lambda: hip.call("vlg_attention_clip_bwd", P(qkv), P(valid), P(out), P(dout), P(lse), P(delta), P(dqkv), B, T, N, d, st)
fl = 4.0 * B * d * N * N * T * (T + 1) / 2.0
for name, fn, work in (("fwd", fwd, fl), ("bwd", bwd, 2.5 * fl)):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5)
    t = sorted(ts)[2]
    print("clip attention %s  (B,T,N,d)=(%d,%d,%d,%d)%s: %.1f us  %.1f TFLOP/s algorithmic (%.3f of 157.3)" % (name, B, T, N, d, " +valid" if valid is not None else "", t * 1e3, work / t / 1e9, work / t / 1e9 / 157.3))
