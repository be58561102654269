#!/usr/bin/env python
"""Development tool: time GridNet / CoordGridNet forward + backward on one MI355X (reference-real conv path).
Algorithmic FLOPs from SURVEY.md section 6: 63.0 GFLOP fwd / 188.7 GFLOP fwd+bwd per 256x256 sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg.gridnet import GridNetHIP, _Conv
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
net = GridNetHIP(10, b, H, W, dev, coord=True)
torch.manual_seed(0)
sd = {k: (torch.randn(v) * 0.05 if len(v) > 1 else torch.full(v, 0.25)) for k, v in net.reference_shapes().items()}
net.load_state_dict(sd)
x = torch.randn(b, 10, H, W, device=dev)
ds, di = torch.randn(b, 20, H, W, device=dev), torch.randn(b, 3, H, W, device=dev)
fl_f = sum(2.0 * op.out.geo.b * op.out.geo.H * op.out.geo.W * 9 * op.cin * op.cout for op in net.tape if isinstance(op, _Conv))
for _ in range(2):
    net.forward(x); net.backward(ds, di)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 5
ev[0].record()
for _ in range(n):
    net.forward(x)
ev[1].record()
for _ in range(n):
    net.backward(ds, di)
ev[2].record()
torch.cuda.synchronize()
tf, tb = ev[0].elapsed_time(ev[1]) / n, ev[1].elapsed_time(ev[2]) / n
print("CoordGridNet b=%d %dx%d: fwd %.2f ms (%.1f TFLOP/s algorithmic), bwd %.2f ms (%.1f TFLOP/s), %.1f samples/s fwd+bwd"
      % (b, H, W, tf, fl_f / tf / 1e9, tb, 2 * fl_f / tb / 1e9, b / (tf + tb) * 1e3))
