#!/usr/bin/env python
"""Development tool: the reference's OWN training step (SURVEY.md section 8d "Spec R": CoordGridNet + frozen HED x2 +
40 L1 + 20 (VGG + GradientLoss + SSIM) + 10 CE + Adam, 256x256 frames) on one MI355X, next to its torch-CPU
restatement on the host cores.  Algorithmic work per sample-step: GridNet 188.7 GFLOP fwd+bwd,
HED 2 x 40.1 GFLOP (the third, tensorboard-only call of trainer.py:214-216 is not made), VGG19[:27] 2 x 46.1 fwd +
~46 input-gradient.   python tools/reference_step_bench.py [batch] [cpu_batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from oracle import gridnet_spec as G, hned_spec as HS, vgg_spec as V, image_step_spec as S
from vlg.image_engine import ImageEngine, synthetic_frames

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H = W = 256
dev = torch.device("cuda:0")
eng = ImageEngine(b, H, W, dev, arch="CoordGridNet", with_hed=True, with_vgg=True)
p = G.test_params(G.param_shapes(10, coord=True), seed=0)
hp, vp = HS.test_params(0), V.test_params(0)
eng.load_state_dict(p); eng.hed.load_state_dict(hp); eng.vgg.load_state_dict(vp)
batch = {k: v.to(dev) for k, v in synthetic_frames(b, H, W, seed=1).items() if k not in ("e1", "e2")}
for _ in range(2):
    eng.train_step(batch)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    eng.train_step(batch)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / n
gflop = b * (188.7 + 2 * 40.1 + 3 * 46.1)
print("GPU  : b=%d  %.2f ms/step  %.1f samples/s  (%.1f TFLOP/s algorithmic over %.0f GFLOP/step)" % (b, gpu * 1e3, b / gpu, gflop / gpu / 1e3, gflop))
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
cpu_batch = synthetic_frames(cb, H, W, seed=1)
with torch.no_grad():
    cpu_batch["e1"] = HS.forward(hp, cpu_batch["frame1"])[5]
t0 = time.perf_counter()
with torch.no_grad():
    cpu_batch["e1"] = HS.forward(hp, cpu_batch["frame1"])[5]
    cpu_batch["e2"] = HS.forward(hp, cpu_batch["frame2"])[5]
parts, grads = S.loss_and_grads(p, cpu_batch, True, vgg_params=vp)
cpu = time.perf_counter() - t0
print("CPU  : b=%d  %.2f s/step  %.2f samples/s on %d threads (torch-CPU restatement, no Adam)" % (cb, cpu, cb / cpu, torch.get_num_threads()))
print("ratio: %.0fx" % ((b / gpu) / (cb / cpu)))
