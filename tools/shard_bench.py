#!/usr/bin/env python
"""Development tool: the strong-scaling shard (the reference's GLOBAL batch of 32 on 8 GPUs = 4 clips per GPU,
src/trainer.py:148) on one MI355X - eager and replayed from the captured hipGraph.
    python tools/shard_bench.py [B] [steps] [--graph]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg.data import synthetic_clips, to_device
from vlg.engine import LayoutEngine
from vlg.spec import LayoutConfig, SEED, step_flops

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(argv[0]) if argv else 4
steps = int(argv[1]) if len(argv) > 1 else 50
dev = torch.device("cuda:0")
cfg = LayoutConfig(B=B, T=16, N=64, d=256, n_layers=4)
eng = LayoutEngine(cfg, dev, seed=SEED)
batch = to_device(synthetic_clips(cfg.B, cfg.T, cfg.N, seed=SEED), dev)
fl = step_flops(cfg)["fwd_bwd"]


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


dt = timed(lambda: eng.train_step(batch), steps)
print("eager : B=%d  %.4f ms/step  %.1f TFLOP/s (%.3f of 157.3)  loss %.5f" % (B, 1e3 * dt, fl / dt / 1e12, fl / dt / 157.3e12, float(eng.loss_out[0])))
if "--graph" in sys.argv:
    run = eng.capture_train_step(batch)
    dt = timed(lambda: run(batch), steps)
    print("graph : B=%d  %.4f ms/step  %.1f TFLOP/s (%.3f of 157.3)  loss %.5f" % (B, 1e3 * dt, fl / dt / 1e12, fl / dt / 157.3e12, float(eng.loss_out[0])))
