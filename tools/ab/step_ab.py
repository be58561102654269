"""A/B of whole-step variants in ONE process on one box (interleaved rounds): env-free switches on the engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg.data import synthetic_clips, to_device
from vlg.engine import LayoutEngine
from vlg.spec import LayoutConfig, SEED

dev = torch.device("cuda:0")
cfg = LayoutConfig(B=32, T=16, N=64, d=256, n_layers=4)
eng = LayoutEngine(cfg, dev, seed=SEED)
batch = to_device(synthetic_clips(cfg.B, cfg.T, cfg.N, seed=SEED), dev)
variants = {"serial": dict(overlap_wgrad=False, overlap_small=False), "overlap_wgrad": dict(overlap_wgrad=True, overlap_small=False),
            "overlap_small": dict(overlap_wgrad=False, overlap_small=True)}
if len(sys.argv) > 1 and sys.argv[1] == "gelu":
    variants = {"store_gl": dict(gelu_on_load=False), "gelu_on_load": dict(gelu_on_load=True)}
res = {k: [] for k in variants}
for rnd in range(6):
    for name, kv in variants.items():
        for k, v in kv.items():
            setattr(eng, k, v)
        for _ in range(3):
            eng.train_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            eng.train_step(batch)
        torch.cuda.synchronize()
        if rnd:
            res[name].append((time.perf_counter() - t0) / 15 * 1e3)
for k, v in res.items():
    v.sort()
    print("%-14s median %.3f ms  min %.3f" % (k, v[len(v) // 2], v[0]))
