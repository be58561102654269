"""Whole-step time of the in-tree engine with whatever libvlg VLG_HIP_LIB names (one process per variant, same box)."""
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
res = []
for rnd in range(6):
    for _ in range(3):
        eng.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        eng.train_step(batch)
    torch.cuda.synchronize()
    if rnd:
        res.append((time.perf_counter() - t0) / 15 * 1e3)
res.sort()
print("%s: median %.3f ms  min %.3f  loss %.5f" % (os.environ.get("TAG", "?"), res[len(res) // 2], res[0], float(eng.loss_out[0])))
