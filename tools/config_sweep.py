#!/usr/bin/env python
"""Development tool: every BASELINE.json configuration (and ragged shapes) x every projection mode takes two training
steps; reports the loss (must be finite) - a shape a mode cannot launch raises HipError here, not in production."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from vlg.data import synthetic_clips, to_device
from vlg.engine import LayoutEngine
from vlg.spec import LayoutConfig

dev = torch.device("cuda:0")
CONFIGS = [dict(B=4, T=4, N=8, d=64, n_layers=2), dict(B=32, T=16, N=32, d=256, n_layers=4), dict(B=8, T=32, N=64, d=512, n_layers=4),
           dict(B=32, T=16, N=64, d=256, n_layers=4), dict(B=3, T=8, N=5, d=128, n_layers=1), dict(B=1, T=16, N=1, d=64, n_layers=1),
           dict(B=5, T=16, N=24, d=192, n_layers=2), dict(B=2, T=32, N=9, d=768, n_layers=1)]
bad = 0
for kw in CONFIGS:
    cfg = LayoutConfig(**kw)
    batch = to_device(synthetic_clips(cfg.B, cfg.T, cfg.N, seed=1, variable_n=cfg.N > 4, min_valid=1), dev)
    row = []
    for prec in ("fp32", "fp32x3", "bf16", "bf16_mfma"):
        try:
            eng = LayoutEngine(cfg, dev, precision=prec)
            for _ in range(2):
                loss = eng.train_step(batch)
            v = float(loss[0])
            ok = v == v and abs(v) < 1e9 and bool(torch.isfinite(eng.params).all())
            row.append("%s %.4f%s" % (prec, v, "" if ok else " NOT FINITE"))
            bad += 0 if ok else 1
            del eng
        except Exception as e:                                  # noqa: BLE001
            row.append("%s FAILED: %s" % (prec, str(e)[:80]))
            bad += 1
    print(kw, "|", " | ".join(row), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
