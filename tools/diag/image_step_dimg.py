"""Diagnostic (GPU): where does ImageEngine's dimg differ from autograd of the restated loss."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F
from oracle import gridnet_spec as G, image_step_spec as S
from vlg import hip
from vlg.image_engine import ImageEngine, synthetic_frames

dev = torch.device("cuda:0")
b, H, W, filt = 2, 32, 40, (8, 16, 24)
eng = ImageEngine(b, H, W, dev, arch="GridNet", filters=filt)
p = G.test_params(G.param_shapes(10, filt), seed=5, linear=False)
eng.load_state_dict(p)
batch = synthetic_frames(b, H, W, seed=21)
eng.forward({k: v.to(dev) for k, v in batch.items()}, flip=False)
torch.cuda.synchronize()
img_h, f3 = eng.img.cpu(), eng.f3.cpu()
a = img_h.clone().requires_grad_(True)
terms = {"l1": 40 * F.l1_loss(a, f3), "gd": 20 * S.gradient_loss(a, f3), "ssim": 20 * S.ssim_loss(a, f3)}
gr = {}
for k, v in terms.items():
    a.grad = None
    v.backward(retain_graph=True)
    gr[k] = a.grad.clone()
want = gr["l1"] + gr["gd"] + gr["ssim"]
got = eng.dimg.cpu()
d = (got - want).abs()
print("dimg rel max", float(d.max() / want.abs().max()), "count > 1e-4", int((d > 1e-4 * want.abs().max()).sum()), "of", d.numel())
idx = torch.nonzero(d > 1e-4 * want.abs().max())
for i in idx[:12]:
    t = tuple(int(v) for v in i)
    print(t, "got", float(got[t]), "want", float(want[t]), {k: float(v[t]) for k, v in gr.items()}, "img", float(img_h[t]), "f3", float(f3[t]))
