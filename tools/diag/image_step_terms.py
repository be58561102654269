"""Diagnostic (GPU): which term of the reference step's loss gradient differs from the torch restatement."""
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
eng.backward()
img_h, seg_h, f3_h = eng.img.cpu(), eng.seg.cpu(), eng.f3.cpu()
# oracle forward
rec = {}
mean_arr = torch.tensor([-0.03, -0.088, -0.188])[None, :, None, None]
std_arr = torch.tensor([0.448, 0.448, 0.450])[None, :, None, None]
seg_o, img_o = G.forward(p, eng.x10.cpu(), False, branches=G.Branches(record=rec))
img_o = (img_o - mean_arr) / std_arr
print("fwd img rel", float((img_h - img_o).abs().max() / img_o.abs().max()), "seg rel", float((seg_h - seg_o).abs().max() / seg_o.abs().max()))
def grad_of(fn, a):
    a = a.clone().requires_grad_(True)
    fn(a).backward()
    return a.grad
S_ = torch.cuda.current_stream().cuda_stream
scr = torch.zeros(hip.load().vlg_image_loss_scratch(), device=dev)
loss = torch.zeros(4, device=dev)
for name, fn, call in (
    ("l1", lambda a: F.l1_loss(a, f3_h), lambda g: hip.call("vlg_l1_mean", eng.img.data_ptr(), eng.f3.data_ptr(), g.data_ptr(), loss.data_ptr(), scr.data_ptr(), eng.img.numel(), 1.0, S_)),
    ("gd", lambda a: S.gradient_loss(a, f3_h), lambda g: hip.call("vlg_gradient_loss", eng.img.data_ptr(), eng.f3.data_ptr(), g.data_ptr(), loss.data_ptr(), scr.data_ptr(), b * 3, H, W, 1.0, S_)),
    ("ssim", lambda a: S.ssim_loss(a, f3_h), lambda g: hip.call("vlg_ssim_loss", eng.img.data_ptr(), eng.f3.data_ptr(), g.data_ptr(), loss.data_ptr(), scr.data_ptr(), b, 3, H, W, 1.0, S_)),
):
    want = grad_of(fn, img_h)
    g = torch.zeros_like(eng.img)
    call(g)
    got = g.cpu()
    d = (got - want).abs()
    print(name, "grad rel max", float(d.max() / want.abs().max()), "L2", float(d.norm() / want.norm()), "n>1e-3", int((d > 1e-3 * want.abs().max()).sum()),
          "value", float(loss[0]), float(fn(img_h)))
want = grad_of(lambda a: F.cross_entropy(a, eng.seg3.cpu()), seg_h)
g = torch.zeros_like(eng.seg)
hip.call("vlg_ce_nchw", eng.seg.data_ptr(), eng.seg3.data_ptr(), g.data_ptr(), loss.data_ptr(), scr.data_ptr(), b, 20, H * W, 1.0, S_)
d = (g.cpu() - want).abs()
print("ce grad rel max", float(d.max() / want.abs().max()))
# full grads, pinned pattern
positive = {k: (v > 0).cpu() for k, v in eng.net.prelu_inputs().items()}
flips = {k: int(((rec[k] > 0) != positive[k]).sum()) for k in rec}
print("flips", sum(flips.values()), {k: v for k, v in flips.items() if v})
_, gp = S.loss_and_grads(p, batch, False, False, branches=G.Branches(positive=positive))
_, gu = S.loss_and_grads(p, batch, False, False)
got = eng.net.named_grads()
for k in list(gp)[:12]:
    w = gp[k]
    print(k, "pinned rel", float((got[k] - w).abs().max() / w.abs().max()), "unpinned rel", float((got[k] - gu[k]).abs().max() / gu[k].abs().max()))
# backward through the net from the ORACLE's dseg/dimg
print("---- all keys, pinned")
for k in gp:
    w = gp[k]
    e = float((got[k] - w).abs().max() / w.abs().max())
    if e > 1e-4:
        print(k, e)
# net backward alone from the oracle's loss gradients
x = eng.x10.cpu()
q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
seg_o, img_raw = G.forward(q, x, False)
seg_l = seg_o.detach().clone().requires_grad_(True)
img_l = img_raw.detach().clone().requires_grad_(True)
imgn = (img_l - mean_arr) / std_arr
f3 = eng.f3.cpu()
tot = 40 * F.l1_loss(imgn, f3) + 20 * (S.gradient_loss(imgn, f3) + S.ssim_loss(imgn, f3)) + 10 * F.cross_entropy(seg_l, eng.seg3.cpu())
tot.backward()
dseg_o, dimg_o = seg_l.grad, img_l.grad
print("dseg rel", float((eng.dseg.cpu() - dseg_o).abs().max() / dseg_o.abs().max()), "dimg_raw rel", float((eng.dtmp.cpu() - dimg_o).abs().max() / dimg_o.abs().max()))
(seg_o * dseg_o).sum().add((img_raw * dimg_o).sum()).backward()
eng.net.backward(dseg_o.to(dev), dimg_o.to(dev))
got2 = eng.net.named_grads()
print("---- net.backward from oracle loss gradients")
for k in q:
    w = q[k].grad
    e = float((got2[k] - w).abs().max() / w.abs().max())
    if e > 1e-4:
        print(k, e)
print("done")
