import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]
import torch
from oracle import gridnet_spec as G
from vlg import hip
from vlg.gridnet import GridNetHIP, _Conv
dev = torch.device("cuda:0")
filters = (32, 64, 96)
b, H, W = 1, 64, 64
net = GridNetHIP(10, b, H, W, dev, filters=filters, need_input_grad=True)
p = G.test_params(G.param_shapes(10, filters), seed=2)
net.load_state_dict(p)
g = torch.Generator().manual_seed(4)
x = torch.randn(b, 10, H, W, generator=g)
r_seg, r_img = torch.randn(b, 20, H, W, generator=g), torch.randn(b, 3, H, W, generator=g)
seg, img = net.forward(x.to(dev))
S = torch.cuda.current_stream().cuda_stream
def to_nchw(t, C):
    o = torch.empty(b, C, t.geo.H, t.geo.W, device=dev)
    hip.call("vlg_padded_to_nchw", t.ptr, o.data_ptr(), b, C, t.geo.H, t.geo.W, t.cp, S)
    return o.cpu()
x0t = [op for op in net.tape if isinstance(op, _Conv) and op.key == "lateral_out_seg.conv.1"][0].x
x0 = to_nchw(x0t, 32).requires_grad_(True)
s2 = G._block(p, "lateral_out_seg", "lateral", x0)
i2 = G._block(p, "lateral_out_img", "lateral", x0)
((s2 * r_seg).sum() + (i2 * r_img).sum()).backward()
net.backward(r_seg.to(dev), r_img.to(dev))
g1 = to_nchw(x0t.grad, 32)
net.backward(r_seg.to(dev), r_img.to(dev))
g2 = to_nchw(x0t.grad, 32)
print("deterministic:", torch.equal(g1, g2))
err = (g1 - x0.grad).abs()
print("max err %.3e of max %.3e" % (float(err.max()), float(x0.grad.abs().max())))
bad = err > 1e-4 * float(x0.grad.abs().max())
print("bad count", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
print("bad channels:", sorted(set(idx[:, 1].tolist()))[:40])
print("bad rows:", sorted(set(idx[:, 2].tolist()))[:70])
print("bad cols:", sorted(set(idx[:, 3].tolist()))[:70])
# separately: only img head / only seg head
for nm, rs, ri in (("img only", torch.zeros_like(r_seg), r_img), ("seg only", r_seg, torch.zeros_like(r_img))):
    x0.grad = None
    ((G._block(p, "lateral_out_seg", "lateral", x0) * rs).sum() + (G._block(p, "lateral_out_img", "lateral", x0) * ri).sum()).backward()
    net.backward(rs.to(dev), ri.to(dev))
    gg = to_nchw(x0t.grad, 32)
    print(nm, "rel err %.3e" % float((gg - x0.grad).abs().max() / x0.grad.abs().max()))
print("---- wgrad of lateral_04.conv.3 from device buffers")
net.backward(r_seg.to(dev), r_img.to(dev))
gx0 = to_nchw(x0t.grad, 32)
op = [o for o in net.tape if isinstance(o, _Conv) and o.key == "lateral_04.conv.3"][0]
tin = to_nchw(op.x, 32)
slope = p["lateral_04.conv.2.weight"]
a = torch.nn.functional.prelu(tin, slope).requires_grad_(False)
w = p["lateral_04.conv.3.weight"].clone().requires_grad_(True)
bb = p["lateral_04.conv.3.bias"].clone().requires_grad_(True)
out = torch.nn.functional.conv2d(a, w, bb, padding=1)
(out * gx0).sum().backward()
gr = net.named_grads()
rel = lambda a_, w_: float((a_ - w_).abs().max() / w_.abs().max())
print("b from device dOut: rel", rel(gr["lateral_04.conv.3.bias"], bb.grad), " w:", rel(gr["lateral_04.conv.3.weight"], w.grad))
print("b sum check", rel(gr["lateral_04.conv.3.bias"], gx0.sum((0, 2, 3))))
# is grad[x0] the same as the full-net CPU gradient at x0?  compare bias grads of the heads' consumers instead
seg_w, img_w, grads_w, dx_w = G.forward_backward(p, x, r_seg, r_img)
print("vs full CPU: b", rel(gr["lateral_04.conv.3.bias"], grads_w["lateral_04.conv.3.bias"]), "head seg conv1 b", rel(gr["lateral_out_seg.conv.1.bias"], grads_w["lateral_out_seg.conv.1.bias"]))
print("CPU b", grads_w["lateral_04.conv.3.bias"][:6], "\nHIP b", gr["lateral_04.conv.3.bias"][:6], "\nsum  ", gx0.sum((0,2,3))[:6])
