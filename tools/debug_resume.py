import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd"), os.path.join(ROOT, "tests")]
import torch
from vlg.image_engine import ImageEngine, synthetic_frames
from vlg.gridnet import _Conv
dev = torch.device("cuda:0")
eng = ImageEngine(2, 32, 32, dev, arch="CoordGridNet", lr=2e-3)
torch.manual_seed(0)
sd = {k: (torch.randn(v) * 0.05 if len(v) > 1 else torch.full(v, 0.25)) for k, v in eng.net.reference_shapes().items()}
eng.load_state_dict(sd)
batch = {k: v.to(dev) for k, v in synthetic_frames(2, 32, 32, seed=3).items()}
for _ in range(3):
    eng.train_step(batch)
before = eng.net.params.clone()
eng.load_state_dict(eng.state_dict())
diff = (eng.net.params != before).nonzero().flatten()
print("mismatches", diff.numel())
net = eng.net
for d in diff[:12].tolist():
    where = None
    for op in net.tape:
        if isinstance(op, _Conv):
            if op.w_off <= d < op.w_off + op.out.cp * 9 * op.x.cp:
                r = d - op.w_off
                where = (op.key, "w", r // (9 * op.x.cp), (r // op.x.cp) % 9, r % op.x.cp, "cout", op.cout, "cin", op.cin)
            elif op.b_off <= d < op.b_off + op.out.cp:
                where = (op.key, "b", d - op.b_off, "cout", op.cout)
    for k, o in net.p_off.items():
        if o <= d < o + 4:
            where = (k, "slope slot", d - o)
    print(d, float(before[d]), where)
