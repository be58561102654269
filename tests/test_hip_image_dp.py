"""GPU, two processes sharing the one MI355X of the test box (gloo as transport: RCCL refuses two ranks on one device):
the REFERENCE-REAL step (CoordGridNet + L1 / GradientLoss / SSIM / CE, reference src/trainer.py:193-258) under data
parallelism - gradient buckets per grid column handed to vlg.dp.GradReducer from inside backward, the loss floats riding
in the last bucket, 1/world folded into Adam (reference: DDP wrapper trainer.py:113 + scalar sync :256) - must
reproduce ONE process stepping on the union batch: every loss term is a mean over its batch, so the mean of the two
shards' gradients is the union batch's gradient."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu
B, HW, FILT, STEPS = 2, 32, (8, 16, 24), 2


def _params(coord=True):
    from oracle import gridnet_spec as G
    return G.test_params(G.param_shapes(10, FILT, coord=coord), seed=4)


def _worker(rank, world, port, out):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, world_size=world, rank=rank)
    from vlg.dp import GradReducer
    from vlg.image_engine import ImageEngine, synthetic_frames
    dev = torch.device("cuda:0")
    eng = ImageEngine(B, HW, HW, dev, arch="CoordGridNet", filters=FILT, lr=2e-3)
    eng.load_state_dict(_params())
    red = GradReducer(eng.net.grads_ext, eng.net.bucket_ranges())
    tags = [t for t, _, _ in eng.net.bucket_ranges()]
    totals = []
    for step in range(STEPS):
        full = synthetic_frames(B * world, HW, HW, seed=70 + step)
        mine = {k: v[rank::world].contiguous().to(dev) for k, v in full.items()}
        totals.append(float(eng.train_step(mine, flip=bool(step % 2), reducer=red)) / world)
        assert not red.pending
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": eng.net.params.cpu(), "totals": totals, "tags": tags,
                    "losses": (eng.losses.cpu() / world)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_image_step_two_processes_match_one_big_batch(dev, tmp_path):
    world, out = 2, str(tmp_path / "dp_img.pt")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    assert got["tags"] == ["head", "col5", "col4", "col3", "col2", "col1", "in", "tail"]    # backward completion order
    from vlg.image_engine import ImageEngine, synthetic_frames
    from vlg.spec import ADAM_LR
    eng = ImageEngine(B * world, HW, HW, dev, arch="CoordGridNet", filters=FILT, lr=2e-3)
    eng.load_state_dict(_params())
    signal = torch.ones(eng.net.params.numel(), dtype=torch.bool)
    for step in range(STEPS):
        full = synthetic_frames(B * world, HW, HW, seed=70 + step)
        # rank r took clips r, r+world, ...: the union batch in the same clip order
        total = float(eng.train_step({k: v.to(dev) for k, v in full.items()}, flip=bool(step % 2)))
        assert abs(got["totals"][step] - total) <= 1e-4 * abs(total), (got["totals"], total)
        g = eng.net.grads.cpu()
        signal &= g.abs() > 1e-3 * float(g.abs().max())
    want_l = eng.losses.cpu()
    assert torch.allclose(got["losses"][:5], want_l[:5], rtol=1e-4, atol=1e-7)              # every term, not just the total
    p = eng.net.params.cpu()
    assert int(signal.sum()) > 1000
    # elements with a real gradient agree; elements with a near-zero one get +-lr of rounding noise from Adam's sign
    assert torch.allclose(got["params"][signal], p[signal], rtol=1e-4, atol=2e-6)
    assert float((got["params"] - p).abs().max()) <= 2 * STEPS * 2e-3 * 1.01


def test_bucketed_backward_equals_single_reduction(dev):
    """The per-bucket slab reductions (reducer attached) produce bit for bit the gradients of the one-launch reduction."""
    from vlg.image_engine import ImageEngine, synthetic_frames

    class Recorder:
        world, grad_scale, pending = 1, 1.0, []

        def __init__(self, net):
            self.net, self.seen = net, []

        def ready(self, tag):
            self.seen.append(tag)

        def wait(self, keep=()):
            pass

    eng = ImageEngine(B, HW, HW, dev, arch="GridNet", filters=FILT)
    eng.load_state_dict(_params(coord=False))
    batch = {k: v.to(dev) for k, v in synthetic_frames(B, HW, HW, seed=5).items()}
    eng.forward(batch)
    eng.backward()
    g0 = eng.net.grads.clone()
    eng.net.grads.zero_()
    rec = Recorder(eng.net)
    eng.forward(batch)
    eng.backward(rec)
    assert rec.seen == [t for t, _, _ in eng.net.bucket_ranges()]
    assert torch.equal(eng.net.grads, g0)
    ranges = sorted((s, e) for _, s, e in eng.net.bucket_ranges())
    assert ranges[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and ranges[-1][1] == eng.net.grads_ext.numel()
