"""GPU, two processes sharing the one MI355X of the test box, gloo as transport (RCCL refuses two ranks on one
device): the REAL engine - HIP kernels, bucket hooks fired from backward, the split Adam that overlaps the last
bucket - must reproduce one process stepping on the union batch.  The multi-GPU RCCL run itself is the driver's;
bench.py rehearses the RCCL code path at world size 1 (VLG_FORCE_COMM=1)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu
CFG = dict(B=2, T=8, N=8, d=64, n_layers=2)


def _worker(rank, world, port, out, precision="fp32"):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, world_size=world, rank=rank)
    from oracle import layout_spec as O
    from vlg.dp import GradReducer, bucket_ranges
    from vlg.engine import LayoutEngine
    from vlg.spec import LayoutConfig
    dev = torch.device("cuda:0")
    cfg = LayoutConfig(**CFG)
    eng = LayoutEngine(cfg, dev, seed=1024, precision=precision)
    red = GradReducer(eng.grads_ext, bucket_ranges(eng.layout, eng.n_params, cfg.n_layers))
    losses = []
    for step in range(2):
        full = O.synthetic_batch(cfg.B * world, cfg.T, cfg.N, seed=60 + step)
        mine = {k: v[rank::world].contiguous().to(dev) for k, v in full.items()}
        loss = eng.train_step(mine, red)
        losses.append(float(loss[0]) / world)
    torch.cuda.synchronize()
    if rank == 0:
        shadow_ok = eng.params_bf16 is None or bool(torch.equal(eng.params_bf16, eng.params.to(torch.bfloat16)))
        torch.save({"params": eng.params.cpu(), "losses": losses, "shadow_ok": shadow_ok}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_match_one_big_batch(dev, tmp_path):
    world, out = 2, str(tmp_path / "dp.pt")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    from oracle import layout_spec as O
    from vlg.engine import LayoutEngine
    from vlg.spec import ADAM_LR, LayoutConfig
    cfg = LayoutConfig(**dict(CFG, B=CFG["B"] * world))
    eng = LayoutEngine(cfg, dev, seed=1024)
    signal = torch.ones(eng.n_params, dtype=torch.bool)
    for step in range(2):
        full = O.synthetic_batch(cfg.B, cfg.T, cfg.N, seed=60 + step)
        loss = float(eng.train_step({k: v.to(dev) for k, v in full.items()})[0])
        assert abs(got["losses"][step] - loss) <= 1e-4 * abs(loss)
        g = eng.grads.cpu()
        signal &= g.abs() > 1e-3 * float(g.abs().max())
    p = eng.params.cpu()
    # elements with a real gradient agree; zero-gradient ones (key bias) get +-lr of rounding noise from Adam
    assert torch.allclose(got["params"][signal], p[signal], rtol=1e-4, atol=1e-6)
    assert float((got["params"] - p).abs().max()) <= 2 * 2 * ADAM_LR * 1.01


def test_two_processes_bf16_mode(dev, tmp_path):
    """The bf16 mode under data parallelism: the Adam of the split-range path (everything but the embeddings first, the
    embedding range after its bucket arrives) must keep the bf16 weight shadow equal to the rounded master weights, and
    the step must track the single-process bf16 step on the union batch at the mode's tolerance."""
    world, out = 2, str(tmp_path / "dp16.pt")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, out, "bf16"), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    assert got["shadow_ok"]
    from oracle import layout_spec as O
    from vlg.engine import LayoutEngine
    from vlg.spec import LayoutConfig
    cfg = LayoutConfig(**dict(CFG, B=CFG["B"] * world))
    eng = LayoutEngine(cfg, dev, seed=1024, precision="bf16")
    for step in range(2):
        full = O.synthetic_batch(cfg.B, cfg.T, cfg.N, seed=60 + step)
        loss = float(eng.train_step({k: v.to(dev) for k, v in full.items()})[0])
        assert abs(got["losses"][step] - loss) <= 2e-2 * abs(loss), (got["losses"], loss)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (the driver's N > 1 command, as the reference's launcher is one
    command: src/main.py:183-185): the parent starts the two ranks itself, relays ONE JSON line and returns their status.  Both
    ranks share the one MI355X of the test box (VLG_BENCH_ONE_DEVICE=1, gloo: RCCL refuses two ranks on one device) - what is
    checked is the launch path and the communication evidence on the line, not a transport."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VLG_BENCH_ONE_DEVICE="1", VLG_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--settle", "0",
                        "--B", "4", "--T", "8", "--N", "16", "--d", "64", "--layers", "2", "--no-cpu-baseline", "--no-extras"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    c = d["comm"]
    assert c["world"] == 2 and c["backend"] == "gloo" and [x["rank"] for x in c["ranks"]] == [0, 1]
    assert c["buckets"] == 4 and c["bytes_allreduced_per_step"] > 0 and "exposed_comm_ms" in c
    assert d["value"] > 0 and d["ms_per_step"] > 0
