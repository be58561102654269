"""CPU, world_size 2 over gloo: the data-parallel path (clip sharding, bucketed asynchronous
gradient all-reduce with the loss scalars riding in the first bucket, mean folded into Adam,
Trainer.sync / validate semantics).  Compute comes from the oracle-backed test double; the
exchange code under test (vlg/dp.py, trainer.py) is the same code that runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

CFG = dict(B=2, T=4, N=8, d=64, n_layers=2)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, world, port):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, world_size=world, rank=rank)


def _dp_worker(rank, world, port, out):
    _setup(rank, world, port)
    from helpers import OracleEngine
    from oracle import layout_spec as O
    from vlg.dp import GradReducer, bucket_ranges
    from vlg.spec import LayoutConfig
    cfg = LayoutConfig(**CFG)
    eng = OracleEngine(cfg, seed=1024)                               # same seed on every rank (main.py:57-60)
    red = GradReducer(eng.grads_ext, bucket_ranges(eng.layout, eng.n_params, cfg.n_layers))
    assert red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
    losses, grads = [], []
    for step in range(2):
        full = O.synthetic_batch(cfg.B * world, cfg.T, cfg.N, seed=50 + step)
        mine = {k: v[rank::world].contiguous() for k, v in full.items()}     # DistributedSampler stride
        loss = eng.train_step(mine, red)
        losses.append(float(loss[0]) / world)                               # summed in the head bucket
        grads.append(eng.grads.clone() * red.grad_scale)                    # what Adam consumed
    if rank == 0:
        torch.save({"params": eng.params.clone(), "losses": losses, "grads": grads}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_big_batch(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    from helpers import OracleEngine
    from oracle import layout_spec as O
    from vlg.spec import LayoutConfig
    cfg = LayoutConfig(**CFG)
    eng = OracleEngine(LayoutConfig(**dict(CFG, B=CFG["B"] * world)), seed=1024)
    from vlg.spec import ADAM_LR
    signal = torch.ones(eng.n_params, dtype=torch.bool)
    for step in range(2):
        full = O.synthetic_batch(cfg.B * world, cfg.T, cfg.N, seed=50 + step)
        loss = float(eng.train_step(full)[0])
        assert abs(got["losses"][step] - loss) < 1e-4 * abs(loss)
        # mean of the two ranks' gradients == gradient of the 2x batch
        g, want = got["grads"][step], eng.grads
        assert float((g - want).abs().max()) <= 1e-5 * float(want.abs().max())
        signal &= want.abs() > 1e-3 * float(want.abs().max())
    # elements with a real gradient end up identical; zero-gradient ones (e.g. the key bias, which softmax
    # ignores) get +-lr of rounding noise from Adam on both sides and are only bounded
    assert int(signal.sum()) > 0.3 * eng.n_params
    assert torch.allclose(got["params"][signal], eng.params[signal], rtol=1e-5, atol=1e-6)
    assert float((got["params"] - eng.params).abs().max()) <= 2 * 2 * ADAM_LR * 1.01


def test_bucket_ranges_tile_the_flat_buffer():
    from vlg.dp import GradReducer, bucket_ranges
    from vlg.spec import LayoutConfig, param_layout
    for kw in (CFG, dict(B=1, T=16, N=4, d=256, n_layers=4)):
        cfg = LayoutConfig(**kw)
        layout, n = param_layout(cfg)
        b = bucket_ranges(layout, n, cfg.n_layers)
        assert [t for t, _, _ in b] == ["head"] + ["l%d" % l for l in reversed(range(cfg.n_layers))] + ["embed"]
        assert b[0][2] == n + 4                                   # loss scalars ride in the first bucket
        GradReducer(torch.zeros(n + 4), b)                        # validates: contiguous, no gaps, full cover
        with pytest.raises(ValueError):
            GradReducer(torch.zeros(n + 4), b[:-1])


def _trainer_worker(rank, world, port, tmp):
    _setup(rank, world, port)
    os.makedirs(os.path.join(tmp, "src%d" % rank), exist_ok=True)
    os.chdir(os.path.join(tmp, "src%d" % rank))
    import random
    random.seed(1024)                                              # main.worker seeds `random` (main.py:57)
    from helpers import oracle_factory, reference_args
    from trainer import Trainer
    args = reference_args(os.path.join(tmp, "exp%d" % rank), rank=rank, gpus=world, batch_size=4, epochs=1,
                          print_freq=1, n_frames=4, n_slots=8, d_model=64, n_layers=1, train_clips=16, val_clips=8)
    tr = Trainer(args, engine_factory=oracle_factory)
    assert tr.cfg.B == 2 and tr.reducer is not None                # per-GPU batch = batch_size // gpus (trainer.py:148)
    t = [torch.tensor([float(rank + 1)])]
    tr.sync(t)                                                     # mean of 1, 2
    assert abs(float(t[0]) - 1.5) < 1e-6
    tr.set_epoch(0)
    tr.train()
    m = tr.validate()
    torch.save({"params": tr.engine.params.clone(), "val": m["loss"]}, os.path.join(tmp, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_two_ranks_stay_in_lockstep(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_trainer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(str(tmp_path / "r0.pt"), weights_only=True)
    b = torch.load(str(tmp_path / "r1.pt"), weights_only=True)
    assert torch.equal(a["params"], b["params"])                   # replicas identical without a broadcast
    assert abs(a["val"] - b["val"]) < 1e-6                         # validate() returns the GLOBAL mean on every rank
