#!/usr/bin/env python
"""Throughput of the layout-token training step on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + fused losses + backward + (bucketed RCCL gradient all-reduce) + Adam over one
batch of synthetic clips already resident in HBM.  Workload at every N: (B,T,N_slots)=(32,16,64),
d=256 per GPU (weak scaling: per-GPU clips fixed, global batch = 32 * N).  One JSON line on rank 0.

Objects on the line besides the contract's fields:
  roofline        the dominant kernel (largest summed device time of the step: gemm_pair_kernel - since round 3 a projection's
                  weight gradient and data gradient run as ONE launch; with VLG_GEMM_PAIR=0 | 1 the weight-gradient GEMM alone):
                  algorithmic FLOP per launch / its average launch duration, from HIP events recorded on the launch stream INSIDE
                  the timed steps (every 4th step), vs the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md);
                  `share_of_step` = launches per step x that duration / ms_per_step; `traffic` = HBM bytes per launch from
                  the committed PMC pass (profiles/pmc_traffic.json) IF that file was taken from these kernel sources
                  (source hash), else null; `step` = the whole step against the same peak
  roofline_hbm    the bandwidth-bound kernels north_star names (embedding, layer-norm, attention, losses, Adam): algorithmic
                  bytes / event time vs 8 TB/s, from an untimed extra pass with every launch bracketed
  cpu_baseline    the CPU oracle's identical step (torch-CPU, host threads stated) on a bounded sample of the SAME batch
  bf16_projections / f32x3_projections   the two other projection modes (informational; `value` stays native fp32)
  strong_scaling_shard   the per-GPU share of the reference's GLOBAL batch semantics (src/trainer.py:148: 32 // 8 = 4 clips)
  comm            (N > 1, or the forced world-1 leg) backend, world size as torch.distributed reports it, the device index of
                  every rank (all-gathered), bytes all-reduced per step, and exposed_comm_ms = ms/step with the bucketed
                  reducer - ms/step of the same ranks with reducer=None (both MAX over ranks)
  rccl_world1     (N = 1) the same step with a real RCCL communicator of world size 1 and the six bucket all-reduces issued
  reference_step  the step that EXISTS in the reference (src/trainer.py:193-258: CoordGridNet + HED x2 + L1/GD/SSIM/VGG/CE +
                  Adam at 256x256), b = 4 and b = 32: ms/step, samples/s, algorithmic TFLOP/s, the event-timed roofline of its
                  dominant convolution shape, and a bounded CPU baseline of its restatement (oracle/image_step_spec.py)

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF (the reference's launcher is one
command too: src/main.py:183-185 mp.spawn): the parent - before any HIP call - runs `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits with the child's status.
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-layout-generation_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md "HBM3E peak BW" (spec)
DOMINANT = "gemm_wgrad"          # largest share of device time (profiles/r0*_kernel_stats.csv); since round 3 the weight gradients run
                                 # paired with their data gradients (family "gemm_pair", 46 % of device time): the fallback below
GEMM_FAMILIES = ("gemm_fwd", "gemm_dgrad", "gemm_wgrad", "gemm_head", "gemm_pair")
KERNEL_OF_FAMILY = {             # rocprofv3 kernel names (profiles/) for each timed family
    "gemm_fwd": "gemm_f32_kernel<128,128,32|16,true,true,*,false,false>",
    "gemm_dgrad": "gemm_f32_kernel<128,128,32|16,true,false,*,false,false>",
    "gemm_wgrad": "gemm_f32_kernel<128,128,32,false,false,0,true,false>",
    "gemm_head": "gemm_f32_kernel<128,32,32,..>/<32,128,32,..>",
    "gemm_pair": "gemm_pair_kernel<128|64, epilogue> (data gradient + weight gradient of one projection in ONE launch)",
    "embed_fwd": "embed_fwd_kernel", "embed_bwd": "embed_bwd_kernel", "ln_fwd": "ln_fwd_kernel", "ln_bwd": "ln_bwd_kernel",
    "attn_fwd": "attn16_fwd_kernel", "attn_bwd": "attn16_bwd_kernel", "loss": "layout_loss_kernel", "adam": "adam_kernel",
}
KERNEL_OF_FAMILY_X3 = {          # --dtype f32x3 (csrc/gemm_split.hip)
    "gemm_wgrad": "gemm_split_kernel<128,128,false,false,0,true>",
}
KERNEL_OF_FAMILY_BF16 = {        # --dtype bf16: bf16 LDS tiles, 64-deep, IO = storage bits (csrc/gemm_bf16.hip)
    "gemm_wgrad": "gemm_bf16_kernel<128,128,false,false,0,true,IO>",
    "gemm_pair": "gemm16_pair_kernel<epilogue, dY bf16> (data gradient + weight gradient of one projection in ONE launch)",
}
TRAFFIC_SOURCES = ("gemm.hip", "gemm_tile.h", "common.h")


def kernel_source_hash() -> str:
    """sha256 (16 hex) of the sources the dominant kernel is built from: ties profiles/pmc_traffic.json to a build."""
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, "video-layout-generation_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_traffic(family: str):
    """HBM bytes per launch of `family` from the committed PMC pass, or None when that pass predates the kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("source_sha16") != kernel_source_hash():
        return None
    return rec.get(family)


def cpu_baseline(cfg, steps: int = 3):
    """Times the CPU oracle (the only CPU implementation of this step: the reference has none) on the SAME batch shape
    (B clips per step), a bounded number of steps."""
    from oracle import layout_spec as O
    from vlg.spec import param_shapes
    # the GPU box gives a one-GPU job a 16-core share; more threads than that only oversubscribes
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    p = O.init_params(param_shapes(cfg), seed=1024)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    batch = O.synthetic_batch(cfg.B, cfg.T, cfg.N, seed=1024)

    def step(i):
        _, g = O.loss_and_grads(p, batch, cfg.n_layers)
        for k in p:
            O.adam_step(p[k], g[k], m[k], v[k], i)

    step(1)                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < steps:
        step(n + 2)
        n += 1
        if time.perf_counter() - t0 > 40.0:
            break
    el = time.perf_counter() - t0
    return {"value": round(cfg.B * n / el, 3), "unit": "clips/s", "cores": threads, "kind": "port", "clips_per_step": cfg.B,
            "sample": "%d steps of %d clips (T=%d,N=%d,d=%d,L=%d), torch-CPU oracle fwd+bwd+Adam, %.1f s"
                      % (n, cfg.B, cfg.T, cfg.N, cfg.d, cfg.n_layers, el)}


class _StdoutToStderr:
    """RCCL prints a version banner on fd 1 when its first communicator comes up; the contract is ONE JSON line on
    stdout, so the communicator is brought up with fd 1 pointing at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> "int":
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as children of THIS process, which has
    made no HIP call (importing torch does not initialise the device), relay rank 0's JSON line, return the children's status.
    Never an exec: a process is started, not replaced."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           "--", os.path.abspath(__file__)] + sys.argv[1:]      # "--": the launcher's argparse would otherwise prefix-match OUR flags (--d ...)
    env = dict(os.environ, VLG_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return r.returncode if (r.returncode != 0 or lines) else 1


def rccl_world1_leg(cfg, dev, batch, steps: int):
    """N = 1 only, informational: the SAME step with a real RCCL communicator (world size 1) and the six bucket all-reduces
    + split Adam really issued (GradReducer(always_communicate=True)) - the data-parallel code path on the one GPU there is."""
    import torch.distributed as dist
    from vlg.dp import GradReducer, bucket_ranges
    from vlg.engine import LayoutEngine
    from vlg.spec import SEED
    try:
        with _StdoutToStderr():
            dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % free_port(), world_size=1, rank=0, device_id=dev)
            dist.all_reduce(torch.zeros(4, device=dev))
            torch.cuda.synchronize()
        e5 = LayoutEngine(cfg, dev, seed=SEED)
        red = GradReducer(e5.grads_ext, bucket_ranges(e5.layout, e5.n_params, cfg.n_layers), always_communicate=True)
        for _ in range(3):
            e5.train_step(batch, red)
        with _StdoutToStderr():
            with_comm = timed_steps(lambda: e5.train_step(batch, red), steps)
        without = timed_steps(lambda: e5.train_step(batch, None), steps)
        out = {"backend": dist.get_backend() + " (RCCL)", "world": dist.get_world_size(), "ms_per_step": round(1e3 * with_comm, 4),
               "ms_per_step_no_reducer": round(1e3 * without, 4), "exposed_comm_ms": round(1e3 * (with_comm - without), 4),
               "bytes_allreduced_per_step": 4 * int(e5.grads_ext.numel()), "buckets": len(red.buckets),
               "final_loss": round(float(e5.loss_out[0]), 5),
               "note": "one process, RCCL communicator of world size 1: six asynchronous bucket all-reduces + split Adam issued every step"}
        del e5
        dist.destroy_process_group()
        return out
    except Exception as ex:                              # informational leg: never costs the line
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


REFERENCE_GFLOP_PER_SAMPLE = 188.7 + 2 * 40.1 + 3 * 46.1      # SURVEY.md section 6 / 8d Spec R: GridNet fwd+bwd, HED x2, VGG19[:27] x2 fwd + dgrad


class _ConvTimer:
    """hip.tracer hook: brackets every vlg_conv3x3_{fwd,dgrad,wgrad} launch with events on the launch stream, grouped by
    (entry point, padded rows, input channels, output channels)."""
    ARGS = {"vlg_conv3x3_fwd": (8, 9, 10), "vlg_conv3x3_dgrad": (9, 10, 11), "vlg_conv3x3_wgrad": (7, 8, 9)}    # rows, cin_p, cout(_p)

    def __init__(self):
        self.rec = {}

    def __call__(self, name, args):
        if name not in self.ARGS:
            return None
        key = (name,) + tuple(int(args[i]) for i in self.ARGS[name])
        return self._section(key)

    @contextlib.contextmanager
    def _section(self, key):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        yield
        e.record()
        self.rec.setdefault(key, []).append((s, e))


def reference_step_leg(dev, cpu: bool = True):
    """The step that exists in the reference (src/trainer.py:193-258), driver-timed: CoordGridNet + frozen HED x2 + 40 L1 +
    20 (VGG + GradientLoss + SSIM) + 10 CE + Adam on 256x256 frames, b = 4 and b = 32 (the reference's default batch), with
    torch-default random weights (the trained HED / VGG weights are not in the reference repository)."""
    from vlg import hip
    from vlg.image_engine import ImageEngine, random_state, synthetic_frames
    H = W = 256
    out = {"gflop_per_sample": round(REFERENCE_GFLOP_PER_SAMPLE, 1), "frames": "%dx%d" % (H, W),
           "gflop_note": "GridNet fwd+bwd 188.7 + HED 2 x 40.1 (the third HED call of trainer.py:214-216 only feeds tensorboard: not made) + "
                         "VGG19[:27] 2 x 46.1 fwd + 46.1 input gradient (SURVEY.md section 6)",
           "step": "src/trainer.py:193-258 with the Appendix-A repairs: prep -> CoordGridNet -> 40 L1 + 20 (VGG + GD + SSIM) + 10 CE "
                   "-> backward -> Adam; HED fused map x2 under no-grad"}
    for b in (4, 32):
        eng = ImageEngine(b, H, W, dev, arch="CoordGridNet", with_hed=True, with_vgg=True)
        eng.load_state_dict(random_state(eng.net.reference_shapes(), 1024))
        eng.hed.load_state_dict(random_state(eng.hed.reference_shapes(), 1041))
        eng.vgg.load_state_dict(random_state(eng.vgg.reference_shapes(), 1042))
        batch = {k: v.to(dev) for k, v in synthetic_frames(b, H, W, seed=1024).items() if k not in ("e1", "e2")}
        for _ in range(3):
            eng.train_step(batch)
        dt = timed_steps(lambda: eng.train_step(batch), 10)
        rec = {"ms_per_step": round(1e3 * dt, 3), "samples_per_s": round(b / dt, 1),
               "algorithmic_tflops": round(b * REFERENCE_GFLOP_PER_SAMPLE / dt / 1e3, 1),
               "frac_of_fp32_mfma_peak": round(b * REFERENCE_GFLOP_PER_SAMPLE / dt / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
               "loss": round(float(eng.total()), 5)}
        ct = _ConvTimer()                                   # untimed pass: every convolution launch bracketed
        hip.tracer = ct
        try:
            eng.train_step(batch)
            eng.train_step(batch)
            torch.cuda.synchronize()
        finally:
            hip.tracer = None
        # per shape group: (summed ms over the two bracketed steps, launches per step, algorithmic GFLOP per launch).  `rows`
        # counts the halo-padded pixels of a level; the algorithmic work is over the b * h * w real ones (rows = b (h + 2)(w + 2),
        # square levels) - exact for the stride-1 convolutions, which every shape that matters here is
        groups = []
        for (name, rows, cin, cout), ev in ct.rec.items():
            ms = sum(s.elapsed_time(e) for s, e in ev)
            side = int(round((rows / b) ** 0.5)) - 2
            groups.append({"entry": name, "cin": cin, "cout": cout, "side": side, "ms": ms, "launches": len(ev) // 2,
                           "gflop": 2.0 * b * side * side * 9 * cin * cout / 1e9})
        rec["conv_ms_per_step"] = round(sum(g["ms"] for g in groups) / 2, 3)
        # the dominant kernel of the step's rocprof summary (profiles/r0*_reference_step_kernel_stats.csv) is
        # conv_gemm_kernel<0, 128, 128, 32>: forward launches on 128 x 128 tiles = the frozen trunks' layers with >= 128 output
        # channels and >= 64 input channels (channel counts there are multiples of 32: padded = real)
        fam = [g for g in groups if g["entry"] == "vlg_conv3x3_fwd" and g["cout"] >= 128 and g["cin"] >= 64]
        if fam:
            ms, n = sum(g["ms"] for g in fam), sum(2 * g["launches"] for g in fam)
            gf = sum(g["gflop"] * 2 * g["launches"] for g in fam)
            rec["roofline"] = {"bound": "mfma", "kernel": "conv_gemm_kernel<0, 128, 128, 32> (forward, >= 128 output channels: VGG19 / HED trunks)",
                               "launches_per_step": n // 2, "avg_launch_us": round(1e3 * ms / n, 2), "gflop_per_launch": round(gf / n, 3),
                               "achieved": round(gf / ms, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(gf / ms / PEAK_F32_MFMA_TFLOPS, 4), "share_of_step": round(ms / 2 / (1e3 * dt), 4)}
        top = sorted(groups, key=lambda g: -g["ms"])[:4]
        rec["top_conv_shapes"] = [{"entry": g["entry"], "shape": "%d->%d @ %dx%dx%d" % (g["cin"], g["cout"], b, g["side"], g["side"]),
                                   "launches_per_step": g["launches"], "avg_launch_us": round(1e3 * g["ms"] / (2 * g["launches"]), 2),
                                   "tflops": round(g["gflop"] * 2 * g["launches"] / g["ms"], 1)} for g in top]
        out["b%d" % b] = rec
        del eng, batch
        torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = reference_cpu_baseline()
    return out


def reference_cpu_baseline():
    """The torch-CPU restatement of the same step (oracle/image_step_spec.py + hned_spec + vgg_spec; the reference's own
    trainer cannot be imported: SURVEY.md section 8c), one sample per step, bounded."""
    from oracle import gridnet_spec as G, hned_spec as HS, image_step_spec as S, vgg_spec as V
    from vlg.image_engine import synthetic_frames
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    p = G.test_params(G.param_shapes(10, coord=True), seed=0)
    hp, vp = HS.test_params(0), V.test_params(0)
    batch = synthetic_frames(1, 256, 256, seed=1024)

    def step():
        with torch.no_grad():
            batch["e1"] = HS.forward(hp, batch["frame1"])[5]
            batch["e2"] = HS.forward(hp, batch["frame2"])[5]
        S.loss_and_grads(p, batch, True, vgg_params=vp)

    step()
    t0 = time.perf_counter()
    n = 0
    while n < 3 and time.perf_counter() - t0 < 15.0:
        step()
        n += 1
    el = time.perf_counter() - t0
    return {"value": round(n / el, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d steps of 1 sample (256x256), torch-CPU restatement fwd+bwd without Adam, %.1f s" % (n, el)}


def timed_steps(step_fn, n: int) -> float:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step_fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--N", type=int, default=64)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--dtype", choices=["f32", "f32x3", "bf16"], default="f32",
                    help="f32 (default): native fp32 MFMA projections, parity 1e-4; f32x3: fp32 tensors, projections on the "
                         "bf16 matrix cores from exact 3-way operand splits (same 1e-4 parity); bf16: BASELINE configs[2] mode")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from one captured hipGraph (single GPU; same kernels, one host call per step)")
    ap.add_argument("--settle", type=float, default=1.0, help="seconds of untimed steps before the warm-up steps (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs (other modes, B=4 shard)")
    ap.add_argument("--no-reference-step", action="store_true", help="skip the reference's own pixel step (reference_step object)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if os.environ.get("VLG_BENCH_ONE_DEVICE", "0") == "1":
        # rehearsal with N processes on ONE device: each HIP process opens 4 hardware queues by default (+ gloo's copy
        # streams); beyond the device's hardware queue slots the driver time-slices the run list and a 6 ms step takes 69 s
        # (measured at 4 ranks: 69 235 ms / step with the default, 36 ms with 2 queues per process - DESIGN.md (e),
        # profiles/r02_dp4_*.log).
        # Read by the HIP runtime when it starts, so it is set before the first HIP call.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path for the product")
    if os.environ.get("VLG_BENCH_ONE_DEVICE", "0") == "1":
        local = 0                                    # rehearsal: every rank on the one GPU of a test box (with gloo, below)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = world > 1 or "RANK" in os.environ          # under torch.distributed.run even at N=1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("VLG_BENCH_BACKEND", "nccl")                       # nccl == RCCL on ROCm
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group(backend="nccl", world_size=world, rank=rank, device_id=dev)
            else:                                    # rehearsal of the multi-rank control flow on one GPU (RCCL refuses that)
                dist.init_process_group(backend=backend, world_size=world, rank=rank)
            # communicator set-up (RCCL builds its rings on the first collective) is not part of a step: do it now
            dist.all_reduce(torch.zeros(4, device=dev))
            torch.cuda.synchronize()

    from vlg.data import synthetic_clips, to_device
    from vlg.dp import GradReducer, bucket_ranges
    from vlg.engine import KernelTimer, LayoutEngine
    from vlg.spec import LayoutConfig, SEED, step_flops

    cfg = LayoutConfig(B=args.B, T=args.T, N=args.N, d=args.d, n_layers=args.layers)
    precision = {"f32": "fp32", "f32x3": "fp32x3", "bf16": "bf16"}[args.dtype]
    eng = LayoutEngine(cfg, dev, seed=SEED, precision=precision)   # same seed on every rank (main.py:57-60)
    batch = to_device(synthetic_clips(cfg.B, cfg.T, cfg.N, seed=SEED + rank), dev)   # each rank its own clips
    reducer = None
    force = os.environ.get("VLG_FORCE_COMM", "0") == "1"      # 1-GPU rehearsal of the RCCL path (torchrun, world size 1)
    if world > 1 or (distributed and force):
        reducer = GradReducer(eng.grads_ext, bucket_ranges(eng.layout, eng.n_params, cfg.n_layers),
                              always_communicate=force)
    trace = os.environ.get("VLG_BENCH_TRACE", "0") == "1"     # host timestamps per step (rehearsal diagnostics)

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    step_fn = lambda: eng.train_step(batch, reducer)
    if args.graph:
        if reducer is not None:
            raise SystemExit("--graph captures the single-process step; the data-parallel hooks are not captured")
        captured = eng.capture_train_step(batch)
        step_fn = lambda: captured(batch)
    # clock settle (untimed, reported as `settle_s`): the shader clock and the HBM power state ramp for the first second of a
    # process (1.96 GHz on the first launches, 2.35-2.39 GHz after a second) - with --warmup 5 the K timed
    # steps would otherwise start 30 ms after the first launch and measure the ramp, not the step
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        eng.forward_backward(batch)                # rank-local: no collective, no parameter update (replicas stay identical)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step_fn()
    if args.warmup == 0:
        # code objects are loaded on a kernel's first launch (tens of ms): with --warmup 0 run one untimed step anyway, so
        # the K timed steps measure the step and not the loader (reported as "primed": true)
        step_fn()
    # Inside the timed region only the dominant kernel (the weight-gradient GEMM instantiation: top of
    # every rocprof summary in profiles/) is bracketed with events, and only on every 4th step (16 launches each):
    # an event pair around each of its launches costs ~2 % of `value`, around all ~70 GEMM launches far more.
    # (the backward pairs a projection's data and weight gradient in one launch, family "gemm_pair"; VLG_GEMM_PAIR=0 | 1 separates them)
    ktimer = None if (args.no_kernel_timing or args.graph) else KernelTimer(only=(DOMINANT, "gemm_pair"))   # events are not part of a replayed graph
    sampled_steps = 0
    sync_all()
    t0 = time.perf_counter()
    for it in range(args.steps):
        sample = ktimer is not None and it % 4 == 0
        eng.timer = ktimer if sample else None
        sampled_steps += int(sample)
        if trace and rank == 0:
            h0 = time.perf_counter()
        step_fn()
        if trace and rank == 0:
            h1 = time.perf_counter()
            torch.cuda.synchronize()
            print("trace step %d: host enqueue %.2f ms, device drained after %.2f ms" % (it, 1e3 * (h1 - h0), 1e3 * (time.perf_counter() - h0)),
                  file=sys.stderr, flush=True)
    eng.timer = ktimer
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = [float(x) / world for x in eng.loss_out.cpu()]     # summed over ranks inside the first bucket
    comm = None
    if reducer is not None:
        # communication evidence: who took part, what moved, and what it cost on top of the same ranks' compute
        eng.timer = None
        for _ in range(2):
            eng.train_step(batch, None)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step(batch, None)
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_no_comm = 1e3 * float(t.item()) / args.steps
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device": torch.cuda.current_device(), "name": prop.name,
                "pci": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0), getattr(prop, "pci_device_id", 0))}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        comm = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""),
                "world": dist.get_world_size(), "ranks": seen,
                "bytes_allreduced_per_step": 4 * int(eng.grads_ext.numel()), "buckets": len(reducer.buckets),
                "ms_per_step_no_reducer": round(ms_no_comm, 4),
                "exposed_comm_ms": round(1e3 * elapsed / args.steps - ms_no_comm, 4),
                "note": "bucketed async all-reduce (SUM) of the flat gradient + 4 loss floats, 1/world folded into Adam; "
                        "exposed = with reducer - reducer=None on the same ranks, both MAX over ranks"}
        eng.timer = ktimer

    if rank == 0:
        clips = world * cfg.B * args.steps
        fl = step_flops(cfg)
        ms_per_step = 1e3 * elapsed / args.steps
        line = {
            "metric": "training clips/sec at (B,T,N)=(32,16,64) d=256; 1/2/4/8-GPU scaling",
            "value": round(clips / elapsed, 2), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "primed": args.warmup == 0, "settle_s": args.settle, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "hipgraph": bool(args.graph),
            "config": {"workload": "layout-token training step, (B,T,N)=(%d,%d,%d) clips per GPU, d=%d"
                                   % (cfg.B, cfg.T, cfg.N, cfg.d),
                       "global_batch": world * cfg.B, "parallelism": "dp%d" % world,
                       "self_oracle_hyperparameters": cfg.describe(),
                       "step_gflop": round(fl["fwd_bwd"] / 1e9, 1),
                       "step_tflops": round(fl["fwd_bwd"] * args.steps / elapsed / 1e12, 2),
                       "final_loss": [round(x, 5) for x in loss]},
        }
        line["comm"] = comm
        roof, roof_hbm = None, None
        if eng.timer is not None:
            summ = eng.timer.summary()
            dom = DOMINANT if DOMINANT in summ else "gemm_pair"
            s = summ[dom]
            # untimed diagnostic pass: EVERY launch bracketed (rank 0 only, so WITHOUT the reducer: a collective here
            # would have no partner on the other ranks)
            eng.timer = KernelTimer()
            for _ in range(2):
                eng.train_step(batch, None)
            torch.cuda.synchronize()
            allf = eng.timer.summary()
            eng.timer = None
            achieved = s["flops_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
            traffic = committed_traffic(dom) if args.dtype == "f32" else None
            bound, peak, unit = "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
            names = KERNEL_OF_FAMILY
            if args.dtype == "f32x3":
                # six bf16 MFMAs stand for one fp32 product block: price the ALGORITHMIC flops against the bf16 peak / 6
                peak, names = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1), dict(KERNEL_OF_FAMILY, **KERNEL_OF_FAMILY_X3)
            if args.dtype == "bf16":
                # with 16x faster MFMAs the same kernel is bound by moving its operands: price it against HBM
                # (algorithmic bytes: both operands once + the slabs written)
                bound, peak, unit = "hbm", PEAK_HBM_GBS, "GB/s"
                achieved = s["bytes_per_launch"] / (s["avg_ms"] * 1e-3) / 1e9
                names = dict(KERNEL_OF_FAMILY, **KERNEL_OF_FAMILY_BF16)
            launches_per_step = s["launches"] / max(sampled_steps, 1)
            step_tflops = fl["fwd_bwd"] / (ms_per_step * 1e-3) / 1e12
            roof = {"bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(s["bytes_per_launch"]),
                    "kernel": names[dom], "launches": s["launches"], "launches_per_step": launches_per_step,
                    "avg_launch_us": round(1e3 * s["avg_ms"], 2),
                    "gflop_per_launch": round(s["flops_per_launch"] / 1e9, 3),
                    "share_of_step": round(launches_per_step * s["avg_ms"] / ms_per_step, 4),
                    "kernel_source_sha16": kernel_source_hash(),
                    "step": {"achieved": round(step_tflops, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(step_tflops / PEAK_F32_MFMA_TFLOPS, 4),
                             "note": "all %.1f GFLOP of the step / ms_per_step vs the native fp32 MFMA peak" % (fl["fwd_bwd"] / 1e9)},
                    "families_untimed_pass": {k: {"avg_us": round(1e3 * v["avg_ms"], 2), "launches": v["launches"],
                                                  "ms_per_step": round(v["total_ms"] / 2, 3),
                                                  "tflops": round(v["flops_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12, 2)}
                                              for k, v in allf.items() if k in GEMM_FAMILIES}}
            attn = {16: "attn16", 32: "attn32"}.get(args.T, "attn")      # csrc/attention.hip: MFMA score tiles at T = 16 / 32
            hbm_names = dict(KERNEL_OF_FAMILY, attn_fwd=attn + "_fwd_kernel", attn_bwd=attn + "_bwd_kernel")
            roof_hbm = [{"kernel": hbm_names.get(k, k), "family": k, "bound": "hbm",
                         "achieved": round(v["bytes_per_launch"] / (v["avg_ms"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(v["bytes_per_launch"] / (v["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                         "algorithmic_bytes_per_launch": int(v["bytes_per_launch"]), "avg_launch_us": round(1e3 * v["avg_ms"], 2),
                         "launches_per_step": v["launches"] // 2, "ms_per_step": round(v["total_ms"] / 2, 4)}
                        for k, v in allf.items() if k not in GEMM_FAMILIES]
        line["roofline"] = roof
        line["roofline_hbm"] = roof_hbm
        if args.dtype == "f32" and world == 1 and not args.no_extras:
            # informational: the same step in the two other projection modes; `value` stays native fp32
            notes = {
                "bf16": ("bf16_projections",
                         "BASELINE.json configs[2]: v_mfma_f32_32x32x16_bf16 projections (fp32 accumulate) with the projection-side "
                         "activations and their gradients stored as bf16 in HBM and a bf16 weight shadow; residual stream, "
                         "statistics, softmax, losses, weight gradients, Adam fp32; loss within 2e-2 of fp32 "
                         "(tests/test_hip_step.py)"),
                "fp32x3": ("f32x3_projections",
                           "fp32 tensors, fp32-grade projections on the bf16 matrix cores: operands split exactly into three bf16 "
                           "terms, six bf16 MFMAs per product block, fp32 accumulate (csrc/gemm_split.hip); passes the same 1e-4 "
                           "parity tests as the native fp32 path, error vs fp64 within 4x of it (tests/test_hip_ops.py)"),
            }
            for prec, (key, note) in notes.items():
                e2 = LayoutEngine(cfg, dev, seed=SEED, precision=prec)
                for _ in range(3):
                    e2.train_step(batch)
                dt = timed_steps(lambda: e2.train_step(batch), 10)
                line[key] = {"value": round(cfg.B / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt, 4),
                             "note": note, "final_loss": round(float(e2.loss_out[0]), 5)}
                if prec == "bf16":
                    # BASELINE configs[2] asks for HBM GB/s + MFMA utilisation of this mode: the dominant kernel's
                    # algorithmic bytes / event time, and the step's projection FLOPs against the dense bf16 MFMA peak
                    e2.timer = KernelTimer()
                    for _ in range(2):
                        e2.train_step(batch)
                    torch.cuda.synchronize()
                    fam = e2.timer.summary()
                    e2.timer = None
                    bdom = DOMINANT if DOMINANT in fam else "gemm_pair"          # (paired launches: gemm16_pair_kernel)
                    w = fam[bdom]
                    line[key]["hbm"] = {"kernel": KERNEL_OF_FAMILY_BF16[DOMINANT] if bdom == DOMINANT else "gemm16_pair_kernel<epilogue, dY bf16> (data + weight gradient of one projection)",
                                        "avg_launch_us": round(1e3 * w["avg_ms"], 2),
                                        "achieved_gbs": round(w["bytes_per_launch"] / (w["avg_ms"] * 1e-3) / 1e9, 1),
                                        "frac_of_8000": round(w["bytes_per_launch"] / (w["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                    line[key]["mfma_util"] = {"step_tflops": round(fl["fwd_bwd"] / dt / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                              "frac": round(fl["fwd_bwd"] / dt / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                                              "kernel_tflops": round(w["flops_per_launch"] / (w["avg_ms"] * 1e-3) / 1e12, 1)}
                    line[key]["families"] = {k: {"avg_us": round(1e3 * v["avg_ms"], 2), "gbs": round(v["bytes_per_launch"] / (v["avg_ms"] * 1e-3) / 1e9, 1)}
                                             for k, v in fam.items()}
                    # the mode is HBM-bound: the WHOLE step against the HBM roofline = sum of every launch's algorithmic bytes
                    # (operands once + outputs once, per launch; the slab reductions' reads are not counted) / step time
                    step_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in fam.values()) / 2.0
                    line[key]["roofline_step"] = {"bound": "hbm", "algorithmic_bytes_per_step": int(step_bytes),
                                                  "achieved": round(step_bytes / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                  "frac": round(step_bytes / dt / 1e9 / PEAK_HBM_GBS, 4)}
                del e2
            # informational: the same native fp32 step with the weight gradients on a second HIP stream (LayoutEngine option
            # VLG_OVERLAP_WGRAD=1): launch boundaries and the bandwidth-bound kernels of the chain then overlap with them.  NOT
            # the default and not `value`: two kernels sharing the chip make a kernel's own duration (the roofline object)
            # meaningless
            os.environ["VLG_OVERLAP_WGRAD"] = "1"
            e3 = LayoutEngine(cfg, dev, seed=SEED)
            os.environ.pop("VLG_OVERLAP_WGRAD")
            for _ in range(3):
                e3.train_step(batch)
            dt = timed_steps(lambda: e3.train_step(batch), 10)
            line["two_stream_backward"] = {"value": round(cfg.B / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt, 4),
                                           "note": "native fp32, weight gradients (+ slab reductions) on a second stream beside the "
                                                   "data-gradient chain; same results bit for bit (tests/test_hip_step.py)",
                                           "final_loss": round(float(e3.loss_out[0]), 5)}
            del e3
            # the reference's batch semantics are GLOBAL (src/main.py:105-106, src/trainer.py:148: per-GPU = 32 // gpus):
            # the 8-GPU share of the metric batch is 4 clips per GPU - measurable on one GPU, informational
            c4 = LayoutConfig(B=max(cfg.B // 8, 1), T=cfg.T, N=cfg.N, d=cfg.d, n_layers=cfg.n_layers)
            e4 = LayoutEngine(c4, dev, seed=SEED)
            b4 = to_device(synthetic_clips(c4.B, c4.T, c4.N, seed=SEED), dev)
            for _ in range(5):
                e4.train_step(b4)
            dt = timed_steps(lambda: e4.train_step(b4), 30)
            f4 = step_flops(c4)
            run4 = e4.capture_train_step(b4)                       # the same step replayed from one hipGraph (one host call)
            for _ in range(5):
                run4(b4)
            dtg = timed_steps(lambda: run4(b4), 30)
            # the shard is what the step looks like when launches are short (71 launches of 13-15 us): replayed from the captured
            # hipGraph it needs ONE host call per step, so `ms_per_step` is the replay; the eager figure (a Python / ctypes call per
            # launch: as fast on an idle host, 2.6x slower on a busy one - seen once on a shared box) is reported beside it
            line["strong_scaling_shard"] = {
                "clips_per_gpu": c4.B, "ms_per_step": round(1e3 * dtg, 4), "clips_per_s_one_gpu": round(c4.B / dtg, 1),
                "step_tflops": round(f4["fwd_bwd"] / dtg / 1e12, 2),
                "frac": round(f4["fwd_bwd"] / dtg / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "peak": PEAK_F32_MFMA_TFLOPS,
                "hipgraph": True, "eager_ms_per_step": round(1e3 * dt, 4),
                "kernels": "64x64-tile fp32 MFMA GEMMs (launches whose 128x128 tiles leave CUs empty), data + weight gradient of a "
                           "projection in one launch (gemm_pair_kernel)",
                "note": "one rank's share if the GLOBAL batch stayed 32 on 8 GPUs (reference semantics); no communication here: "
                        "the 12.7 MB gradient all-reduce would have to hide inside this step time"}
            del e4
            # informational, SELF-ORACLE: the same step with the per-clip reading of the temporal encoder (attention = "clip":
            # block-causal attention over all T*N tokens of a clip on fp32 MFMA flash-style kernels, csrc/attention_clip.hip)
            cc = LayoutConfig(B=cfg.B, T=cfg.T, N=cfg.N, d=cfg.d, n_layers=cfg.n_layers, attention="clip")
            e6 = LayoutEngine(cc, dev, seed=SEED, padded_slots=False)
            for _ in range(3):
                e6.train_step(batch)
            dt = timed_steps(lambda: e6.train_step(batch), 10)
            e6.timer = KernelTimer(only=("attn_clip_fwd", "attn_clip_bwd"))
            e6.train_step(batch)
            e6.train_step(batch)
            torch.cuda.synchronize()
            ks = e6.timer.summary()
            e6.timer = None
            f6 = step_flops(cc)
            line["per_clip_attention"] = {
                "value": round(cfg.B / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt, 4),
                "step_gflop": round(f6["fwd_bwd"] / 1e9, 1), "step_tflops": round(f6["fwd_bwd"] / dt / 1e12, 2),
                "frac": round(f6["fwd_bwd"] / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "kernels": {k: {"avg_launch_us": round(1e3 * v["avg_ms"], 2), "gflop_per_launch": round(v["flops_per_launch"] / 1e9, 2),
                                "tflops": round(v["flops_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12, 2),
                                "frac_of_fp32_mfma_peak": round(v["flops_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
                            for k, v in ks.items()},
                "self_oracle_hyperparameters": cc.describe()["attention"], "final_loss": round(float(e6.loss_out[0]), 5),
                "note": "SELF-ORACLE option (VLG_ATTENTION=clip), not the headline: token (t, n) attends to every slot of frames <= t "
                        "of its clip; attn_clip_bwd = two launches (query owner dQ, key owner dK / dV), algorithmic flops = visible "
                        "(query, key) pairs x 4 x 64 x {1 fwd, 2.5 bwd}"}
            del e6
            if not distributed:
                line["rccl_world1"] = rccl_world1_leg(cfg, dev, batch, args.steps)
            if not args.no_reference_step:
                line["reference_step"] = reference_step_leg(dev, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
