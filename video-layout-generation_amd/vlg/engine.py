"""Hand-scheduled training step of the layout-token model on one MI355X.

No autograd, no tracing compiler: forward and backward are explicit sequences of
launches of the gfx950 kernels in libvlg_hip.so on the caller's HIP stream, over
buffers allocated once (parameters, gradients and Adam moments are single flat
fp32 buffers; activations live in a preallocated workspace sized for the largest
batch).  PyTorch supplies device memory and streams only.

Step order mirrors the reference's Trainer.train() body (reference
src/trainer.py:209-258): forward -> weighted loss (40/20/10) -> backward ->
[gradient all-reduce by the caller] -> Adam.  Gradients are overwritten every
step, i.e. the zero_grad() the reference forgot is implied.

Row order of every (M, .) activation is the internal  m = (b*N + n)*T + t.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch

from . import hip
from .hip import (EPI_A_BF16, EPI_ACT_GELU, EPI_B_BF16, EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_NONE, EPI_OUT_BF16, EPI_RESID,
                  call, ptr)
from .spec import (ADAM_BETA1, ADAM_BETA2, ADAM_EPS, ADAM_LR, BOX_DIM, IOU_EPS, LN_EPS, LOSS_W_CE,
                   LOSS_W_REG, LOSS_W_STRUCT, SMOOTH_L1_BETA, LayoutConfig, param_layout)


def init_params(cfg: LayoutConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic CPU initialisation (one generator, tensors in flat-buffer order).

    Embedding tables ~ N(0,1) (nn.Embedding default, reference src/models/simple.py:23),
    projections ~ U(+-1/sqrt(fan_in)) for weight and bias (nn.Linear default),
    layer-norm gain 1 / bias 0.  Every rank uses the same seed, as the reference does
    (src/main.py:57-60), so replicas start identical without a broadcast.
    """
    layout, _ = param_layout(cfg)
    g = torch.Generator().manual_seed(seed)
    fan_in = {name[:-2] + "_b": shape[1] for name, (_, shape) in layout.items() if name.endswith("_w")}
    out: Dict[str, torch.Tensor] = {}
    for name, (_, shape) in layout.items():
        base = name.split(".")[-1]
        if base in ("cls_emb", "time_emb"):
            t = torch.randn(shape, generator=g, dtype=torch.float32)
        elif base.endswith("_g"):
            t = torch.ones(shape)
        elif base.startswith("ln") and base.endswith("_b"):
            t = torch.zeros(shape)
        elif base.endswith("_w"):
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * (1.0 / math.sqrt(shape[1]))
        else:
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * (1.0 / math.sqrt(fan_in[name]))
        out[name] = t
    return out


class LayoutEngine:
    """Owns parameters, optimiser state and workspace; runs forward/backward/Adam."""

    def __init__(self, cfg: LayoutConfig, device: torch.device, seed: int = 1024,
                 lr: float = ADAM_LR, beta1: float = ADAM_BETA1, precision: str = "fp32", padded_slots: bool = True):
        """precision:
        "fp32"      exact-fp32 MFMA projections, every tensor fp32 (parity 1e-4);
        "bf16"      BASELINE.json configs[2]: bf16 MFMA projections (fp32 accumulate) AND the activations that only
                    feed projections / attention (normalised inputs, q k v, attention output, FFN hidden and their
                    gradients) stored as bf16 in HBM - the mode is HBM-bound, so the bytes are what count;
        "bf16_mfma" bf16 MFMA projections with every tensor still fp32 in HBM (operands rounded on their way to LDS);
        "fp32x3"    fp32 tensors and fp32-grade projections on the bf16 matrix cores: every operand split exactly into
                    three bf16 terms, six bf16 MFMAs per product block (csrc/gemm_split.hip).
        The residual stream and its gradient, layer-norm statistics, softmax, losses, weight gradients, Adam and the
        master weights are fp32 in all three."""
        cfg.validate()
        if precision not in ("fp32", "bf16", "bf16_mfma", "fp32x3"):
            raise ValueError("precision must be fp32, fp32x3, bf16 or bf16_mfma")
        if cfg.attention == "clip" and precision == "bf16":
            raise ValueError("attention='clip' runs on fp32 tensors (precision fp32 | fp32x3 | bf16_mfma)")
        # attention = "clip": padded slots (batch['valid'] == 0) must not be attended to; padded_slots = False tells the engine
        # that its batches never hold any (fixed-N feeds), so the kernels skip the per-key validity masks
        self.padded_slots = bool(padded_slots)
        self.precision = precision
        self.gemm_flags = {"fp32": 0, "fp32x3": hip.EPI_SPLIT3}.get(precision, hip.EPI_BF16)
        self.bf16_store = precision == "bf16"
        # optional (native fp32 only): gelu(u) is never stored - the FFN's first projection writes the pre-activation u
        # only and the second projection / its weight gradient apply GELU while staging their operand (VLG_EPI_ACT_GELU)
        self.gelu_on_load = False      # measured slower end to end (GEMM notes): the recomputation is not hidden
        # native fp32 and the bf16 modes: the FFN's first projection stores gelu'(u) in the pre-activation buffer instead of u (VLG_EPI_GELU_GRAD;
        # nothing else reads u) and the second projection's data gradient multiplies by it (VLG_EPI_MUL): ~20 vector
        # instructions per element less in a kernel that pays for each of them in matrix time (csrc/common.h)
        self.gelu_grad_saved = precision in ("fp32", "bf16", "bf16_mfma") and not self.gelu_on_load and os.environ.get("VLG_GELU_GRAD_SAVED", "1") != "0"
        self._epi_ff1 = EPI_BIAS | EPI_GELU | (hip.EPI_GELU_GRAD if self.gelu_grad_saved else 0)
        self._epi_dff2 = hip.EPI_MUL if self.gelu_grad_saved else EPI_DGELU
        self._sfx = "_bf16" if self.bf16_store else ""
        hip.load()                                   # fail loudly before touching the GPU
        if device.type != "cuda":
            raise hip.HipError("LayoutEngine needs a HIP device (got %s); there is no CPU path" % device)
        self.cfg, self.device = cfg, device
        self.lr, self.beta1 = float(lr), float(beta1)
        self.layout, self.n_params = param_layout(cfg)
        f32 = dict(dtype=torch.float32, device=device)
        self.params = torch.zeros(self.n_params, **f32)
        # 4 extra floats after the last parameter hold the loss scalars, so they travel inside the
        # first gradient bucket of the data-parallel all-reduce (vlg/dp.py)
        self.grads_ext = torch.zeros(self.n_params + 4, **f32)
        self.grads = self.grads_ext[:self.n_params]
        self.exp_avg = torch.zeros(self.n_params, **f32)
        self.exp_avg_sq = torch.zeros(self.n_params, **f32)
        self.step_count = 0
        self.adam_state = None                       # device-side {step_size, sqrt_bc2, step} once a step is captured in a hipGraph
        # bf16 mode: a bf16 copy of the weights feeds the projections (the Adam kernel refreshes it with every update)
        self.params_bf16 = torch.zeros(self.n_params, dtype=torch.bfloat16, device=device) if self.bf16_store else None
        self.load_params(init_params(cfg, seed))
        self._alloc_workspace(cfg.tokens)
        self.loss_out = self.grads_ext[self.n_params:]   # {total, smooth_l1, iou, ce}
        self.timer = None                            # optional KernelTimer (bench.py roofline leg)

    # ------------------------------------------------------------------ parameters
    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.layout[name]
        return flat[off:off + math.prod(shape)].view(shape)

    def p(self, name: str) -> torch.Tensor:
        return self.view(self.params, name)

    def pw(self, name: str) -> torch.Tensor:
        """weight operand of a projection: the fp32 master, or its bf16 shadow in the bf16 mode"""
        return self.view(self.params_bf16 if self.bf16_store else self.params, name)

    def _sync_device_step(self) -> None:
        if self.adam_state is not None:
            self.adam_state.view(torch.int32)[2] = int(self.step_count)

    def _refresh_shadow(self) -> None:
        if self.params_bf16 is not None:
            self.params_bf16.copy_(self.params)          # round to nearest even, as the Adam kernel does

    def g(self, name: str) -> torch.Tensor:
        return self.view(self.grads, name)

    def load_params(self, tensors: Dict[str, torch.Tensor]) -> None:
        for name in self.layout:
            self.p(name).copy_(tensors[name].to(torch.float32))
        self._refresh_shadow()

    def named_params(self) -> Dict[str, torch.Tensor]:
        return {n: self.p(n) for n in self.layout}

    def named_grads(self) -> Dict[str, torch.Tensor]:
        return {n: self.g(n) for n in self.layout}

    def state_dict(self) -> Dict[str, object]:
        return {"params": self.params.detach().cpu().clone(), "exp_avg": self.exp_avg.cpu().clone(),
                "exp_avg_sq": self.exp_avg_sq.cpu().clone(), "step": self.step_count,
                "layout": {k: (o, tuple(s)) for k, (o, s) in self.layout.items()}}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        if tuple(sd["params"].shape) != (self.n_params,):
            raise ValueError("checkpoint has %d parameters, model has %d" % (sd["params"].numel(), self.n_params))
        self.params.copy_(sd["params"])
        self._refresh_shadow()
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self._sync_device_step()

    def optimizer_state(self) -> Dict[str, object]:
        """Adam state for the checkpoint's 'optimizer' entry (flat tensors, CPU)."""
        return {"exp_avg": self.exp_avg.cpu().clone(), "exp_avg_sq": self.exp_avg_sq.cpu().clone(),
                "step": int(self.step_count), "lr": self.lr, "beta1": self.beta1}

    def load_optimizer(self, st: Dict[str, object]) -> None:
        if st["exp_avg"].numel() != self.n_params:
            raise ValueError("optimizer state has %d elements, model has %d" % (st["exp_avg"].numel(), self.n_params))
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        self.step_count = int(st["step"])
        self._sync_device_step()

    # ------------------------------------------------------------------- workspace
    def _alloc_workspace(self, tokens: int) -> None:
        cfg, d, ff = self.cfg, self.cfg.d, self.cfg.d_ff
        f32 = dict(dtype=torch.float32, device=self.device)
        M = self.capacity = tokens
        L = cfg.n_layers
        lib = hip.load()
        act = dict(dtype=torch.bfloat16 if self.bf16_store else torch.float32, device=self.device)
        self.x = torch.empty(L + 1, M, d, **f32)          # residual stream entering each layer (+ final)
        self.h1 = torch.empty(L, M, d, **act)
        self.qkv = torch.empty(L, M, 3 * d, **act)
        self.att = torch.empty(L, M, d, **act)
        self.xmid = torch.empty(L, M, d, **f32)
        self.h2 = torch.empty(L, M, d, **act)
        self.u = torch.empty(L, M, ff, **act)             # FFN pre-activation
        self.gl = None if self.gelu_on_load else torch.empty(L, M, ff, **act)            # gelu(u)
        self.stats = torch.empty(2 * L + 1, 2, M, **f32)  # mean / rstd of every layer-norm
        self.xf = torch.empty(M, d, **act)
        self.out = torch.empty(M, cfg.n_out, **f32)
        self.dout = torch.empty(M, cfg.n_out, **f32)
        self.dx = torch.empty(M, d, **f32)                # gradient of the residual stream
        self.dh = torch.empty(M, d, **act)
        self.du = torch.empty(M, ff, **act)
        self.dqkv = torch.empty(M, 3 * d, **act)
        self.loss_scratch = torch.zeros(lib.vlg_layout_loss_scratch(), **f32)
        if cfg.attention == "clip":       # per query and head: log-sum-exp of every layer (saved for backward), <dO, O> scratch
            self.lse = torch.empty(L, cfg.n_heads * M, **f32)
            self.delta = torch.empty(cfg.n_heads * M, **f32)
        # partial-sum ("slab") arenas: floats needed by the embedding, a layer-norm, and each weight-gradient shape
        emb_len = self.layout["l0.ln1_g"][0]
        need = [lib.vlg_embed_bwd_slabs() * emb_len, lib.vlg_layernorm_bwd_slabs(M) * 2 * d]
        for (n, k) in ((3 * d, d), (d, d), (ff, d), (d, ff), (cfg.n_out, d)):
            need.append(lib.vlg_linear_wgrad_slabs_for(M, n, k, self.gemm_flags) * (n * k + n))
        # option: backward can run the weight gradients on a second HIP stream, concurrently with the data-gradient chain
        # (see backward).  Measured -1.7 % step time at the metric shape (6.12 -> 6.02 ms): a 512-block launch takes every
        # CU slot, so the other stream's kernel only overlaps its tail.  OFF by default: with two kernels sharing the chip
        # a kernel's own duration no longer says anything about that kernel (bench.py's per-kernel roofline).
        self.side = torch.cuda.Stream(device=self.device)
        self.overlap_wgrad = os.environ.get("VLG_OVERLAP_WGRAD", "0") == "1"
        # second option: the bandwidth-bound kernels of backward (layer-norm backward, attention backward) run on the side
        # stream BESIDE the weight-gradient GEMM that does not depend on them (see backward)
        self.overlap_small = os.environ.get("VLG_OVERLAP_SMALL", "0") == "1" and cfg.attention == "slot"
        # Default (VLG_GROUP_REDUCE=1, single-stream backward): the partial-sum producers of one bucket (a layer: four weight
        # gradients + two layer-norms) write side by side into ONE arena and ONE table-driven launch (vlg_reduce_slabs_table,
        # the form the GridNet path uses) reduces them all when the bucket is complete: 27 launches of ~5.5 us -> 6 per step.
        self.group_reduce = os.environ.get("VLG_GROUP_REDUCE", "1") == "1" and not self.overlap_wgrad and not self.overlap_small
        # a projection's data gradient and weight gradient through one call (one launch at few tokens): fp32 tensors, single
        # stream; the library decides per shape (VLG_GEMM_PAIR)
        self.pair_backward = (not self.overlap_wgrad and not self.overlap_small and not self.gelu_on_load
                              and self.precision in ("fp32", "bf16") and os.environ.get("VLG_PAIR_BACKWARD", "1") == "1")
        self.ride_reduces = False
        if self.group_reduce:
            pad = lambda v: (v + 3) // 4 * 4
            layer = sum(pad(v) for v in need[2:6]) + 2 * pad(need[1])
            head = pad(need[6]) + pad(need[1])
            # TWO arenas: a finished bucket's table may be reduced by rider blocks of the NEXT paired launch (which writes its
            # own partial sums into the other arena) instead of by a launch of its own - _join_reduces(defer=True)
            self.garenas = [torch.empty(max(layer, head, pad(need[0])), **f32) for _ in range(2)]
            self._gsel = 0
            self._goff = 0
            self._grows: list = []
            self._gtables: Dict[tuple, torch.Tensor] = {}
            self._pending = None                     # (table rows, callback) of a bucket waiting for its ride
            self.ride_reduces = self.pair_backward and os.environ.get("VLG_RIDE_REDUCE", "1") == "1"
        else:   # one arena per producer family, each reduced right behind its producer (before the next one writes)
            self.slabs = torch.empty(max(need[:2]), **f32)          # layer-norm, embedding (main stream)
            self.slabs_w = torch.empty(max(need[2:]), **f32)        # weight gradients (the side stream in a two-stream backward)

    # --------------------------------------------------------------------- helpers
    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def _arena(self, kind: str, need: int = 0) -> torch.Tensor:
        """the slab arena the next partial-sum producer writes (`need` floats).  Grouped reductions: the next free range of the
        bucket's arena.  Otherwise one arena per producer family ("w": weight gradients, "s": layer-norm / embedding), whose
        previous contents were reduced on the same stream right behind their producer."""
        if self.group_reduce:
            garena = self.garenas[self._gsel]
            off = self._goff
            if off + need > garena.numel():
                raise RuntimeError("partial-sum arena of %d floats is too small for %d more" % (garena.numel(), need))
            self._goff = off + (need + 3) // 4 * 4
            self._gcur = garena.data_ptr() + 4 * off
            return garena[off:off + need]
        return self.slabs_w if kind == "w" else self.slabs

    def _reduce(self, kind: str, stride: int, n_slabs: int, dst_off: int, dst_len: int) -> None:
        """sum the slabs the producer just launched on the current stream wrote into its arena -> grads[dst_off : +dst_len]:
        a row of the bucket's table (grouped), or a launch of its own right behind the producer"""
        dst = self.grads.data_ptr() + 4 * dst_off
        if self.group_reduce:
            self._grows.append((self._gcur, stride, n_slabs, dst, dst_len))
            return
        call("vlg_reduce_slabs", ptr(self.slabs_w if kind == "w" else self.slabs), stride, n_slabs, dst, dst_len, self._stream())

    def _table(self, rows: tuple) -> torch.Tensor:
        table = self._gtables.get(rows)
        if table is None:
            table = torch.tensor([v for row in rows for v in row], dtype=torch.int64, device=self.device)
            self._gtables[rows] = table
        return table

    def _join_reduces(self, defer: bool = False, then=None) -> None:
        """grouped reductions: the bucket that is complete now (a table of {slabs, stride, count, destination, length} rows in
        device memory, built once per batch geometry) is reduced by ONE launch on the current stream - or, with defer=True
        and a paired launch to follow, handed to that launch as rider blocks (vlg_linear_dgrad_wgrad: the next producers then
        write the other arena).  `then` (e.g. reducer.ready of the bucket) runs once the reduction has been enqueued.  A
        bucket still waiting when another one completes travels in the same table."""
        if not self.group_reduce:
            if then is not None:
                then()
            return
        rows = tuple(self._grows)
        self._grows = []
        self._goff = 0
        if defer and self.ride_reduces and rows and self._pending is None:
            self._pending = (rows, then)
            self._gsel ^= 1
            return
        waiting, self._pending = self._pending, None
        if waiting is not None:
            rows = waiting[0] + rows
        if rows:
            call("vlg_reduce_slabs_table", ptr(self._table(rows)), len(rows), 128, self._stream())
        if waiting is not None and waiting[1] is not None:
            waiting[1]()
        if then is not None:
            then()

    def _take_rider(self):
        """(table pointer, rows, callback) of the bucket waiting for a ride, for the paired launch about to be enqueued"""
        if not self.group_reduce or self._pending is None:
            self._rider_bytes = 0.0
            return 0, 0, None
        (rows, then), self._pending = self._pending, None
        self._rider_bytes = 4.0 * sum((n_slabs + 1) * length for (_, _, n_slabs, _, length) in rows)   # slabs read + sums written
        return ptr(self._table(rows)), len(rows), then

    def _timed(self, family: str, flops: float, name: str, *args, nbytes: float = 0.0) -> None:
        """Launch through the C ABI; when a timer is attached, bracket the launch with events on
        the launch stream (torch's current stream IS the stream handed to the kernel)."""
        if self.timer is None or (self.timer.only is not None and family not in self.timer.only):
            call(name, *args)
        else:
            with self.timer.section(family, flops, nbytes):
                call(name, *args)

    @staticmethod
    def _storage_bits(a=None, b=None, out=None) -> int:
        """storage flags of a projection call from the dtypes of its activation operands"""
        bf = torch.bfloat16
        return ((EPI_A_BF16 if a is not None and a.dtype == bf else 0) | (EPI_B_BF16 if b is not None and b.dtype == bf else 0) |
                (EPI_OUT_BF16 if out is not None and out.dtype == bf else 0))

    def _linear(self, a, w, b, c, M, N, K, epi, aux_in=None, aux_out=None):
        nb = (a.element_size() * M * K + w.element_size() * N * K +
              c.element_size() * M * N * (1 + (aux_in is not None) + (aux_out is not None)))
        flags = epi | self.gemm_flags | self._storage_bits(a, w, c)
        self._timed("gemm_fwd" if N > 32 else "gemm_head", 2.0 * M * N * K, "vlg_linear_fwd", ptr(a), K, ptr(w), K,
                    ptr(b), ptr(c), N, ptr(aux_in), ptr(aux_out), M, N, K, flags, self._stream(), nbytes=nb)

    def _dgrad(self, dy, w, dx, M, N, K, epi=EPI_NONE, aux_in=None):
        nb = dy.element_size() * M * N + w.element_size() * N * K + dx.element_size() * M * K * (1 + (aux_in is not None))
        flags = epi | self.gemm_flags | self._storage_bits(dy, w, dx)
        self._timed("gemm_dgrad", 2.0 * M * N * K, "vlg_linear_dgrad", ptr(dy), N, ptr(w), K, ptr(dx), K,
                    ptr(aux_in), M, N, K, flags, self._stream(), nbytes=nb)

    def _wgrad(self, dy, x, wname, M, N, K, extra=0):
        """grad[w | b] = (dy^T . x | colsum dy): split partials -> slab arena -> flat gradient.  extra = EPI_ACT_GELU: x holds
        pre-activations, the kernel applies GELU while staging it.  Runs on the CURRENT stream (backward makes that the
        side stream)."""
        lib = hip.load()
        stride = N * K + N
        n_slabs = lib.vlg_linear_wgrad_slabs_for(M, N, K, self.gemm_flags)
        arena = self._arena("w", n_slabs * stride)
        s = self._stream()
        self._timed("gemm_wgrad" if N > 32 else "gemm_head", 2.0 * M * N * K, "vlg_linear_wgrad", ptr(dy), N, ptr(x),
                    K, ptr(arena), stride, arena.numel(), M, N, K, self.gemm_flags | self._storage_bits(dy, x) | extra, s,
                    nbytes=dy.element_size() * M * N + x.element_size() * M * K + 4.0 * n_slabs * stride)
        self._reduce("w", stride, n_slabs, self.layout[wname][0], stride)

    def _dgrad_wgrad(self, dy, w, dx, x, wname, M, N, K, epi=EPI_NONE, aux_in=None):
        """backward of one projection y = x W^T + b given dy: grad[w | b] (slab partials -> flat gradient) AND dx = dy . W
        (x aux_in with EPI_MUL) through ONE C-ABI call, which the library runs as one launch where neither product fills
        the chip alone (few tokens per GPU; vlg_linear_dgrad_wgrad).  Same results, bit for bit, as _wgrad then _dgrad."""
        lib = hip.load()
        stride = N * K + N
        n_slabs = lib.vlg_linear_wgrad_slabs_for(M, N, K, self.gemm_flags)
        arena = self._arena("w", n_slabs * stride)
        nb = (dy.element_size() * M * N * 2 + w.element_size() * N * K + dx.element_size() * M * K * (1 + (aux_in is not None)) +
              x.element_size() * M * K + 4.0 * n_slabs * stride)
        bits = self._storage_bits(dy, w, dx)       # bf16 mode: A = the shared dY, B = W (data gradient) AND X (weight gradient), OUT = dX
        if (x.dtype == torch.bfloat16) != (w.dtype == torch.bfloat16):
            raise ValueError("paired backward: X and W must have the same storage type")
        rider, rider_rows, then = self._take_rider()       # a finished bucket's reduction rides in this launch (its bytes count)
        self._timed("gemm_pair", 4.0 * M * N * K, "vlg_linear_dgrad_wgrad", ptr(dy), N, ptr(w), K, ptr(dx), K, ptr(aux_in),
                    ptr(x), K, ptr(arena), stride, arena.numel(), M, N, K, epi | self.gemm_flags | bits, rider, rider_rows,
                    self._stream(), nbytes=nb + self._rider_bytes)
        if then is not None:
            then()
        self._reduce("w", stride, n_slabs, self.layout[wname][0], stride)

    def _attn_fwd(self, l: int, batch, B, T, N, M) -> None:
        d, s = self.cfg.d, self._stream()
        if self.cfg.attention == "clip":
            fl = 4.0 * B * d * N * N * T * (T + 1) / 2.0            # visible (query, key) pairs x 2 products x 2 x head dim, all heads
            self._timed("attn_clip_fwd", fl, "vlg_attention_clip_fwd", ptr(self.qkv[l]), ptr(batch["valid"]) if self.padded_slots else 0,
                        ptr(self.att[l]), ptr(self.lse[l]), B, T, N, d, s, nbytes=16.0 * M * d)
            return
        self._timed("attn_fwd", 0.0, "vlg_attention_fwd" + self._sfx, ptr(self.qkv[l]), ptr(self.att[l]), B * N, T, d, s,
                    nbytes=4.0 * self.qkv.element_size() * M * d)

    def _attn_bwd(self, l: int, batch, B, T, N, M) -> None:
        d, s = self.cfg.d, self._stream()
        if self.cfg.attention == "clip":
            fl = 2.5 * 4.0 * B * d * N * N * T * (T + 1) / 2.0       # ALGORITHMIC: 5 products against the forward's 2 (the two-kernel
                                                                     # backward recomputes S and dP: 7 are executed)
            self._timed("attn_clip_bwd", fl, "vlg_attention_clip_bwd", ptr(self.qkv[l]), ptr(batch["valid"]) if self.padded_slots else 0,
                        ptr(self.att[l]), ptr(self.dh), ptr(self.lse[l]), ptr(self.delta), ptr(self.dqkv), B, T, N, d, s,
                        nbytes=28.0 * M * d)
            return
        self._timed("attn_bwd", 0.0, "vlg_attention_bwd" + self._sfx, ptr(self.qkv[l]), ptr(self.dh), ptr(self.dqkv), B * N, T, d, s,
                    nbytes=7.0 * self.qkv.element_size() * M * d)

    def _ln_fwd(self, x, gname, y, stat, M):
        d = self.cfg.d
        self._timed("ln_fwd", 0.0, "vlg_layernorm_fwd_bf16" if y.dtype == torch.bfloat16 else "vlg_layernorm_fwd", ptr(x), ptr(self.p(gname)),
                    ptr(self.p(gname[:-1] + "b")), ptr(y), ptr(stat[0]), ptr(stat[1]), M, d, LN_EPS, self._stream(),
                    nbytes=(4.0 + y.element_size()) * M * d)

    def _ln_bwd(self, dy, x, stat, gname, dres, dx_out, M):
        d = self.cfg.d
        lib = hip.load()
        n_slabs = lib.vlg_layernorm_bwd_slabs(M)
        arena = self._arena("s", n_slabs * 2 * d)
        s = self._stream()
        self._timed("ln_bwd", 0.0, "vlg_layernorm_bwd_bf16" if dy.dtype == torch.bfloat16 else "vlg_layernorm_bwd", ptr(dy), ptr(x), ptr(stat[0]),
                    ptr(stat[1]), ptr(self.p(gname)), ptr(dres), ptr(dx_out), ptr(arena), 2 * d, arena.numel(), M, d, s,
                    nbytes=(dy.element_size() + 4.0 + 4.0 + (4.0 if dres is not None else 0.0)) * M * d)
        self._reduce("s", 2 * d, n_slabs, self.layout[gname][0], 2 * d)

    def _check_batch(self, batch) -> tuple:
        sc = batch["slot_class"]
        B, T, N = sc.shape
        if T != self.cfg.T:
            raise ValueError("batch has T=%d, engine was built for T=%d" % (T, self.cfg.T))
        M = B * T * N
        if M > self.capacity:
            raise ValueError("batch of %d tokens exceeds workspace capacity %d" % (M, self.capacity))
        for k, dt in (("slot_class", torch.int64), ("slot_box", torch.float32), ("tgt_class", torch.int64),
                      ("tgt_box", torch.float32), ("valid", torch.float32)):
            t = batch[k]
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise ValueError("batch[%r] must be a contiguous %s HIP tensor" % (k, dt))
        return B, T, N, M

    # --------------------------------------------------------------------- forward
    def forward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Forward + fused loss (the loss kernel also leaves d(total)/d(out) in self.dout).
        Returns the device tensor {total, smooth_l1, iou, ce}."""
        cfg, d, ff = self.cfg, self.cfg.d, self.cfg.d_ff
        B, T, N, M = self._check_batch(batch)
        self._shape = (B, T, N, M)
        s = self._stream()
        self._timed("embed_fwd", 0.0, "vlg_embed_fwd", ptr(batch["slot_class"]), ptr(batch["slot_box"]), ptr(self.p("cls_emb")),
                    ptr(self.p("box_w")), ptr(self.p("box_b")), ptr(self.p("time_emb")), ptr(self.x[0]),
                    B, T, N, d, cfg.vocab, s, nbytes=4.0 * M * d + 24.0 * M)
        for l in range(cfg.n_layers):
            pre = "l%d." % l
            x = self.x[l]
            self._ln_fwd(x, pre + "ln1_g", self.h1[l], self.stats[2 * l], M)
            self._linear(self.h1[l], self.pw(pre + "qkv_w"), self.p(pre + "qkv_b"), self.qkv[l], M, 3 * d, d, EPI_BIAS)
            self._attn_fwd(l, batch, B, T, N, M)
            self._linear(self.att[l], self.pw(pre + "proj_w"), self.p(pre + "proj_b"), self.xmid[l], M, d, d,
                         EPI_BIAS | EPI_RESID, aux_in=x)
            self._ln_fwd(self.xmid[l], pre + "ln2_g", self.h2[l], self.stats[2 * l + 1], M)
            if self.gelu_on_load:
                self._linear(self.h2[l], self.pw(pre + "ff1_w"), self.p(pre + "ff1_b"), self.u[l], M, ff, d, EPI_BIAS)
                self._linear(self.u[l], self.pw(pre + "ff2_w"), self.p(pre + "ff2_b"), self.x[l + 1], M, d, ff,
                             EPI_BIAS | EPI_RESID | EPI_ACT_GELU, aux_in=self.xmid[l])
            else:
                self._linear(self.h2[l], self.pw(pre + "ff1_w"), self.p(pre + "ff1_b"), self.gl[l], M, ff, d,
                             self._epi_ff1, aux_out=self.u[l])
                self._linear(self.gl[l], self.pw(pre + "ff2_w"), self.p(pre + "ff2_b"), self.x[l + 1], M, d, ff,
                             EPI_BIAS | EPI_RESID, aux_in=self.xmid[l])
        L = cfg.n_layers
        self._ln_fwd(self.x[L], "lnf_g", self.xf, self.stats[2 * L], M)
        self._linear(self.xf, self.pw("head_w"), self.p("head_b"), self.out, M, cfg.n_out, d, EPI_BIAS)
        self._timed("loss", 0.0, "vlg_layout_loss", ptr(self.out), cfg.n_out, ptr(batch["tgt_class"]), ptr(batch["tgt_box"]),
                    ptr(batch["valid"]), ptr(self.dout), ptr(self.loss_out), ptr(self.loss_scratch), B, T, N,
                    cfg.n_classes, SMOOTH_L1_BETA, IOU_EPS, LOSS_W_REG, LOSS_W_STRUCT, LOSS_W_CE, s,
                    nbytes=(2.0 * 4 * cfg.n_out + 8 + 16 + 4) * M)
        return self.loss_out

    # -------------------------------------------------------------------- backward
    def backward(self, batch: Dict[str, torch.Tensor], reducer=None) -> None:
        """Fills self.grads (every element overwritten) from the state forward() left.  `reducer`
        (vlg.dp.GradReducer) is told as soon as each contiguous gradient bucket is complete so its
        all-reduce overlaps the rest of backward.

        Two HIP streams.  The data-gradient chain (dgrad GEMMs, attention / layer-norm backward) stays on the caller's
        stream; every weight gradient (+ its slab reduction) is launched on `self.side` as soon as its dY exists and runs
        CONCURRENTLY with the chain.  Each GEMM launch leaves the matrix pipes idle while its blocks load their first tiles
        and drain their stores (all blocks of a launch do that in step: 10-25 % of a K = 256 launch), and the
        bandwidth-bound kernels of the chain leave them idle altogether; blocks of the other stream's kernel fill those
        gaps.  Ordering is by events: a side kernel waits for the producer of its dY, and the chain waits before it
        overwrites a buffer a side kernel still reads (du, dqkv, dx) and before a bucket is handed to the reducer."""
        cfg, d, ff = self.cfg, self.cfg.d, self.cfg.d_ff
        B, T, N, M = self._shape
        main = torch.cuda.current_stream(self.device)
        side = self.side if self.overlap_wgrad else main
        s = main.cuda_stream
        L = cfg.n_layers
        last_read: Dict[str, torch.cuda.Event] = {}

        def on_side(reads, fn):
            """launch fn on the side stream once everything enqueued on the main stream so far is done; remember when the
            buffers it reads become free again"""
            if side is main:
                fn()
                return
            e = torch.cuda.Event()
            e.record(main)
            side.wait_event(e)
            with torch.cuda.stream(side):
                fn()
                done = torch.cuda.Event()
                done.record(side)
            for name in reads:
                last_read[name] = done

        def before_write(*names):
            for name in names:
                e = last_read.pop(name, None)
                if e is not None:
                    main.wait_event(e)

        def join():
            if side is not main:
                e = torch.cuda.Event()
                e.record(side)
                main.wait_event(e)
                last_read.clear()

        if self.group_reduce:
            self._gsel = 0          # every step uses the arenas in the same order: the reduction tables (device memory, cached by
                                    # their rows) are the same from step to step - also what a captured hipGraph needs
        on_side(("dout",), lambda: self._wgrad(self.dout, self.xf, "head_w", M, cfg.n_out, d))
        self._dgrad(self.dout, self.pw("head_w"), self.dh, M, cfg.n_out, d)
        self._ln_bwd(self.dh, self.x[L], self.stats[2 * L], "lnf_g", None, self.dx, M)
        paired = self.pair_backward and self._pair_shapes(M)
        ready = (lambda tag: (lambda: reducer.ready(tag))) if reducer is not None else (lambda tag: None)
        if reducer is not None or self.group_reduce:
            join()
            self._join_reduces(defer=paired, then=ready("head"))      # (paired: rides in the last layer's first launch)
        if self.overlap_small and not self.overlap_wgrad:
            self._backward_layers_paired(B, T, N, M, reducer)
            self._backward_tail(batch, B, T, N, M, reducer)
            return
        for l in reversed(range(L)):
            pre = "l%d." % l
            if paired:
                self._dgrad_wgrad(self.dx, self.pw(pre + "ff2_w"), self.du, self.gl[l], pre + "ff2_w", M, d, ff, self._epi_dff2, aux_in=self.u[l])
                self._dgrad_wgrad(self.du, self.pw(pre + "ff1_w"), self.dh, self.h2[l], pre + "ff1_w", M, ff, d)
                self._ln_bwd(self.dh, self.xmid[l], self.stats[2 * l + 1], pre + "ln2_g", self.dx, self.dx, M)
                self._dgrad_wgrad(self.dx, self.pw(pre + "proj_w"), self.dh, self.att[l], pre + "proj_w", M, d, d)
                self._attn_bwd(l, batch, B, T, N, M)
                self._dgrad_wgrad(self.dqkv, self.pw(pre + "qkv_w"), self.dh, self.h1[l], pre + "qkv_w", M, 3 * d, d)
                self._ln_bwd(self.dh, self.x[l], self.stats[2 * l], pre + "ln1_g", self.dx, self.dx, M)
                # the layer's bucket rides in the next layer's first paired launch (layer 0: in the embedding bucket's table)
                self._join_reduces(defer=True, then=ready("l%d" % l))
                continue
            # FFN:  x_out = xmid + W2 gelu(W1 h2 + b1) + b2
            if self.gelu_on_load:
                on_side(("dx",), lambda: self._wgrad(self.dx, self.u[l], pre + "ff2_w", M, d, ff, extra=EPI_ACT_GELU))
            else:
                on_side(("dx",), lambda: self._wgrad(self.dx, self.gl[l], pre + "ff2_w", M, d, ff))
            before_write("du")
            self._dgrad(self.dx, self.pw(pre + "ff2_w"), self.du, M, d, ff, self._epi_dff2, aux_in=self.u[l])
            on_side(("du",), lambda: self._wgrad(self.du, self.h2[l], pre + "ff1_w", M, ff, d))
            self._dgrad(self.du, self.pw(pre + "ff1_w"), self.dh, M, ff, d)
            before_write("dx")
            self._ln_bwd(self.dh, self.xmid[l], self.stats[2 * l + 1], pre + "ln2_g", self.dx, self.dx, M)
            # attention:  xmid = x + Wo attn(Wqkv h1 + b) + bo
            on_side(("dx",), lambda: self._wgrad(self.dx, self.att[l], pre + "proj_w", M, d, d))
            self._dgrad(self.dx, self.pw(pre + "proj_w"), self.dh, M, d, d)
            before_write("dqkv")
            self._attn_bwd(l, batch, B, T, N, M)
            on_side(("dqkv",), lambda: self._wgrad(self.dqkv, self.h1[l], pre + "qkv_w", M, 3 * d, d))
            self._dgrad(self.dqkv, self.pw(pre + "qkv_w"), self.dh, M, 3 * d, d)
            before_write("dx")
            self._ln_bwd(self.dh, self.x[l], self.stats[2 * l], pre + "ln1_g", self.dx, self.dx, M)
            if reducer is not None or self.group_reduce:
                join()
                self._join_reduces()
            if reducer is not None:
                reducer.ready("l%d" % l)
        join()
        self._backward_tail(batch, B, T, N, M, reducer)

    def _pair_shapes(self, M: int) -> bool:
        """the paired backward (one C-ABI call per projection; the library fuses the two launches where its VLG_GEMM_PAIR mode
        says so: 2 = every shape, the default; 1 = few tokens only - then larger batches keep the separate calls and with
        them bench.py's per-kernel timing families; 0 = never)"""
        mode = int(os.environ.get("VLG_GEMM_PAIR", "2"))
        return mode >= 2 or (mode == 1 and M <= 16384)

    def _backward_tail(self, batch, B, T, N, M, reducer) -> None:
        cfg, d = self.cfg, self.cfg.d
        lib = hip.load()
        s = self._stream()
        emb_len = self.layout["l0.ln1_g"][0]
        n_slabs = lib.vlg_embed_bwd_slabs_for(B, T, N, d, cfg.vocab)          # by shape: few clips write (and reduce) few slabs
        arena = self._arena("s", n_slabs * emb_len)
        self._timed("embed_bwd", 0.0, "vlg_embed_bwd", ptr(self.dx), ptr(batch["slot_class"]), ptr(batch["slot_box"]), ptr(arena),
                    emb_len, arena.numel(), B, T, N, d, cfg.vocab, s, nbytes=4.0 * M * d + 24.0 * M)
        self._reduce("s", emb_len, n_slabs, 0, emb_len)
        # (flushes a bucket still waiting for a ride in the same table: the gradient buffer is complete for whoever runs next)
        self._join_reduces(then=(lambda: reducer.ready("embed")) if reducer is not None else None)

    def _backward_layers_paired(self, B, T, N, M, reducer) -> None:
        """Backward of the layers with every bandwidth-bound kernel of the chain launched on the side stream BESIDE the
        weight-gradient GEMM that does not depend on it: layer-norm-2 backward beside the FFN1 weight gradient, attention
        backward beside the output-projection weight gradient, layer-norm-1 backward beside the QKV weight gradient.  The
        GEMM is bound by the matrix pipes and leaves most of the HBM bandwidth idle; its partner needs no LDS and few
        registers, so its blocks fit next to the GEMM's on every CU.  Each pair is fork -> two launches -> join."""
        cfg, d, ff = self.cfg, self.cfg.d, self.cfg.d_ff
        main = torch.cuda.current_stream(self.device)
        side = self.side

        def beside(side_fn, main_fn):
            e = torch.cuda.Event()
            e.record(main)
            side.wait_event(e)
            with torch.cuda.stream(side):
                side_fn()
                done = torch.cuda.Event()
                done.record(side)
            main_fn()
            main.wait_event(done)

        for l in reversed(range(cfg.n_layers)):
            pre = "l%d." % l
            self._wgrad(self.dx, self.gl[l], pre + "ff2_w", M, d, ff)
            self._dgrad(self.dx, self.pw(pre + "ff2_w"), self.du, M, d, ff, self._epi_dff2, aux_in=self.u[l])
            self._dgrad(self.du, self.pw(pre + "ff1_w"), self.dh, M, ff, d)
            beside(lambda: self._ln_bwd(self.dh, self.xmid[l], self.stats[2 * l + 1], pre + "ln2_g", self.dx, self.dx, M),
                   lambda: self._wgrad(self.du, self.h2[l], pre + "ff1_w", M, ff, d))
            self._dgrad(self.dx, self.pw(pre + "proj_w"), self.dh, M, d, d)
            beside(lambda: self._timed("attn_bwd", 0.0, "vlg_attention_bwd" + self._sfx, ptr(self.qkv[l]), ptr(self.dh), ptr(self.dqkv),
                                       B * N, T, d, self._stream(), nbytes=7.0 * self.qkv.element_size() * M * d),
                   lambda: self._wgrad(self.dx, self.att[l], pre + "proj_w", M, d, d))
            self._dgrad(self.dqkv, self.pw(pre + "qkv_w"), self.dh, M, 3 * d, d)
            beside(lambda: self._ln_bwd(self.dh, self.x[l], self.stats[2 * l], pre + "ln1_g", self.dx, self.dx, M),
                   lambda: self._wgrad(self.dqkv, self.h1[l], pre + "qkv_w", M, 3 * d, d))
            if reducer is not None:
                self._join_reduces()
                reducer.ready("l%d" % l)

    def forward_backward(self, batch: Dict[str, torch.Tensor], reducer=None) -> torch.Tensor:
        loss = self.forward(batch)
        self.backward(batch, reducer)
        return loss

    def train_step(self, batch: Dict[str, torch.Tensor], reducer=None) -> torch.Tensor:
        """forward -> loss -> backward (+ overlapped gradient all-reduce) -> Adam, as one call.
        Returns the loss scalars (summed over ranks when a reducer is attached)."""
        loss = self.forward_backward(batch, reducer)
        if reducer is not None:
            # the embedding bucket is the last to be produced, so its all-reduce would be fully exposed: update
            # everything else while it is in flight, then the embedding range
            split = self.layout["l0.ln1_g"][0]
            reducer.wait(keep=("embed",))
            self.adam_step(reducer.grad_scale, lo=split, hi=self.n_params)
            reducer.wait()
            self.adam_step(reducer.grad_scale, lo=0, hi=split, advance=False)
        else:
            self.adam_step()
        return loss

    # ------------------------------------------------------------------- optimiser
    def adam_step(self, grad_scale: float = 1.0, lo: int = 0, hi: Optional[int] = None, advance: bool = True) -> None:
        """torch.optim.Adam(lr, betas=(beta1, 0.999)) on the flat buffer (reference src/trainer.py:83,258), or on its
        [lo, hi) slice (both multiples of 4); `advance` = False keeps the step count (second slice of one step)."""
        hi = self.n_params if hi is None else hi
        if advance:
            self.step_count += 1
        o = 4 * lo
        shadow = self.params_bf16.data_ptr() + o // 2 if self.params_bf16 is not None else 0
        if self.adam_state is not None:          # captured / capturable step: the counter and its factors live on the device
            call("vlg_adam_step_graph", self.params.data_ptr() + o, self.grads.data_ptr() + o, self.exp_avg.data_ptr() + o,
                 self.exp_avg_sq.data_ptr() + o, shadow, hi - lo, ptr(self.adam_state), 1 if advance else 0, self.lr,
                 self.beta1, ADAM_BETA2, ADAM_EPS, grad_scale, self._stream())
            return
        if self.params_bf16 is not None:
            call("vlg_adam_step_bf16", self.params.data_ptr() + o, self.grads.data_ptr() + o, self.exp_avg.data_ptr() + o,
                 self.exp_avg_sq.data_ptr() + o, shadow, hi - lo, self.step_count, self.lr,
                 self.beta1, ADAM_BETA2, ADAM_EPS, grad_scale, self._stream())
            return
        self._timed("adam", 0.0, "vlg_adam_step", self.params.data_ptr() + o, self.grads.data_ptr() + o, self.exp_avg.data_ptr() + o,
                    self.exp_avg_sq.data_ptr() + o, hi - lo, self.step_count, self.lr, self.beta1, ADAM_BETA2, ADAM_EPS,
                    grad_scale, self._stream(), nbytes=28.0 * (hi - lo))

    # ------------------------------------------------------------------- hipGraph
    def use_device_step_counter(self) -> None:
        """Move Adam's step counter (and the bias-correction factors derived from it) to device memory, the form a
        captured step needs; the host count `step_count` keeps mirroring it."""
        if self.adam_state is None:
            st = torch.zeros(4, dtype=torch.float32, device=self.device)
            st.view(torch.int32)[2] = int(self.step_count)
            self.adam_state = st

    def capture_train_step(self, example_batch: Dict[str, torch.Tensor]):
        """Capture forward -> loss -> backward -> Adam for this batch SHAPE in one hipGraph (single process: the
        data-parallel hooks are not captured) and return `run(batch) -> loss scalars`: it copies the batch into the
        graph's static input buffers and replays ~110 kernel launches with one host call.  Everything the step
        touches is preallocated and no launch argument changes between steps (the Adam step counter moves to the
        device), so replay is bitwise identical to the eager step."""
        self._check_batch(example_batch)
        self.use_device_step_counter()
        static = {k: example_batch[k].clone() for k in ("slot_class", "slot_box", "tgt_class", "tgt_box", "valid")}
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        keep = (self.params.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.adam_state.clone(), self.step_count)
        with torch.cuda.stream(side):            # warm-up on a side stream, as stream capture requires
            self.train_step(static)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        # undo the warm-up step: capture must not change the training trajectory
        self.params.copy_(keep[0]); self.exp_avg.copy_(keep[1]); self.exp_avg_sq.copy_(keep[2])
        self.adam_state.copy_(keep[3]); self.step_count = keep[4]
        self._refresh_shadow()
        timer, self.timer = self.timer, None     # events are not capturable work
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            self.train_step(static)
        self.timer = timer
        self.params.copy_(keep[0]); self.exp_avg.copy_(keep[1]); self.exp_avg_sq.copy_(keep[2])
        self.adam_state.copy_(keep[3]); self.step_count = keep[4]
        self._refresh_shadow()

        def run(batch: Dict[str, torch.Tensor]) -> torch.Tensor:
            for k, t in static.items():
                if batch[k] is not t:
                    t.copy_(batch[k], non_blocking=True)
            graph.replay()
            self.step_count += 1
            return self.loss_out
        run.graph, run.static_batch = graph, static
        return run

    # ---------------------------------------------------------------- public views
    def outputs_btn(self) -> tuple:
        """(class logits (B,T,N,C), raw boxes (B,T,N,4)) of the last forward, in public order."""
        B, T, N, M = self._shape
        o = self.out[:M].view(B, N, T, self.cfg.n_out).permute(0, 2, 1, 3)
        return o[..., :self.cfg.n_classes], o[..., self.cfg.n_classes:]


class KernelTimer:
    """HIP-event timing of selected launches on the launch stream (bench.py roofline leg).

    Events are recorded on torch's current stream, which is the stream passed to the kernels,
    so each start/end pair brackets exactly one launch; durations are read after a sync."""

    def __init__(self, only=None):
        self.records = {}          # family -> list of (start, end, flops)
        self.only = only           # None = every family, else a tuple of family names to bracket

    class _Section:
        def __init__(self, timer, family, flops, nbytes=0.0):
            self.t, self.family, self.flops, self.nbytes = timer, family, flops, nbytes

        def __enter__(self):
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

        def __exit__(self, *a):
            self.e.record()
            self.t.records.setdefault(self.family, []).append((self.s, self.e, self.flops, self.nbytes))

    def section(self, family: str, flops: float, nbytes: float = 0.0):
        return KernelTimer._Section(self, family, flops, nbytes)

    def summary(self):
        """family -> {launches, avg_ms, total_ms, flops_per_launch} (call after a device sync)."""
        out = {}
        for fam, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            fl = [f for _, _, f, _ in recs]
            nb = [b for _, _, _, b in recs]
            out[fam] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms),
                        "flops_per_launch": sum(fl) / len(fl), "bytes_per_launch": sum(nb) / len(nb)}
        return out
