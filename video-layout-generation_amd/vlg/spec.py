"""Frozen contract of the per-clip layout-generation training step.

Two kinds of constants live here:

* reference-real ones, each citing the reference line it restates
  (loss weights, Adam hyper-parameters, class count, seed, normalisation
  constants), and
* self-chosen ones for the layout-token model that BASELINE.json names but the
  reference does not contain: depth, heads, FFN width,
  attention pattern, box parameterisation.  Those are marked SELF-ORACLE and are
  printed on every bench line.

Nothing in this file touches the GPU; it is shared by the HIP engine, the
trainer, bench.py and the tests.
"""
from __future__ import annotations

import dataclasses
import math
from collections import OrderedDict
from typing import Dict, Tuple

# ---- reference-real constants -------------------------------------------------
N_CLASSES = 20                  # reference src/models/gridnet.py:9 (seg_out = 20), trainer.py:31-52
LOSS_W_REG = 40.0               # reference src/trainer.py:248  (L1 term  * 40)
LOSS_W_STRUCT = 20.0            # reference src/trainer.py:249  (CombinedLoss * 20)
LOSS_W_CE = 10.0                # reference src/trainer.py:250  (cross entropy * 10)
ADAM_LR = 2e-4                  # reference src/main.py:139-140
ADAM_BETA1 = 0.5                # reference src/main.py:141
ADAM_BETA2 = 0.999              # reference src/trainer.py:83
ADAM_EPS = 1e-8                 # torch.optim.Adam default used at trainer.py:83
SEED = 1024                     # reference src/main.py:121
IMG_MEAN = (0.485, 0.456, 0.406)        # reference src/trainer.py:123
IMG_STD = (0.229, 0.224, 0.225)         # reference src/trainer.py:122
OUT_MEAN = (-0.03, -0.088, -0.188)      # reference src/trainer.py:120
OUT_STD = (0.448, 0.448, 0.450)         # reference src/trainer.py:121

# ---- SELF-ORACLE constants (no reference counterpart) -------------------------
HEAD_DIM = 64                   # one 64-lane wavefront row per head
SMOOTH_L1_BETA = 0.1            # boxes are normalised to [0,1]; beta=0.1 exercises both branches
IOU_EPS = 1e-7
LN_EPS = 1e-5
BOX_DIM = 4                     # (cx, cy, w, h), all in [0,1]


@dataclasses.dataclass(frozen=True)
class LayoutConfig:
    """Shape of one batch of clips and of the token model (SELF-ORACLE)."""
    B: int = 32                 # clips per step per GPU
    T: int = 16                 # frames per clip
    N: int = 64                 # object slots per frame
    d: int = 256                # token width
    n_layers: int = 4
    n_classes: int = N_CLASSES
    # "slot": causal attention along T independently per (clip, slot) - the default the headline is quoted on;
    # "clip": block-causal attention over all T*N tokens of a clip (token (t, n) sees every slot of frames <= t)
    attention: str = "slot"

    @property
    def n_heads(self) -> int:
        return self.d // HEAD_DIM

    @property
    def d_ff(self) -> int:
        return 4 * self.d

    @property
    def vocab(self) -> int:
        # classes + 1 reserved id, as the only nn.Embedding in the reference does
        # (src/models/simple.py:23 "29+1(cropped)")
        return self.n_classes + 1

    @property
    def n_out(self) -> int:
        return self.n_classes + BOX_DIM

    @property
    def tokens(self) -> int:
        return self.B * self.T * self.N

    def validate(self) -> None:
        if self.d % HEAD_DIM != 0:
            raise ValueError("d must be a multiple of %d" % HEAD_DIM)
        if self.T not in (4, 8, 16, 32):
            raise ValueError("T must be one of 4, 8, 16, 32 (temporal tile held by one wavefront)")
        if self.N < 1 or self.B < 1 or self.n_layers < 1:
            raise ValueError("B, N, n_layers must be >= 1")
        if self.attention not in ("slot", "clip"):
            raise ValueError("attention must be 'slot' or 'clip'")
        if self.attention == "clip" and (self.T * self.N) % 32 != 0:
            raise ValueError("attention='clip' needs T*N to be a multiple of 32 (32-token MFMA tiles)")

    def describe(self) -> Dict[str, object]:
        return {"B": self.B, "T": self.T, "N": self.N, "d": self.d, "layers": self.n_layers,
                "heads": self.n_heads, "d_ff": self.d_ff, "classes": self.n_classes,
                "attention": ("causal-temporal-per-slot" if self.attention == "slot"
                              else "block-causal-per-clip (every slot of frames <= t)"), "box": "cxcywh-sigmoid",
                "loss": "40*smoothL1(beta=%g)+20*(1-IoU)+10*CE" % SMOOTH_L1_BETA}


def param_shapes(cfg: LayoutConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Names and shapes of every trainable tensor, in flat-buffer order.

    The order is chosen for the backward pass: the tensors whose gradients are
    finished LAST (embeddings) come first and the ones finished FIRST (heads,
    last layer) come last, so gradient buckets can be all-reduced from the tail
    of the flat buffer while earlier layers are still in backward.
    """
    d, ff = cfg.d, cfg.d_ff
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["cls_emb"] = (cfg.vocab, d)
    s["box_w"] = (d, BOX_DIM)
    s["box_b"] = (d,)
    s["time_emb"] = (cfg.T, d)
    for l in range(cfg.n_layers):
        p = "l%d." % l
        s[p + "ln1_g"] = (d,)
        s[p + "ln1_b"] = (d,)
        s[p + "qkv_w"] = (3 * d, d)
        s[p + "qkv_b"] = (3 * d,)
        s[p + "proj_w"] = (d, d)
        s[p + "proj_b"] = (d,)
        s[p + "ln2_g"] = (d,)
        s[p + "ln2_b"] = (d,)
        s[p + "ff1_w"] = (ff, d)
        s[p + "ff1_b"] = (ff,)
        s[p + "ff2_w"] = (d, ff)
        s[p + "ff2_b"] = (d,)
    s["lnf_g"] = (d,)
    s["lnf_b"] = (d,)
    s["head_w"] = (cfg.n_out, d)
    s["head_b"] = (cfg.n_out,)
    return s


def param_layout(cfg: LayoutConfig) -> Tuple["OrderedDict[str, Tuple[int, Tuple[int, ...]]]", int]:
    """name -> (offset in floats, shape); every offset is a multiple of 4 floats (16 B)."""
    out: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
    off = 0
    for name, shape in param_shapes(cfg).items():
        n = int(math.prod(shape))
        out[name] = (off, shape)
        off += (n + 3) // 4 * 4
    return out, off


def step_flops(cfg: LayoutConfig) -> Dict[str, float]:
    """Algorithmic FLOPs of one training step (Spec N)."""
    M, d = cfg.tokens, cfg.d
    per_layer_fwd = 24.0 * M * d * d            # QKV 6 + proj 2 + FFN 16  (x M d^2)
    attn_fwd = 4.0 * cfg.B * cfg.N * cfg.T * cfg.T * d / 2.0   # causal half
    if cfg.attention == "clip":                                # allowed (query, key) pairs per clip: N^2 T (T + 1) / 2
        attn_fwd = 4.0 * cfg.B * d * cfg.N * cfg.N * cfg.T * (cfg.T + 1) / 2.0
    head_fwd = 2.0 * M * d * cfg.n_out
    fwd = cfg.n_layers * (per_layer_fwd + attn_fwd) + head_fwd
    # backward = 2 x forward for the projections; attention backward = 2.5 x its forward (5 products against 2)
    return {"gemm_fwd_per_layer": per_layer_fwd, "fwd": fwd, "fwd_bwd": 3.0 * fwd + 0.5 * cfg.n_layers * attn_fwd}
