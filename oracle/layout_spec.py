"""TEST INFRASTRUCTURE ONLY - CPU oracle for the layout-token training step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  Nothing under video-layout-generation_amd/ does.

PARITY UNPINNED / SELF-ORACLE: the reference (gongaa/video-layout-generation)
contains no layout-token model (no slot embedding, temporal
encoder, layer-norm, smooth-L1 or IoU anywhere in /root/reference/src).  This file
is therefore the arithmetic DEFINITION of those ops, written with stock
torch-CPU primitives (F.embedding, F.layer_norm, softmax, F.gelu,
F.smooth_l1_loss, F.cross_entropy) and differentiated by torch autograd.  The
pieces that DO have a reference counterpart are restated from it and cited:

* embedding lookup   - plain nn.Embedding row gather, classes + 1 reserved id
                       (reference src/models/simple.py:23,41)
* cross entropy      - nn.CrossEntropyLoss(reduction='mean')
                       (reference src/trainer.py:124,250)
* loss weighting     - 40 * regression + 20 * structure + 10 * CE
                       (reference src/trainer.py:248-251)
* Adam               - torch.optim.Adam(lr, betas=(beta1, 0.999))
                       (reference src/trainer.py:83,258; src/main.py:139-141)
* gradient averaging - DDP mean over ranks (reference src/trainer.py:113)

Tensor conventions (public, reference-style batch-major):
    slot_class  int64 (B,T,N)      class id per slot, values 0..n_classes (n_classes = reserved)
    slot_box    f32   (B,T,N,4)    (cx,cy,w,h) in [0,1]
    tgt_class   int64 (B,T,N)      next-frame class
    tgt_box     f32   (B,T,N,4)    next-frame box
    valid       f32   (B,T,N)      1 = slot scored, 0 = padded slot
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

HEAD_DIM = 64
LN_EPS = 1e-5
SMOOTH_L1_BETA = 0.1
IOU_EPS = 1e-7
W_REG, W_IOU, W_CE = 40.0, 20.0, 10.0   # reference src/trainer.py:248-250


def init_params(shapes: "Dict[str, Tuple[int, ...]]", seed: int = 1024, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Default torch initialisers for every tensor, from one generator.

    nn.Embedding -> N(0,1) (what reference src/models/simple.py:23 gets by default),
    nn.Linear    -> U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias,
    LayerNorm    -> gain 1, bias 0.
    Seed default 1024 = reference src/main.py:121.
    """
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    fan_in_of_bias: Dict[str, int] = {}
    for name, shape in shapes.items():
        if name.endswith("_w"):
            fan_in_of_bias[name[:-2] + "_b"] = shape[1]
    for name, shape in shapes.items():
        base = name.split(".")[-1]
        if base in ("cls_emb", "time_emb"):
            t = torch.randn(shape, generator=g, dtype=torch.float32)
        elif base.endswith("_g"):
            t = torch.ones(shape)
        elif base.startswith("ln") and base.endswith("_b"):
            t = torch.zeros(shape)
        elif base.endswith("_w"):
            bound = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif base.endswith("_b"):
            bound = 1.0 / math.sqrt(fan_in_of_bias[name])
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        else:
            raise KeyError(name)
        out[name] = t.to(dtype)
    return out


def embed(p, slot_class, slot_box):
    """Object-slot embedding: class row gather + box affine + frame-index row."""
    B, T, N = slot_class.shape
    x = F.embedding(slot_class, p["cls_emb"])                       # (B,T,N,d)
    x = x + F.linear(slot_box, p["box_w"], p["box_b"])
    x = x + p["time_emb"][:T][None, :, None, :]
    return x


def temporal_attention(qkv: torch.Tensor, n_heads: int) -> torch.Tensor:
    """Causal attention along T, independently per (clip, slot, head).

    qkv: (B,T,N,3d) with [q | k | v] blocks, head h owning columns h*64..h*64+63
    of each block.  Returns (B,T,N,d).
    """
    B, T, N, d3 = qkv.shape
    d = d3 // 3
    hd = d // n_heads
    q, k, v = qkv.split(d, dim=-1)

    def heads(t):  # (B,T,N,d) -> (B,N,H,T,hd)
        return t.reshape(B, T, N, n_heads, hd).permute(0, 2, 3, 1, 4)

    q, k, v = heads(q), heads(k), heads(v)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)        # (B,N,H,T,T)
    causal = torch.ones(T, T, dtype=torch.bool).tril()
    s = s.masked_fill(~causal, float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v)                                          # (B,N,H,T,hd)
    return o.permute(0, 3, 1, 2, 4).reshape(B, T, N, d)


def clip_attention(qkv: torch.Tensor, n_heads: int, valid: torch.Tensor = None) -> torch.Tensor:
    """Block-causal attention over ALL T*N tokens of a clip, per (clip, head) - the "per-clip tile" reading of
    BASELINE.json's temporal encoder (SELF-ORACLE: the reference has no attention at all; option attention="clip").

    Token (t, n) attends to every token (t', n') of the same clip with t' <= t: all slots of its own and of every
    earlier frame (slots exchange information, the future stays hidden).  Padded slots (valid == 0, variable-N
    batches) are never attended to - except by themselves, so that every softmax row has a key.
    qkv: (B,T,N,3d), valid: (B,T,N) 0/1 floats or None.  Returns (B,T,N,d)."""
    B, T, N, d3 = qkv.shape
    d = d3 // 3
    hd = d // n_heads
    S = T * N
    q, k, v = qkv.split(d, dim=-1)

    def heads(t):  # (B,T,N,d) -> (B,H,S,hd), tokens frame-major: s = t*N + n
        return t.reshape(B, S, n_heads, hd).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)        # (B,H,S,S)
    frame = torch.arange(S) // N
    allowed = (frame[None, :] <= frame[:, None])[None, None]        # key frame <= query frame
    if valid is not None:
        allowed = allowed & (valid.reshape(B, 1, 1, S) > 0)
    allowed = allowed | torch.eye(S, dtype=torch.bool)[None, None]
    s = s.masked_fill(~allowed, float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v)                                          # (B,H,S,hd)
    return o.permute(0, 2, 1, 3).reshape(B, T, N, d)


def encoder_layer(p, pre: str, x: torch.Tensor, n_heads: int, attention: str = "slot", valid=None) -> torch.Tensor:
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), p[pre + "ln1_g"], p[pre + "ln1_b"], LN_EPS)
    qkv = F.linear(h, p[pre + "qkv_w"], p[pre + "qkv_b"])
    o = clip_attention(qkv, n_heads, valid) if attention == "clip" else temporal_attention(qkv, n_heads)
    x = x + F.linear(o, p[pre + "proj_w"], p[pre + "proj_b"])
    h = F.layer_norm(x, (d,), p[pre + "ln2_g"], p[pre + "ln2_b"], LN_EPS)
    u = F.linear(h, p[pre + "ff1_w"], p[pre + "ff1_b"])
    x = x + F.linear(F.gelu(u), p[pre + "ff2_w"], p[pre + "ff2_b"])
    return x


def forward(p, slot_class, slot_box, n_layers: int, n_classes: int = 20, attention: str = "slot", valid=None):
    """Returns (class logits (B,T,N,C), raw box outputs (B,T,N,4)).  attention = "slot" (causal along T per slot, the
    default) or "clip" (block-causal over all slots of a clip; `valid` masks padded slots as keys)."""
    x = embed(p, slot_class, slot_box)
    d = x.shape[-1]
    n_heads = d // HEAD_DIM
    for l in range(n_layers):
        x = encoder_layer(p, "l%d." % l, x, n_heads, attention, valid)
    x = F.layer_norm(x, (d,), p["lnf_g"], p["lnf_b"], LN_EPS)
    out = F.linear(x, p["head_w"], p["head_b"])
    return out[..., :n_classes], out[..., n_classes:]


def box_iou_cxcywh(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    ax1, ay1 = a[..., 0] - a[..., 2] / 2, a[..., 1] - a[..., 3] / 2
    ax2, ay2 = a[..., 0] + a[..., 2] / 2, a[..., 1] + a[..., 3] / 2
    bx1, by1 = b[..., 0] - b[..., 2] / 2, b[..., 1] - b[..., 3] / 2
    bx2, by2 = b[..., 0] + b[..., 2] / 2, b[..., 1] + b[..., 3] / 2
    iw = (torch.minimum(ax2, bx2) - torch.maximum(ax1, bx1)).clamp(min=0)
    ih = (torch.minimum(ay2, by2) - torch.maximum(ay1, by1)).clamp(min=0)
    inter = iw * ih
    union = a[..., 2] * a[..., 3] + b[..., 2] * b[..., 3] - inter
    return inter / (union + IOU_EPS)


def losses(logits, box_raw, tgt_class, tgt_box, valid):
    """(total, smooth_l1, iou_loss, ce), each a mean over the valid slots."""
    cnt = valid.sum().clamp(min=1.0)
    box = torch.sigmoid(box_raw)
    sl1 = F.smooth_l1_loss(box, tgt_box, beta=SMOOTH_L1_BETA, reduction="none").sum(-1)
    l_reg = (sl1 * valid).sum() / (4.0 * cnt)
    l_iou = ((1.0 - box_iou_cxcywh(box, tgt_box)) * valid).sum() / cnt
    C = logits.shape[-1]
    ce = F.cross_entropy(logits.reshape(-1, C), tgt_class.reshape(-1), reduction="none").reshape(valid.shape)
    l_ce = (ce * valid).sum() / cnt
    total = W_REG * l_reg + W_IOU * l_iou + W_CE * l_ce
    return total, l_reg, l_iou, l_ce


def loss_and_grads(p, batch, n_layers: int, n_classes: int = 20, attention: str = "slot"):
    """One forward + backward.  Returns (loss parts tuple of floats, {name: grad})."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logits, box_raw = forward(q, batch["slot_class"], batch["slot_box"], n_layers, n_classes, attention,
                              batch["valid"] if attention == "clip" else None)
    parts = losses(logits, box_raw, batch["tgt_class"], batch["tgt_box"], batch["valid"])
    parts[0].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in q.items()}
    return tuple(float(x.detach()) for x in parts), grads


def adam_step(p, g, m, v, step: int, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor arithmetic (no weight decay, no amsgrad).

    Restates what reference src/trainer.py:83,258 runs:  step is 1-based.
    """
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def synthetic_batch(B: int, T: int, N: int, n_classes: int = 20, seed: int = 1024,
                    variable_n: bool = False, min_valid: int = 8) -> Dict[str, torch.Tensor]:
    """SURVEY.md section 8d Spec N generator: T+1 frames drawn, inputs = frames 0..T-1,
    targets = frames 1..T.  valid is all ones unless variable_n (per-clip valid slot
    count ~ U{min_valid..N})."""
    g = torch.Generator().manual_seed(seed)
    cls = torch.randint(0, n_classes, (B, T + 1, N), generator=g, dtype=torch.int64)
    box = torch.rand((B, T + 1, N, 4), generator=g, dtype=torch.float32) * 0.9 + 0.05
    # keep boxes inside the unit square: w,h <= 2*min(c, 1-c)
    c = box[..., :2]
    wh = torch.minimum(box[..., 2:], 2 * torch.minimum(c, 1 - c))
    box = torch.cat([c, wh], dim=-1).contiguous()
    valid = torch.ones((B, T, N), dtype=torch.float32)
    if variable_n:
        lo = min(min_valid, N)
        nv = torch.randint(lo, N + 1, (B,), generator=g)
        valid = (torch.arange(N)[None, None, :] < nv[:, None, None]).float().expand(B, T, N).contiguous()
        pad = valid == 0
        cls_in = cls[:, :T].clone()
        cls_in[pad] = n_classes                       # reserved id for padded slots
        cls = torch.cat([cls_in, cls[:, T:]], dim=1)
    return {"slot_class": cls[:, :T].contiguous(), "slot_box": box[:, :T].contiguous(),
            "tgt_class": cls[:, 1:].clamp(max=n_classes - 1).contiguous(),
            "tgt_box": box[:, 1:].contiguous(), "valid": valid}
