"""GPU parity tests, one op at a time, through the C ABI (ctypes -> libvlg_hip.so).

Checker: oracle/layout_spec.py (torch-CPU arithmetic + autograd).  The layout-token ops have
no reference counterpart, so these are SELF-ORACLE parity tests; the
tolerance is BASELINE.json's 1e-4 (fp32) unless a test states otherwise.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import layout_spec as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from vlg import hip
    hip.load()
    return hip


def stream():
    return torch.cuda.current_stream().cuda_stream


def reduce_slabs(H, slabs, stride, n, length, dev):
    dst = torch.empty(length, device=dev)
    H.call("vlg_reduce_slabs", slabs.data_ptr(), stride, n, dst.data_ptr(), length, stream())
    return dst


# ------------------------------------------------------------------------ per-clip attention (option attention = "clip")
@pytest.mark.parametrize("B,T,N,d,masked", [(2, 4, 8, 64, False), (2, 4, 8, 64, True), (3, 8, 16, 128, True), (2, 16, 24, 64, True),
                                              (1, 16, 64, 256, False), (2, 32, 5, 64, True), (1, 4, 40, 64, False),
                                              (1, 32, 64, 128, True)])       # BASELINE configs[3] clip: 2 048 tokens, 16 query blocks
def test_clip_attention_fwd_bwd(H, dev, B, T, N, d, masked):
    """vlg_attention_clip_fwd / _bwd (block-causal attention over all T*N tokens of a clip, fp32 MFMA, flash-style) against the
    CPU specification oracle.layout_spec.clip_attention and its autograd (SELF-ORACLE): tiles that straddle frames (N = 5, 24,
    40: element-wise masks), padded slots as keys (valid == 0: never attended to, except by themselves), several query blocks
    (T*N = 1024), NaN-prefilled outputs."""
    torch.manual_seed(B * 1000 + T * N)
    heads = d // 64
    qkv = torch.randn(B, T, N, 3 * d) * 0.7
    valid = (torch.rand(B, T, N) > 0.3).float() if masked else None
    q = qkv.clone().requires_grad_(True)
    want = O.clip_attention(q, heads, valid)
    gy = torch.randn(B, T, N, d)
    want.backward(gy)
    M, S = B * T * N, T * N
    to_rows = lambda t: t.permute(0, 2, 1, 3).contiguous().view(M, t.shape[-1])       # (B,T,N,.) -> internal rows (b, n, t)
    qd, gd = to_rows(qkv).to(dev), to_rows(gy).to(dev)
    vd = valid.to(dev) if masked else None
    out = torch.full((M, d), float("nan"), device=dev)
    lse = torch.full((B * heads * S,), float("nan"), device=dev)
    H.call("vlg_attention_clip_fwd", qd.data_ptr(), H.ptr(vd), out.data_ptr(), lse.data_ptr(), B, T, N, d, stream())
    got = out.view(B, N, T, d).permute(0, 2, 1, 3)
    assert_close(got, want.detach(), rtol=1e-4, atol=2e-5, what="clip attention fwd")
    dqkv = torch.full((M, 3 * d), float("nan"), device=dev)
    delta = torch.full((B * heads * S,), float("nan"), device=dev)
    H.call("vlg_attention_clip_bwd", qd.data_ptr(), H.ptr(vd), out.data_ptr(), gd.data_ptr(), lse.data_ptr(), delta.data_ptr(),
           dqkv.data_ptr(), B, T, N, d, stream())
    gq = dqkv.view(B, N, T, 3 * d).permute(0, 2, 1, 3)
    scale = float(q.grad.abs().max())
    assert_close(gq.cpu() / scale, q.grad / scale, rtol=1e-4, atol=2e-5, what="clip attention bwd (dq | dk | dv)")
    # shapes the tiles cannot hold are refused, not mangled
    assert H.load().vlg_attention_clip_fwd(qd.data_ptr(), 0, out.data_ptr(), lse.data_ptr(), 1, 3, 5, 64, stream()) == 1001


# ------------------------------------------------------------------------ embedding
@pytest.mark.parametrize("B,T,N,d", [(2, 4, 8, 64), (3, 16, 5, 256), (2, 32, 4, 512),
                                     # widths whose d/4-lane groups straddle 64-lane waves: 48, 96, 80, 112 lanes
                                     (3, 16, 7, 192), (2, 16, 9, 384), (2, 8, 6, 320), (2, 4, 5, 448), (5, 32, 3, 64)])
def test_embed_fwd_bwd(H, dev, B, T, N, d):
    torch.manual_seed(0)
    vocab = 21
    p = {"cls_emb": torch.randn(vocab, d), "box_w": torch.randn(d, 4) * 0.5, "box_b": torch.randn(d) * 0.1,
         "time_emb": torch.randn(T, d)}
    cls = torch.randint(0, vocab, (B, T, N))
    box = torch.rand(B, T, N, 4)
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    want = O.embed(q, cls, box)                                  # (B,T,N,d)
    gy = torch.randn(B, T, N, d)
    want.backward(gy)
    pd = {k: v.to(dev) for k, v in p.items()}
    x = torch.empty(B * N * T, d, device=dev)
    clsd, boxd = cls.to(dev), box.to(dev)           # keep references: data_ptr() of a temporary dangles
    H.call("vlg_embed_fwd", clsd.data_ptr(), boxd.data_ptr(), pd["cls_emb"].data_ptr(),
           pd["box_w"].data_ptr(), pd["box_b"].data_ptr(), pd["time_emb"].data_ptr(), x.data_ptr(),
           B, T, N, d, vocab, stream())
    got = x.view(B, N, T, d).permute(0, 2, 1, 3)
    assert_close(got, want, what="embed fwd")
    # backward
    dx = gy.permute(0, 2, 1, 3).contiguous().view(B * N * T, d).to(dev)
    L = vocab * d + d * 4 + d + T * d
    ns = H.load().vlg_embed_bwd_slabs_for(B, T, N, d, vocab)
    assert 1 <= ns <= H.load().vlg_embed_bwd_slabs()
    slabs = torch.empty(ns * L, device=dev)
    H.call("vlg_embed_bwd", dx.data_ptr(), clsd.data_ptr(), boxd.data_ptr(), slabs.data_ptr(), L, slabs.numel(), B, T, N, d, vocab,
           stream())
    g = reduce_slabs(H, slabs, L, ns, L, dev)
    o = 0
    for name, n in (("cls_emb", vocab * d), ("box_w", d * 4), ("box_b", d), ("time_emb", T * d)):
        assert_close(g[o:o + n].view(p[name].shape), q[name].grad, rtol=1e-4, atol=1e-4, what="embed d" + name)
        o += n


def test_embed_class_table_matches_reference_embedding(H, dev):
    """The class-table term of vlg_embed_fwd / _bwd against the reference's own nn.Embedding(30, dim) lookup and its
    gradient (reference src/models/simple.py:23,41-42; golden written by oracle/make_golden_embedding.py): box and frame
    terms switched off (zero weights), reserved id 29 included."""
    import numpy as np
    from conftest import GOLDEN
    z = np.load(GOLDEN + "/embedding_simple.npz")
    # the kernels take T in {4, 8, 16, 32}: read the 84 stored ids as (B,T,N) = (3,4,7) - a row gather does not care
    B, T, N = 3, 4, 7
    ids = torch.from_numpy(z["ids"]).reshape(B, T, N).contiguous().to(dev)
    vocab, d = z["table"].shape
    table = torch.from_numpy(z["table"]).to(dev)
    zeros = torch.zeros(d * 4 + d + T * d, device=dev)
    box = torch.zeros(B, T, N, 4, device=dev)
    x = torch.empty(B * N * T, d, device=dev)
    H.call("vlg_embed_fwd", ids.data_ptr(), box.data_ptr(), table.data_ptr(), zeros.data_ptr(), zeros[d * 4:].data_ptr(),
           zeros[d * 5:].data_ptr(), x.data_ptr(), B, T, N, d, vocab, stream())
    got = x.view(B, N, T, d).permute(0, 2, 1, 3).cpu()
    assert torch.equal(got, torch.from_numpy(z["out"]).reshape(B, T, N, d))  # a row gather: exact
    dx = torch.from_numpy(z["r"]).reshape(B, T, N, d).permute(0, 2, 1, 3).contiguous().view(B * N * T, d).to(dev)
    L = vocab * d + d * 4 + d + T * d
    ns = H.load().vlg_embed_bwd_slabs_for(B, T, N, d, vocab)
    slabs = torch.empty(ns * L, device=dev)
    H.call("vlg_embed_bwd", dx.data_ptr(), ids.data_ptr(), box.data_ptr(), slabs.data_ptr(), L, slabs.numel(), B, T, N, d, vocab,
           stream())
    g = reduce_slabs(H, slabs, L, ns, L, dev)
    assert_close(g[:vocab * d].view(vocab, d), torch.from_numpy(z["dtable"]), rtol=1e-5, atol=1e-6, what="class-table gradient")


# ----------------------------------------------------------------------- layer-norm
@pytest.mark.parametrize("rows,d", [(7, 64), (1000, 256), (1003, 256), (513, 512), (64, 128), (40, 1024), (3, 192)])
def test_layernorm_fwd_bwd(H, dev, rows, d):
    torch.manual_seed(1)
    x = (torch.randn(rows, d) * 2 + 0.5).requires_grad_(True)
    g = (torch.rand(d) + 0.5).requires_grad_(True)
    b = torch.randn(d).requires_grad_(True)
    y = F.layer_norm(x, (d,), g, b, 1e-5)
    dy = torch.randn(rows, d)
    y.backward(dy)
    xd, gd, bd = x.detach().to(dev), g.detach().to(dev), b.detach().to(dev)
    yd = torch.empty(rows, d, device=dev)
    mean = torch.empty(rows, device=dev)
    rstd = torch.empty(rows, device=dev)
    H.call("vlg_layernorm_fwd", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), yd.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, d, 1e-5, stream())
    assert_close(yd, y, what="ln fwd")
    assert_close(mean, x.detach().mean(-1), what="ln mean")
    ns = H.load().vlg_layernorm_bwd_slabs(rows)
    slabs = torch.empty(ns * 2 * d, device=dev)
    res = torch.randn(rows, d)
    dres = res.to(dev)
    dyd = dy.to(dev)
    H.call("vlg_layernorm_bwd", dyd.data_ptr(), xd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
           dres.data_ptr(), dres.data_ptr(), slabs.data_ptr(), 2 * d, slabs.numel(), rows, d, stream())      # in place
    assert_close(dres, res + x.grad, rtol=1e-4, atol=2e-5, what="ln dx (+residual, in place)")
    gb = reduce_slabs(H, slabs, 2 * d, ns, 2 * d, dev)
    assert_close(gb[:d], g.grad, rtol=1e-4, atol=1e-4, what="ln dgamma")
    assert_close(gb[d:], b.grad, rtol=1e-4, atol=1e-4, what="ln dbeta")
    # without residual
    dx2 = torch.empty(rows, d, device=dev)
    H.call("vlg_layernorm_bwd", dyd.data_ptr(), xd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
           0, dx2.data_ptr(), slabs.data_ptr(), 2 * d, slabs.numel(), rows, d, stream())
    assert_close(dx2, x.grad, rtol=1e-4, atol=2e-5, what="ln dx")


# ----------------------------------------------------------------------------- GEMM
def _exact_gelu(u):
    return F.gelu(u)


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (200, 768, 256), (384, 256, 1024), (128, 64, 64), (1000, 24, 256)])
def test_linear_fwd_bias(H, dev, M, N, K):
    torch.manual_seed(2)
    a, w, b = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    want = F.linear(a.double(), w.double(), b.double()).float()
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    c = torch.full((M, N), float("nan"), device=dev)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, 0, M, N, K,
           H.EPI_BIAS, stream())
    assert_close(c, want, rtol=1e-4, atol=1e-5, what="linear fwd bias")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 1024, 256)])
def test_linear_fwd_gelu_and_resid(H, dev, M, N, K):
    torch.manual_seed(3)
    a, w, b, r = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N), torch.randn(M, N)
    pre = F.linear(a.double(), w.double(), b.double())
    ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
    c = torch.empty(M, N, device=dev)
    u = torch.empty(M, N, device=dev)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, u.data_ptr(),
           M, N, K, H.EPI_BIAS | H.EPI_GELU, stream())
    assert_close(u, pre.float(), what="ffn pre-activation")
    assert_close(c, F.gelu(pre).float(), what="ffn gelu")
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, rd.data_ptr(), 0,
           M, N, K, H.EPI_BIAS | H.EPI_RESID, stream())
    assert_close(c, (pre + r.double()).float(), what="linear + residual")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 1024, 256), (520, 384, 96)])
def test_linear_gelu_grad_saved_and_mul(H, dev, M, N, K):
    """VLG_EPI_GELU_GRAD: the first FFN projection stores gelu'(pre) instead of pre;  VLG_EPI_MUL: the data gradient of the
    second projection multiplies by it - together they must give what VLG_EPI_GELU / VLG_EPI_DGELU give (interior and
    edge tiles, even and odd K-tile counts)."""
    torch.manual_seed(11)
    a, w, b = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K) * 1.5, torch.randn(N)
    pre = F.linear(a.double(), w.double(), b.double())
    uu = pre.clone().requires_grad_(True)
    F.gelu(uu).sum().backward()
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    c = torch.full((M, N), float("nan"), device=dev)
    dsave = torch.full((M, N), float("nan"), device=dev)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, dsave.data_ptr(),
           M, N, K, H.EPI_BIAS | H.EPI_GELU | H.EPI_GELU_GRAD, stream())
    assert_close(c, F.gelu(pre).float(), what="ffn gelu (grad saved)")
    assert_close(dsave, uu.grad.float(), rtol=1e-4, atol=1e-5, what="saved gelu'")
    # data gradient of the second projection: dU = (dY . W2) * saved
    K2 = 128
    dy, w2 = torch.randn(M, K2), torch.randn(K2, N) / math.sqrt(K2)
    dyd, w2d = dy.to(dev), w2.to(dev)
    du = torch.full((M, N), float("nan"), device=dev)
    H.call("vlg_linear_dgrad", dyd.data_ptr(), K2, w2d.data_ptr(), N, du.data_ptr(), N, dsave.data_ptr(), M, K2, N,
           H.EPI_MUL, stream())
    want = (dy.double() @ w2.double()) * uu.grad
    assert_close(du, want.float(), rtol=1e-4, atol=1e-5, what="dgrad * saved gelu'")
    # the bf16 MFMA kernel takes the same flags (fp32 tensors here: operands rounded to bf16 on their way to LDS)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, dsave.data_ptr(),
           M, N, K, H.EPI_BIAS | H.EPI_GELU | H.EPI_GELU_GRAD | H.EPI_BF16, stream())
    assert_close(c, F.gelu(pre).float(), rtol=2e-2, atol=2e-2, what="ffn gelu (grad saved, bf16 MFMA)")
    assert_close(dsave, uu.grad.float(), rtol=2e-2, atol=2e-2, what="saved gelu' (bf16 MFMA)")
    H.call("vlg_linear_dgrad", dyd.data_ptr(), K2, w2d.data_ptr(), N, du.data_ptr(), N, dsave.data_ptr(), M, K2, N,
           H.EPI_MUL | H.EPI_BF16, stream())
    assert_close(du, ((dy.double() @ w2.double()) * dsave.double().cpu()).float(), rtol=2e-2, atol=3e-2, what="dgrad * saved (bf16 MFMA)")
    # the split-bf16 fp32 mode refuses the flags instead of ignoring them
    lib = H.load()
    assert lib.vlg_linear_fwd(ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, dsave.data_ptr(),
                              M, N, K, H.EPI_BIAS | H.EPI_GELU | H.EPI_GELU_GRAD | H.EPI_SPLIT3, stream()) == 1001
    assert lib.vlg_linear_dgrad(dyd.data_ptr(), K2, w2d.data_ptr(), N, du.data_ptr(), N, dsave.data_ptr(), M, K2, N,
                                H.EPI_MUL | H.EPI_SPLIT3, stream()) == 1001
    assert lib.vlg_linear_dgrad(dyd.data_ptr(), K2, w2d.data_ptr(), N, du.data_ptr(), N, 0, M, K2, N, H.EPI_MUL, stream()) == 1001


@pytest.mark.parametrize("M,N,K,mul", [(4096, 256, 1024, True), (4096, 1024, 256, False), (4096, 256, 256, False), (4096, 768, 256, False),
                                       (2048, 256, 256, False), (1000, 192, 320, True), (32768, 256, 256, False)])
def test_linear_dgrad_wgrad_is_the_two_calls(H, dev, M, N, K, mul):
    """vlg_linear_dgrad_wgrad = vlg_linear_wgrad + vlg_linear_dgrad of one projection, bit for bit - as one launch on 64x64
    tiles at few tokens (the strong-scaling shard: M = 4 096; ragged M / N / K included), as two launches elsewhere -
    and right against fp64."""
    lib = H.load()
    torch.manual_seed(M + N)
    dy, w, x = torch.randn(M, N), torch.randn(N, K) / math.sqrt(K), torch.randn(M, K)
    aux = torch.randn(M, K) if mul else None
    dyd, wd, xd = dy.to(dev), w.to(dev), x.to(dev)
    auxd = aux.to(dev) if mul else None
    epi = H.EPI_MUL if mul else H.EPI_NONE
    ns = lib.vlg_linear_wgrad_slabs_for(M, N, K, 0)
    L = N * K + N
    out = []
    for fused in (False, True):
        slabs = torch.full((ns * L,), float("nan"), device=dev)
        dx = torch.full((M, K), float("nan"), device=dev)
        if fused:
            H.call("vlg_linear_dgrad_wgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, H.ptr(auxd), xd.data_ptr(), K,
                   slabs.data_ptr(), L, slabs.numel(), M, N, K, epi, 0, 0, stream())
        else:
            H.call("vlg_linear_wgrad", dyd.data_ptr(), N, xd.data_ptr(), K, slabs.data_ptr(), L, slabs.numel(), M, N, K, 0, stream())
            H.call("vlg_linear_dgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, H.ptr(auxd), M, N, K, epi, stream())
        out.append((dx, reduce_slabs(H, slabs, L, ns, L, dev)))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    want_dx = dy.double() @ w.double()
    if mul:
        want_dx = want_dx * aux.double()
    assert_close(out[1][0], want_dx.float(), rtol=1e-4, atol=1e-4, what="paired dgrad")
    g = out[1][1].cpu()
    assert_close(g[:N * K].view(N, K), (dy.double().t() @ x.double()).float(), rtol=1e-4, atol=2e-3, what="paired wgrad")
    assert_close(g[N * K:], dy.double().sum(0).float(), rtol=1e-4, atol=1e-3, what="paired bias gradient")


@pytest.mark.parametrize("M,N,K,mul,dy16", [(4096, 256, 1024, True, False), (4096, 1024, 256, False, True), (2048, 256, 256, False, False),
                                            (32768, 768, 256, False, True)])
def test_linear_dgrad_wgrad_bf16_storage_is_the_two_calls(H, dev, M, N, K, mul, dy16):
    """The bf16-storage step's pairs (bf16 W / X / dX, dY bf16 or fp32; csrc/gemm_bf16.hip:gemm16_pair_kernel): one launch,
    bit for bit the two launches."""
    lib = H.load()
    torch.manual_seed(M + N + K)
    bf = torch.bfloat16
    dy = torch.randn(M, N).to(bf if dy16 else torch.float32).to(dev)
    w, x = (torch.randn(N, K) / math.sqrt(K)).to(bf).to(dev), torch.randn(M, K).to(bf).to(dev)
    aux = torch.randn(M, K).to(bf).to(dev) if mul else None
    FL = H.EPI_BF16 | (H.EPI_A_BF16 if dy16 else 0) | H.EPI_B_BF16
    epi = H.EPI_MUL if mul else H.EPI_NONE
    ns = lib.vlg_linear_wgrad_slabs_for(M, N, K, H.EPI_BF16)
    L = N * K + N
    out = []
    for fused in (False, True):
        slabs = torch.full((ns * L,), float("nan"), device=dev)
        dx = torch.full((M, K), float("nan"), device=dev, dtype=bf)
        if fused:
            H.call("vlg_linear_dgrad_wgrad", dy.data_ptr(), N, w.data_ptr(), K, dx.data_ptr(), K, H.ptr(aux), x.data_ptr(), K,
                   slabs.data_ptr(), L, slabs.numel(), M, N, K, epi | FL | H.EPI_OUT_BF16, 0, 0, stream())
        else:
            H.call("vlg_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, slabs.data_ptr(), L, slabs.numel(), M, N, K, FL, stream())
            H.call("vlg_linear_dgrad", dy.data_ptr(), N, w.data_ptr(), K, dx.data_ptr(), K, H.ptr(aux), M, N, K, epi | FL | H.EPI_OUT_BF16, stream())
        out.append((dx, reduce_slabs(H, slabs, L, ns, L, dev)))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert bool(torch.isfinite(out[1][0].float()).all()) and bool(torch.isfinite(out[1][1]).all())


def test_linear_chained_tiles(H, dev):
    """Multi-round launches: a block computes a run of N tiles back to back (the K loop continues into the next tile).
    1024 tiles here -> runs of two; forward with bias, forward with GELU + saved derivative, data gradient with the
    multiply epilogue.  (The ping-pong variant of this path is a diagnostic-build option: csrc/gemm.hip, VLG_DIAG.)"""
    M, N, K = 8192, 2048, 128
    if True:
        torch.manual_seed(21)
        a, w, b = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
        pre = F.linear(a.double(), w.double(), b.double())
        ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
        c = torch.full((M, N), float("nan"), device=dev)
        H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, 0, M, N, K, H.EPI_BIAS, stream())
        assert_close(c, pre.float(), rtol=1e-4, atol=1e-5, what="chained fwd bias")
        dsave = torch.full((M, N), float("nan"), device=dev)
        H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, dsave.data_ptr(), M, N, K,
               H.EPI_BIAS | H.EPI_GELU | H.EPI_GELU_GRAD, stream())
        uu = pre.clone().requires_grad_(True)
        F.gelu(uu).sum().backward()
        assert_close(c, F.gelu(pre).float(), what="chained gelu")
        assert_close(dsave, uu.grad.float(), rtol=1e-4, atol=1e-5, what="chained saved gelu'")
        # data gradient into a wide output: dX[M, N] = dY[M, K] . W2[K, N], times the saved derivative
        dy, w2 = torch.randn(M, K), torch.randn(K, N) / math.sqrt(K)
        dyd, w2d = dy.to(dev), w2.to(dev)
        dx = torch.full((M, N), float("nan"), device=dev)
        H.call("vlg_linear_dgrad", dyd.data_ptr(), K, w2d.data_ptr(), N, dx.data_ptr(), N, dsave.data_ptr(), M, K, N, H.EPI_MUL, stream())
        assert_close(dx, ((dy.double() @ w2.double()) * uu.grad).float(), rtol=1e-4, atol=1e-5, what="chained dgrad * saved")


@pytest.mark.parametrize("M,N,K", [(256, 256, 1024), (300, 128, 512), (1000, 256, 96)])
def test_linear_gelu_on_load(H, dev, M, N, K):
    """VLG_EPI_ACT_GELU: the activation operand holds PRE-activations u and gelu(u) is formed while the tile is staged -
    forward  C = gelu(u) . W^T + b + resid  and weight gradient  dW = dY^T . gelu(u)  (the FFN's second projection)."""
    torch.manual_seed(7)
    u, w, b, r = torch.randn(M, K) * 1.5, torch.randn(N, K) / math.sqrt(K), torch.randn(N), torch.randn(M, N)
    dy = torch.randn(M, N)
    g = F.gelu(u.double())
    ud, wd, bd, rd, dyd = u.to(dev), w.to(dev), b.to(dev), r.to(dev), dy.to(dev)
    c = torch.full((M, N), float("nan"), device=dev)
    H.call("vlg_linear_fwd", ud.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, rd.data_ptr(), 0,
           M, N, K, H.EPI_BIAS | H.EPI_RESID | H.EPI_ACT_GELU, stream())
    assert_close(c, (F.linear(g, w.double(), b.double()) + r.double()).float(), rtol=1e-4, atol=1e-5, what="gelu-on-load fwd")
    assert torch.equal(ud.cpu(), u)                                      # the stored pre-activation is untouched
    ns = H.load().vlg_linear_wgrad_slabs(M, N, K)
    stride = N * K + N
    slabs = torch.full((ns * stride,), float("nan"), device=dev)
    H.call("vlg_linear_wgrad", dyd.data_ptr(), N, ud.data_ptr(), K, slabs.data_ptr(), stride, slabs.numel(), M, N, K,
           H.EPI_ACT_GELU, stream())
    gr = reduce_slabs(H, slabs, stride, ns, stride, dev)
    sc = math.sqrt(M)
    assert_close(gr[:N * K].view(N, K) / sc, (dy.double().t() @ g).float() / sc, rtol=1e-4, atol=1e-5, what="gelu-on-load wgrad")
    assert_close(gr[N * K:] / sc, dy.double().sum(0).float() / sc, rtol=1e-4, atol=1e-5, what="bias grad")
    # only the native fp32 path implements it: the other modes refuse instead of silently ignoring the flag
    rc = H.load().vlg_linear_fwd(ud.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, rd.data_ptr(), 0,
                                 M, N, K, H.EPI_BIAS | H.EPI_RESID | H.EPI_ACT_GELU | H.EPI_BF16, stream())
    assert rc == 1001


@pytest.mark.parametrize("M,N,K", [(256, 768, 256), (200, 256, 1024), (384, 1024, 256), (500, 24, 256), (128, 64, 64)])
def test_linear_dgrad_wgrad(H, dev, M, N, K):
    torch.manual_seed(4)
    dy, w, x = torch.randn(M, N), torch.randn(N, K) / math.sqrt(N), torch.randn(M, K)
    dyd, wd, xd = dy.to(dev), w.to(dev), x.to(dev)
    dx = torch.empty(M, K, device=dev)
    H.call("vlg_linear_dgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, 0, M, N, K, H.EPI_NONE, stream())
    assert_close(dx, (dy.double() @ w.double()).float(), rtol=1e-4, atol=1e-5, what="dgrad")
    u = torch.randn(M, K)
    ud = u.to(dev)
    H.call("vlg_linear_dgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, ud.data_ptr(), M, N, K,
           H.EPI_DGELU, stream())
    uu = u.double().requires_grad_(True)
    F.gelu(uu).backward(dy.double() @ w.double())
    assert_close(dx, uu.grad.float(), rtol=1e-4, atol=1e-5, what="dgrad * gelu'")
    ns = H.load().vlg_linear_wgrad_slabs(M, N, K)
    stride = N * K + N
    slabs = torch.full((ns * stride,), float("nan"), device=dev)
    H.call("vlg_linear_wgrad", dyd.data_ptr(), N, xd.data_ptr(), K, slabs.data_ptr(), stride, slabs.numel(), M, N, K, 0, stream())
    g = reduce_slabs(H, slabs, stride, ns, stride, dev)
    sc = math.sqrt(M)
    assert_close(g[:N * K].view(N, K) / sc, (dy.double().t() @ x.double()).float() / sc, rtol=1e-4, atol=1e-5, what="wgrad")
    assert_close(g[N * K:] / sc, dy.double().sum(0).float() / sc, rtol=1e-4, atol=1e-5, what="bias grad")


# ------------------------------------------------------------------------ attention
@pytest.mark.parametrize("T,d,n_seq", [(4, 64, 5), (8, 128, 3), (16, 256, 6), (32, 256, 3), (16, 512, 2), (32, 512, 5), (32, 64, 7)])
def test_attention_fwd_bwd(H, dev, T, d, n_seq):
    torch.manual_seed(5)
    Hn = d // 64
    qkv = torch.randn(1, T, n_seq, 3 * d, requires_grad=True)        # (B=1,T,N=n_seq,3d)
    want = O.temporal_attention(qkv, Hn)                              # (1,T,n_seq,d)
    do = torch.randn(1, T, n_seq, d)
    want.backward(do)
    to_int = lambda t: t[0].permute(1, 0, 2).contiguous().view(n_seq * T, -1)
    qd = to_int(qkv.detach()).to(dev)
    o = torch.full((n_seq * T, d), float("nan"), device=dev)
    H.call("vlg_attention_fwd", qd.data_ptr(), o.data_ptr(), n_seq, T, d, stream())
    assert_close(o, to_int(want.detach()), what="attention fwd")
    dod = to_int(do).to(dev)
    dq = torch.full((n_seq * T, 3 * d), float("nan"), device=dev)
    H.call("vlg_attention_bwd", qd.data_ptr(), dod.data_ptr(), dq.data_ptr(), n_seq, T, d, stream())
    assert_close(dq, to_int(qkv.grad), rtol=1e-4, atol=2e-5, what="attention bwd")


# ----------------------------------------------------------------------------- loss
@pytest.mark.parametrize("B,T,N,var", [(2, 4, 8, False), (3, 16, 7, True), (4, 8, 33, True)])
def test_layout_loss(H, dev, B, T, N, var):
    torch.manual_seed(6)
    batch = O.synthetic_batch(B, T, N, seed=6, variable_n=var, min_valid=2)
    logits = (torch.randn(B, T, N, 20) * 2).requires_grad_(True)
    raw = torch.randn(B, T, N, 4).requires_grad_(True)
    parts = O.losses(logits, raw, batch["tgt_class"], batch["tgt_box"], batch["valid"])
    parts[0].backward()
    M = B * T * N
    to_int = lambda t: t.permute(0, 2, 1, 3).contiguous().view(M, -1)
    out = torch.cat([to_int(logits.detach()), to_int(raw.detach())], dim=1).contiguous().to(dev)
    dout = torch.full((M, 24), float("nan"), device=dev)
    loss = torch.zeros(4, device=dev)
    scratch = torch.zeros(H.load().vlg_layout_loss_scratch(), device=dev)
    tc, tb, va = batch["tgt_class"].to(dev), batch["tgt_box"].to(dev), batch["valid"].to(dev)
    H.call("vlg_layout_loss", out.data_ptr(), 24, tc.data_ptr(), tb.data_ptr(), va.data_ptr(), dout.data_ptr(),
           loss.data_ptr(), scratch.data_ptr(), B, T, N, 20, O.SMOOTH_L1_BETA, O.IOU_EPS, O.W_REG, O.W_IOU, O.W_CE,
           stream())
    assert_close(loss, torch.stack([x.detach() for x in parts]), rtol=1e-4, atol=1e-6, what="loss values")
    want = torch.cat([to_int(logits.grad), to_int(raw.grad)], dim=1)
    assert_close(dout, want, rtol=1e-4, atol=1e-7, what="loss grads")


def test_layout_loss_all_masked(H, dev):
    """Edge case: no valid slot at all -> losses 0, gradients 0 (count clamps to 1)."""
    B, T, N, M = 1, 4, 4, 16
    out = torch.randn(M, 24, device=dev)
    dout = torch.full((M, 24), float("nan"), device=dev)
    loss = torch.ones(4, device=dev)
    scratch = torch.zeros(H.load().vlg_layout_loss_scratch(), device=dev)
    tc = torch.zeros(B, T, N, dtype=torch.int64, device=dev)
    tb = torch.rand(B, T, N, 4, device=dev)
    va = torch.zeros(B, T, N, device=dev)
    H.call("vlg_layout_loss", out.data_ptr(), 24, tc.data_ptr(), tb.data_ptr(), va.data_ptr(), dout.data_ptr(),
           loss.data_ptr(), scratch.data_ptr(), B, T, N, 20, 0.1, 1e-7, 40.0, 20.0, 10.0, stream())
    assert float(loss.abs().max()) == 0.0
    assert float(dout.abs().max()) == 0.0


# ----------------------------------------------------------------- reduce and Adam
def test_reduce_slabs(H, dev):
    torch.manual_seed(7)
    for n, L, stride in ((1, 8, 8), (7, 1000, 1024), (41, 4096, 4096)):
        s = torch.randn(n, stride)
        got = reduce_slabs(H, s.to(dev).flatten(), stride, n, L, dev)
        assert_close(got, s[:, :L].double().sum(0).float(), rtol=1e-5, atol=1e-5, what="reduce_slabs")


def test_adam_matches_torch_optim(H, dev):
    """Three Adam(beta1=0.5) steps from a fixed state vs torch.optim.Adam (reference src/trainer.py:83)."""
    torch.manual_seed(8)
    n = 4096 + 8
    p0 = torch.randn(n)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4, betas=(0.5, 0.999))
    p = p0.to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        g = torch.randn(n) * (10.0 ** (step - 2))
        ref.grad = g.clone()
        opt.step()
        gd = (g * 4).to(dev)            # grad_scale 0.25 undoes the x4: models all-reduce SUM over 4 ranks
        H.call("vlg_adam_step", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, step, 2e-4, 0.5, 0.999,
               1e-8, 0.25, stream())
        assert_close(p, ref.detach(), rtol=1e-6, atol=1e-7, what="adam step %d" % step)


def test_bad_arguments_raise(H, dev):
    x = torch.zeros(64, 64, device=dev)
    with pytest.raises(H.HipError):
        H.call("vlg_attention_fwd", x.data_ptr(), x.data_ptr(), 1, 5, 64, stream())          # T not supported
    with pytest.raises(H.HipError):
        H.call("vlg_layernorm_fwd", x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
               x.data_ptr(), 64, 100, 1e-5, stream())                                            # d not supported
    with pytest.raises(H.HipError):
        H.call("vlg_linear_fwd", x.data_ptr() + 4, 64, x.data_ptr(), 64, x.data_ptr(), x.data_ptr(), 64, 0, 0,
               8, 64, 64, H.EPI_BIAS, stream())                                                  # misaligned A


# -------------------------------------------------- bf16 activation storage (BASELINE.json configs[2])
# Arithmetic is the fp32 kernels' (fp32 accumulate / statistics / softmax); only the HBM element type of the
# projection-side activations changes.  So each kernel must reproduce, to fp32 rounding, the fp32 result computed
# from the SAME bf16-representable inputs, up to one final round-to-nearest-even of a bf16 output (2^-8 relative).
BF = torch.bfloat16
BF_OUT = dict(rtol=2.0 ** -7, atol=1e-6)            # one bf16 rounding of the output (+ fp32 noise around a tie)


def _bf(t):
    return t.to(BF).float()                          # bf16-representable fp32 values


@pytest.mark.parametrize("rows,d", [(7, 64), (1000, 256), (65, 512)])
def test_layernorm_bf16_storage(H, dev, rows, d):
    torch.manual_seed(11)
    x = (torch.randn(rows, d) * 2 + 0.5).requires_grad_(True)
    g = (torch.rand(d) + 0.5).requires_grad_(True)
    b = torch.randn(d).requires_grad_(True)
    y = F.layer_norm(x, (d,), g, b, 1e-5)
    dy = _bf(torch.randn(rows, d))
    y.backward(dy)
    xd, gd, bd = x.detach().to(dev), g.detach().to(dev), b.detach().to(dev)
    yd = torch.empty(rows, d, device=dev, dtype=BF)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    H.call("vlg_layernorm_fwd_bf16", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), yd.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, d, 1e-5, stream())
    assert_close(yd.float(), y.detach(), what="ln fwd -> bf16", **BF_OUT)
    ns = H.load().vlg_layernorm_bwd_slabs(rows)
    slabs = torch.empty(ns * 2 * d, device=dev)
    res = torch.randn(rows, d)
    dres, dyd = res.to(dev), dy.to(dev).to(BF)
    H.call("vlg_layernorm_bwd_bf16", dyd.data_ptr(), xd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
           dres.data_ptr(), dres.data_ptr(), slabs.data_ptr(), 2 * d, slabs.numel(), rows, d, stream())
    assert_close(dres, res + x.grad, rtol=1e-4, atol=2e-5, what="ln dx from bf16 dy")
    gb = reduce_slabs(H, slabs, 2 * d, ns, 2 * d, dev)
    assert_close(gb[:d], g.grad, rtol=1e-4, atol=1e-4, what="ln dgamma from bf16 dy")
    assert_close(gb[d:], b.grad, rtol=1e-4, atol=1e-4, what="ln dbeta from bf16 dy")


@pytest.mark.parametrize("T,d,n_seq", [(4, 64, 5), (8, 128, 3), (16, 256, 6), (32, 128, 3)])
def test_attention_bf16_storage(H, dev, T, d, n_seq):
    torch.manual_seed(12)
    qkv = _bf(torch.randn(1, T, n_seq, 3 * d)).requires_grad_(True)
    want = O.temporal_attention(qkv, d // 64)
    do = _bf(torch.randn(1, T, n_seq, d))
    want.backward(do)
    to_int = lambda t: t[0].permute(1, 0, 2).contiguous().view(n_seq * T, -1)
    qd = to_int(qkv.detach()).to(dev).to(BF)
    o = torch.zeros(n_seq * T, d, device=dev, dtype=BF)
    H.call("vlg_attention_fwd_bf16", qd.data_ptr(), o.data_ptr(), n_seq, T, d, stream())
    assert_close(o.float(), to_int(want.detach()), what="attention fwd (bf16 storage)", **BF_OUT)
    dod = to_int(do).to(dev).to(BF)
    dq = torch.zeros(n_seq * T, 3 * d, device=dev, dtype=BF)
    H.call("vlg_attention_bwd_bf16", qd.data_ptr(), dod.data_ptr(), dq.data_ptr(), n_seq, T, d, stream())
    assert_close(dq.float(), to_int(qkv.grad), rtol=2.0 ** -7, atol=1e-5, what="attention bwd (bf16 storage)")


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (200, 768, 256), (333, 256, 1024)])
def test_linear_bf16_storage(H, dev, M, N, K):
    """Every (operand storage, epilogue) combination the bf16 step launches: bf16 activations and the bf16 shadow of the
    weights (vlg_adam_step_bf16 keeps it), fp32 where the residual stream is involved.  The fp64 reference is computed
    from the same bf16-representable values, so outputs differ from it by fp32 accumulation noise and, for bf16
    outputs, one final rounding."""
    torch.manual_seed(13)
    FL = H.EPI_BF16
    a, w, bias = _bf(torch.randn(M, K)), _bf(torch.randn(N, K) / math.sqrt(K)), torch.randn(N)
    ad, wd, bd = a.to(dev).to(BF), w.to(dev).to(BF), bias.to(dev)
    WB = H.EPI_B_BF16                                # the weight operand is the bf16 shadow
    ref = (a.double() @ w.double().t() + bias.double())
    # forward: bias -> bf16 (QKV), bias+gelu -> bf16 pair (FFN1), bias+resid -> fp32 (out-proj / FFN2)
    c = torch.zeros(M, N, device=dev, dtype=BF)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, 0, M, N, K,
           H.EPI_BIAS | FL | H.EPI_A_BF16 | WB | H.EPI_OUT_BF16, stream())
    assert_close(c.float(), ref.float(), what="fwd bias (bf16 in/out)", **BF_OUT)
    u = torch.zeros(M, N, device=dev, dtype=BF)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, u.data_ptr(), M, N, K,
           H.EPI_BIAS | H.EPI_GELU | FL | H.EPI_A_BF16 | WB | H.EPI_OUT_BF16, stream())
    assert_close(u.float(), ref.float(), what="fwd pre-activation (bf16)", **BF_OUT)
    assert_close(c.float(), F.gelu(ref).float(), what="fwd gelu (bf16)", rtol=2.0 ** -7, atol=2e-6)
    resid = torch.randn(M, N)
    rd, c32 = resid.to(dev), torch.zeros(M, N, device=dev)
    H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c32.data_ptr(), N, rd.data_ptr(), 0, M, N, K,
           H.EPI_BIAS | H.EPI_RESID | FL | H.EPI_A_BF16 | WB, stream())
    assert_close(c32, (ref + resid.double()).float(), rtol=1e-4, atol=1e-5, what="fwd bias+resid (bf16 A, fp32 out)")
    # data gradient: dY fp32 or bf16 -> dX bf16; with gelu' of a bf16 pre-activation
    dy = _bf(torch.randn(M, N))
    dref = dy.double() @ w.double()
    for dy_dev, bits, tag in ((dy.to(dev), 0, "fp32 dY"), (dy.to(dev).to(BF), H.EPI_A_BF16, "bf16 dY")):
        dx = torch.zeros(M, K, device=dev, dtype=BF)
        H.call("vlg_linear_dgrad", dy_dev.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, 0, M, N, K,
               FL | bits | WB | H.EPI_OUT_BF16, stream())
        assert_close(dx.float(), dref.float(), what="dgrad -> bf16 (%s)" % tag, rtol=2.0 ** -7, atol=1e-5)
    pre = _bf(torch.randn(M, K))
    pd = pre.to(dev).to(BF)
    dx = torch.zeros(M, K, device=dev, dtype=BF)
    H.call("vlg_linear_dgrad", dy.to(dev).data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, pd.data_ptr(), M, N, K,
           H.EPI_DGELU | FL | WB | H.EPI_OUT_BF16, stream())
    uu = pre.double().requires_grad_(True)
    F.gelu(uu).backward(dref)
    assert_close(dx.float(), uu.grad.float(), what="dgrad * gelu' (bf16 aux/out)", rtol=2.0 ** -7, atol=1e-5)
    # weight gradient: (fp32 | bf16) dY with bf16 X -> fp32 slabs
    x = _bf(torch.randn(M, K))
    xd = x.to(dev).to(BF)
    ns = H.load().vlg_linear_wgrad_slabs_for(M, N, K, H.EPI_BF16)
    stride = N * K + N
    sc = math.sqrt(M)
    for dy_dev, bits, tag in ((dy.to(dev), 0, "fp32 dY"), (dy.to(dev).to(BF), H.EPI_A_BF16, "bf16 dY")):
        slabs = torch.full((ns * stride,), float("nan"), device=dev)
        H.call("vlg_linear_wgrad", dy_dev.data_ptr(), N, xd.data_ptr(), K, slabs.data_ptr(), stride, slabs.numel(), M, N, K,
               FL | bits | H.EPI_B_BF16, stream())
        g = reduce_slabs(H, slabs, stride, ns, stride, dev)
        assert_close(g[:N * K].view(N, K) / sc, (dy.double().t() @ x.double()).float() / sc, rtol=1e-4, atol=1e-5,
                     what="wgrad (%s, bf16 X)" % tag)
        assert_close(g[N * K:] / sc, dy.double().sum(0).float() / sc, rtol=1e-4, atol=1e-5, what="bias grad (%s)" % tag)
    # storage flags without the bf16 MFMA flag, or a combination no step uses, are refused
    with pytest.raises(H.HipError):
        H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, 0, M, N, K,
               H.EPI_BIAS | H.EPI_A_BF16, stream())


def test_adam_keeps_the_bf16_shadow(H, dev):
    """vlg_adam_step_bf16 = vlg_adam_step + shadow[i] = round-to-nearest-even bf16 of the updated parameter."""
    torch.manual_seed(14)
    n = 4096 + 8
    p0, g = torch.randn(n), torch.randn(n)
    outs = []
    for name in ("vlg_adam_step", "vlg_adam_step_bf16"):
        p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sh = torch.zeros(n, device=dev, dtype=BF)
        gd = g.to(dev)
        for step in (1, 2):
            args = [p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr()] + ([sh.data_ptr()] if name.endswith("bf16") else []) + \
                   [n, step, 2e-4, 0.5, 0.999, 1e-8, 1.0, stream()]
            H.call(name, *args)
        outs.append((p.cpu(), sh.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]), "the shadow variant must not change the fp32 update"
    assert torch.equal(outs[1][1], outs[1][0].to(BF))


# ------------------------------------------------ fp32 on the bf16 matrix cores (3-way split, csrc/gemm_split.hip)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (200, 768, 256), (333, 256, 1024), (1000, 24, 256)])
def test_linear_split3_is_fp32_grade(H, dev, M, N, K):
    """VLG_EPI_SPLIT3: fp32 tensors, each operand split exactly into three bf16 terms, six bf16 MFMAs per product block.
    Same 1e-4 parity bar as the native fp32 MFMA path, and its error against fp64 must stay within 4x of the native
    path's (it is typically about equal: both round every accumulation step to fp32)."""
    torch.manual_seed(15)
    S3 = H.EPI_SPLIT3
    a, w, bias = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    ref = a.double() @ w.double().t() + bias.double()
    err = {}
    for tag, fl in (("native", 0), ("split", S3)):
        c = torch.zeros(M, N, device=dev)
        H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, 0, M, N, K,
               H.EPI_BIAS | fl, stream())
        assert_close(c, ref.float(), rtol=1e-4, atol=1e-5, what="fwd bias (%s)" % tag)
        err[tag] = float((c.cpu().double() - ref).abs().max())
    assert err["split"] <= 4 * err["native"] + 1e-7, err
    if N > 32:
        u, c = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
        H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, 0, u.data_ptr(), M, N, K,
               H.EPI_BIAS | H.EPI_GELU | S3, stream())
        assert_close(u, ref.float(), rtol=1e-4, atol=1e-5, what="split fwd pre-activation")
        assert_close(c, F.gelu(ref).float(), rtol=1e-4, atol=1e-5, what="split fwd gelu")
        resid = torch.randn(M, N)
        rd = resid.to(dev)
        H.call("vlg_linear_fwd", ad.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), c.data_ptr(), N, rd.data_ptr(), 0, M, N, K,
               H.EPI_BIAS | H.EPI_RESID | S3, stream())
        assert_close(c, (ref + resid.double()).float(), rtol=1e-4, atol=1e-5, what="split fwd bias+resid")
    if N % 8:
        return
    dy = torch.randn(M, N)
    dyd = dy.to(dev)
    dref = dy.double() @ w.double()
    dx = torch.zeros(M, K, device=dev)
    H.call("vlg_linear_dgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, 0, M, N, K, S3, stream())
    assert_close(dx, dref.float(), rtol=1e-4, atol=1e-5, what="split dgrad")
    if N > 32:
        pre = torch.randn(M, K)
        pd = pre.to(dev)
        H.call("vlg_linear_dgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, pd.data_ptr(), M, N, K,
               H.EPI_DGELU | S3, stream())
        uu = pre.double().requires_grad_(True)
        F.gelu(uu).backward(dref)
        assert_close(dx, uu.grad.float(), rtol=1e-4, atol=1e-5, what="split dgrad * gelu'")
    x = torch.randn(M, K)
    xd = x.to(dev)
    ns = H.load().vlg_linear_wgrad_slabs_for(M, N, K, S3)
    stride = N * K + N
    slabs = torch.full((ns * stride,), float("nan"), device=dev)
    H.call("vlg_linear_wgrad", dyd.data_ptr(), N, xd.data_ptr(), K, slabs.data_ptr(), stride, slabs.numel(), M, N, K, S3, stream())
    g = reduce_slabs(H, slabs, stride, ns, stride, dev)
    sc = math.sqrt(M)
    assert_close(g[:N * K].view(N, K) / sc, (dy.double().t() @ x.double()).float() / sc, rtol=1e-4, atol=1e-5, what="split wgrad")
    assert_close(g[N * K:] / sc, dy.double().sum(0).float() / sc, rtol=1e-4, atol=1e-5, what="split bias grad")
    with pytest.raises(H.HipError):                                  # the split is an fp32-tensor mode
        H.call("vlg_linear_dgrad", dyd.data_ptr(), N, wd.data_ptr(), K, dx.data_ptr(), K, 0, M, N, K, S3 | H.EPI_BF16, stream())


def test_short_slab_buffers_are_refused(H, dev):
    """Every slab producer checks `slab_capacity` against slab count x stride on the host and refuses the launch
    (VLG_ERR_SHAPE) - an undersized buffer must never become an out-of-bounds device write."""
    M, N, K, d = 4096, 256, 128, 128
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    lib = H.load()
    ns = lib.vlg_linear_wgrad_slabs(M, N, K)
    stride = N * K + N
    slabs = torch.empty(ns * stride, device=dev)
    with pytest.raises(H.HipError):
        H.call("vlg_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, slabs.data_ptr(), stride, ns * stride - 1, M, N, K, 0, stream())
    H.call("vlg_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, slabs.data_ptr(), stride, ns * stride, M, N, K, 0, stream())
    xs = torch.randn(M, d, device=dev)
    mean, rstd, g = torch.zeros(M, device=dev), torch.ones(M, device=dev), torch.ones(d, device=dev)
    nl = lib.vlg_layernorm_bwd_slabs(M)
    ls = torch.empty(nl * 2 * d, device=dev)
    out = torch.empty(M, d, device=dev)
    with pytest.raises(H.HipError):
        H.call("vlg_layernorm_bwd", xs.data_ptr(), xs.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), 0, out.data_ptr(),
               ls.data_ptr(), 2 * d, nl * 2 * d - 1, M, d, stream())
    torch.cuda.synchronize()
