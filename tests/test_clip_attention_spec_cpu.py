"""CPU: properties of the per-clip attention SPECIFICATION (oracle/layout_spec.py:clip_attention, SELF-ORACLE - the reference
has no attention; the HIP kernels are checked against this function in tests/test_hip_ops.py / test_hip_step.py)."""
import torch

from oracle import layout_spec as O


def test_one_slot_per_frame_is_the_per_slot_attention():
    torch.manual_seed(0)
    qkv = torch.randn(2, 8, 1, 3 * 128)
    assert torch.allclose(O.clip_attention(qkv, 2), O.temporal_attention(qkv, 2), atol=1e-6)


def test_future_frames_and_padded_slots_are_invisible():
    torch.manual_seed(1)
    B, T, N, d = 2, 4, 6, 64
    qkv = torch.randn(B, T, N, 3 * d)
    valid = torch.ones(B, T, N)
    valid[:, :, 4:] = 0.0                                             # slots 4, 5 are padding
    base = O.clip_attention(qkv, 1, valid)
    # 1. changing k / v of a LATER frame changes nothing in earlier frames
    q2 = qkv.clone()
    q2[:, 3, :, d:] += 5.0
    out = O.clip_attention(q2, 1, valid)
    assert torch.equal(out[:, :3], base[:, :3]) and not torch.equal(out[:, 3], base[:, 3])
    # 2. changing k / v of a PADDED slot changes only that slot's own rows
    q3 = qkv.clone()
    q3[:, 1, 5, d:] += 5.0
    out = O.clip_attention(q3, 1, valid)
    same = torch.ones(B, T, N, dtype=torch.bool)
    same[:, 1, 5] = False
    assert torch.equal(out[same], base[same]) and not torch.equal(out[:, 1, 5], base[:, 1, 5])
    # 3. slots of the SAME frame see each other (the point of the per-clip reading)
    q4 = qkv.clone()
    q4[:, 2, 0, d:] += 5.0
    out = O.clip_attention(q4, 1, valid)
    assert not torch.equal(out[:, 2, 1], base[:, 2, 1])


def test_step_option_changes_the_model_and_is_differentiable():
    from vlg.spec import LayoutConfig, param_shapes, step_flops
    cfg = LayoutConfig(B=2, T=4, N=8, d=64, n_layers=1, attention="clip")
    cfg.validate()
    assert "per-clip" in cfg.describe()["attention"]
    assert step_flops(cfg)["fwd_bwd"] > step_flops(LayoutConfig(B=2, T=4, N=8, d=64, n_layers=1))["fwd_bwd"]
    p = O.init_params(param_shapes(cfg), seed=3)
    batch = O.synthetic_batch(cfg.B, cfg.T, cfg.N, seed=3, variable_n=True, min_valid=3)
    slot, _ = O.loss_and_grads(p, batch, cfg.n_layers)
    clip, g = O.loss_and_grads(p, batch, cfg.n_layers, attention="clip")
    assert abs(slot[0] - clip[0]) > 1e-6 and all(torch.isfinite(v).all() for v in g.values())
    try:
        LayoutConfig(B=1, T=3, N=5, d=64, attention="clip").validate()
        raise AssertionError("T*N = 15 must be refused")
    except ValueError:
        pass
