"""Rotary positions folded into the attention kernels (pk_attn_fwd_rope / pk_attn_bwd_rope: the kernels rotate q and k as
they load them and rotate dQ / dK back as they store them) against the two-pass form they replace (pk_rope on the packed
projection, attention, pk_rope inverse on the gradient: RotaryEmbedding.forward, pasero/models/modules.py:617-623,982-1025).
Same arithmetic in the forward direction (fp32 rotation of 16-bit inputs, rounded once), so o agrees to the storage
precision; dq / dk are rotated back in fp32 BEFORE their one rounding, so they agree to round-off.  Every kernel family:
fp32 (thread per row), the 16-bit flash kernels (T, S > 128), the fused backward (T, S <= 128), heads of 64 and 128,
causal, key padding, ragged lengths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pasero_amd import functional
    return functional


def _tables(max_pos, hd, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('hd', [64, 128])
def test_rope_inside_the_attention_kernels_matches_the_separate_pass(F, dtype, hd):
    rs = np.random.RandomState(3 + hd)
    H = 4
    D = H * hd
    tol = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]
    for (B, T, causal, pad) in ((3, 64, False, True), (2, 128, True, False), (2, 200, True, True), (1, 333, False, False),
                                (2, 77, False, True)):
        cos_t, sin_t = _tables(512, hd)
        qkv = torch.from_numpy(rs.standard_normal((B, T, 3 * D)).astype(np.float32)).to(dtype).cuda()
        d_o = torch.from_numpy(rs.standard_normal((B, T, D)).astype(np.float32)).to(dtype).cuda()
        key_pad = None
        if pad and not causal:
            lens = rs.randint(T // 2, T + 1, size=B)
            key_pad = (torch.arange(T)[None, :] >= torch.from_numpy(lens)[:, None]).cuda()
        scale = hd ** -0.5
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        # two passes: rotate the packed projection, attend, rotate the gradient back
        rot = F.rope(qkv, cos_t, sin_t, 2 * D, 0)
        rq, rk, rv = rot[..., :D], rot[..., D:2 * D], rot[..., 2 * D:]
        o2, lse2 = F.attn_fwd(rq, rk, rv, H, key_pad, causal, scale)
        g = torch.empty_like(qkv)
        F.attn_bwd(rq, rk, rv, o2, d_o, lse2, H, key_pad, causal, scale, dq=g[..., :D], dk=g[..., D:2 * D], dv=g[..., 2 * D:])
        g2 = F.rope(g, cos_t, sin_t, 2 * D, 0, inverse=True)
        # one pass: the kernels rotate
        rope = (cos_t, sin_t, 0, 0)
        o1, lse1 = F.attn_fwd(q, k, v, H, key_pad, causal, scale, rope=rope)
        g1 = torch.empty_like(qkv)
        F.attn_bwd(q, k, v, o1, d_o, lse1, H, key_pad, causal, scale, dq=g1[..., :D], dk=g1[..., D:2 * D], dv=g1[..., 2 * D:],
                   rope=rope)
        what = (str(dtype), hd, B, T, causal, pad)
        assert torch.isfinite(o1.float()).all() and torch.isfinite(g1.float()).all(), what
        assert (o1.float() - o2.float()).abs().max().item() <= tol * max(1.0, o2.float().abs().max().item()), what
        assert (lse1 - lse2).abs().max().item() <= 10 * tol, what
        for name, lo in (('dq', 0), ('dk', D), ('dv', 2 * D)):
            a, b_ = g1[..., lo:lo + D].float(), g2[..., lo:lo + D].float()
            assert (a - b_).abs().max().item() <= tol * max(1.0, b_.abs().max().item()), (what, name)


def test_rope_positions_start_where_they_are_told(F):
    """q_pos0 / k_pos0: the kernels' positions are offsets into the table (keys of a window that starts at position 5)"""
    rs = np.random.RandomState(9)
    B, T, H, hd = 2, 40, 2, 64
    D = H * hd
    cos_t, sin_t = _tables(128, hd)
    qkv = torch.from_numpy(rs.standard_normal((B, T, 3 * D)).astype(np.float32)).cuda()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    rot = F.rope(qkv, cos_t, sin_t, 2 * D, 5)
    o2, _ = F.attn_fwd(rot[..., :D], rot[..., D:2 * D], rot[..., 2 * D:], H, None, True, 0.125)
    o1, _ = F.attn_fwd(q, k, v, H, None, True, 0.125, rope=(cos_t, sin_t, 5, 5))
    assert (o1 - o2).abs().max().item() <= 2e-5
    with pytest.raises(RuntimeError, match='positions exceed'):
        F.attn_fwd(q, k, v, H, None, True, 0.125, rope=(cos_t[:32].contiguous(), sin_t[:32].contiguous(), 0, 0))
