"""The long-sequence attention kernels of csrc/attention_long.hip (heads of 64, no causal mask: forward `attn_fwd_long_kernel`
with its lagging maximum and dQ `attn_dq_long_kernel` at S >= 256, dK / dV `attn_dkv_long_kernel` at T >= 256; K / V resp.
Q / dO tiles by LDS-DMA) against the oracle's explicit softmax attention on the cases their shortcuts could get wrong:
rows whose maximum keeps growing from tile to tile (every tile re-anchors) or grows once by a large step; key-padding
masks that are no suffix (whole tiles masked in front of, between and behind visible keys; rows with no visible key);
query / key counts off the tile sizes; packed-projection strides; both 16-bit types.  Reference: modules.py:654-677, 707-771."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pasero_amd import functional
    return functional


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(0.05)).item()


def run(F, q, k, v, H, key_pad, scale):
    B, T, D = q.shape
    S, hd = k.size(1), D // H
    out, _ = O.attention_core(q.float().view(B, T, H, hd), k.float().view(B, S, H, hd), v.float().view(B, S, H, hd),
                              key_pad, False, scale)
    sc = torch.einsum('bthd,bshd->bhts', q.float().view(B, T, H, hd), k.float().view(B, S, H, hd)) * scale
    if key_pad is not None:
        sc = sc.masked_fill(key_pad[:, None, None, :], float('-inf'))
    lse_ref = torch.logsumexp(sc, -1)
    o, lse = F.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, key_pad.cuda() if key_pad is not None else None, False, scale)
    return o.cpu(), lse.cpu(), out.reshape(B, T, D), lse_ref


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,T,S', [(2, 2, 130, 256), (1, 3, 257, 500), (2, 1, 64, 1500), (1, 2, 1500, 1500)])
@pytest.mark.parametrize('growth', ['flat', 'ramp', 'step'])
def test_long_forward_growing_maximum(F, dtype, B, H, T, S, growth):
    """flat: unit-variance scores (the maximum settles in the first tiles); ramp: key norms grow along the sequence so the
    row maximum rises in (nearly) every tile and each tile re-anchors; step: one key far down the sequence beats everything
    before it by ~40 in the exp2 domain (the accumulator is rescaled by 2^-40 once, late)."""
    g = torch.Generator().manual_seed(S * 7 + T)
    D = H * 64
    q, k, v = (torch.randn(B, n, D, generator=g) for n in (T, S, S))
    if growth == 'ramp':
        k = k * torch.linspace(0.5, 6.0, S)[None, :, None]
    elif growth == 'step':
        q = q + 1.0
        k[:, (3 * S) // 4] = 3.0   # score ~ 3 * 64 * 0.125 = 24 (35 in the exp2 domain) against a spread of ~1.4
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    o, lse, ref, lse_ref = run(F, q, k, v, H, None, 0.125)
    assert torch.isfinite(o.float()).all()
    assert rel_err(o, ref) < (2.5e-2 if dtype == torch.bfloat16 else 4e-3), rel_err(o, ref)
    assert (lse - lse_ref).abs().max().item() < 2e-3 * max(1.0, lse_ref.abs().max().item())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_long_forward_masks_that_are_no_suffix(F, dtype):
    """batch row 0: keys 0..199 masked (three whole tiles before the first visible key: the row stays unanchored through
    them); row 1: visible keys only in [70, 90) and [400, 410) (masked tiles between and behind); row 2: nothing visible
    (output 0, lse 0: the reference's nan_to_num); row 3: every second key masked; row 4: no mask."""
    B, H, T, S, D = 5, 2, 200, 453, 128
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, S, 3 * D, generator=g).to(dtype)
    q, k, v = qkv[:, :T, :D], qkv[..., D:2 * D], qkv[..., 2 * D:]   # strided views of one packed buffer
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[0, :200] = True
    pad[1] = True
    pad[1, 70:90] = False
    pad[1, 400:410] = False
    pad[2] = True
    pad[3, ::2] = True
    B_, T_, D_ = q.shape
    qc = qkv.cuda()
    o, lse = F.attn_fwd(qc[:, :T, :D], qc[..., D:2 * D], qc[..., 2 * D:], H, pad.cuda(), False, 0.125)
    out, _ = O.attention_core(q.float().reshape(B, T, H, 64), k.float().reshape(B, S, H, 64), v.float().reshape(B, S, H, 64),
                              pad, False, 0.125)
    out = out.reshape(B, T, D)
    assert torch.isfinite(o.float()).all()
    assert (o[2] == 0).all() and (lse[2] == 0).all()
    assert rel_err(o.cpu(), out) < (2.5e-2 if dtype == torch.bfloat16 else 4e-3)


def test_long_forward_is_row_local(F):
    """a row's output does not depend on which other rows share its wave / workgroup / batch: the same (b, h) rows computed
    inside a larger batch and on their own are bitwise equal (the re-anchoring decision is per row)."""
    g = torch.Generator().manual_seed(11)
    B, H, T, S, D = 3, 2, 300, 700, 128
    q, k, v = (torch.randn(B, n, D, generator=g) for n in (T, S, S))
    k = k * torch.linspace(0.5, 5.0, S)[None, :, None]
    q, k, v = (t.bfloat16().cuda() for t in (q, k, v))
    o, lse = F.attn_fwd(q, k, v, H, None, False, 0.125)
    o1, lse1 = F.attn_fwd(q[1:2, 37:200].contiguous(), k[1:2].contiguous(), v[1:2].contiguous(), H, None, False, 0.125)
    assert torch.equal(o[1:2, 37:200], o1) and torch.equal(lse[1:2, :, 37:200], lse1)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,T,S,packed', [(2, 2, 300, 700, False), (1, 3, 500, 500, True), (2, 1, 257, 256, False),
                                            (1, 2, 1500, 1500, True), (3, 2, 64, 1500, False), (2, 2, 700, 100, False)])
def test_long_backward_against_the_oracle(F, dtype, B, H, T, S, packed):
    """dQ (S >= 256) and dK / dV (T >= 256) through pk_attn_bwd with key-padding masks that are no suffix (row 0: the first
    70 keys and every third key masked; the last batch row: nothing but keys 5..9 visible) — gradients of the oracle's
    attention by autograd in fp32.  (T = 64 x S = 1500: the long dQ kernel beside the tiled dK / dV kernel; 700 x 100 the
    other way round.)"""
    g = torch.Generator().manual_seed(T * 3 + S)
    D = H * 64
    if packed:
        qkv = torch.randn(B, T, 3 * D, generator=g).to(dtype)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        qc = qkv.cuda()
        qg, kg, vg = qc[..., :D], qc[..., D:2 * D], qc[..., 2 * D:]
    else:
        q, k, v = (torch.randn(B, n, D, generator=g).to(dtype) for n in (T, S, S))
        qg, kg, vg = q.cuda(), k.cuda(), v.cuda()
    dy = torch.randn(B, T, D, generator=g).to(dtype)
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[0, :min(70, S - 1)] = True
    pad[0, ::3] = True
    pad[0, S - 1] = False
    pad[B - 1] = True
    pad[B - 1, 5:10] = False
    qf, kf, vf = (x.float().clone().requires_grad_() for x in (q, k, v))
    out, _ = O.attention_core(qf.view(B, T, H, 64), kf.view(B, S, H, 64), vf.view(B, S, H, 64), pad, False, 0.125)
    out.reshape(B, T, D).backward(dy.float())
    o, lse = F.attn_fwd(qg, kg, vg, H, pad.cuda(), False, 0.125)
    dq, dk, dv = F.attn_bwd(qg, kg, vg, o, dy.cuda(), lse, H, pad.cuda(), False, 0.125)
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
    for name, got, want in (('o', o, out.detach().reshape(B, T, D)), ('dq', dq, qf.grad), ('dk', dk, kf.grad), ('dv', dv, vf.grad)):
        assert torch.isfinite(got.float()).all(), name
        assert rel_err(got, want) < tol, (name, rel_err(got, want))
    assert (dk[0][pad[0].cuda()] == 0).all() and (dv[0][pad[0].cuda()] == 0).all()  # masked keys receive no gradient


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,T,S', [(2, 2, 500, 500), (1, 2, 1500, 1500), (2, 1, 130, 700), (1, 3, 64, 64), (2, 2, 129, 129)])
def test_long_forward_causal(F, dtype, B, H, T, S):
    """the same kernel under the causal mask (query t sees keys <= t + S - T): tiles in a wave's future skipped, the
    diagonal's masked, the heaviest query blocks dealt first — output, lse and (through the tiled backward kernels) the
    gradients against the oracle"""
    g = torch.Generator().manual_seed(T + 2 * S)
    D = H * 64
    q, k, v, dy = (torch.randn(B, n, D, generator=g).to(dtype) for n in (T, S, S, T))
    qf, kf, vf = (x.float().clone().requires_grad_() for x in (q, k, v))
    out, _ = O.attention_core(qf.view(B, T, H, 64), kf.view(B, S, H, 64), vf.view(B, S, H, 64), None, True, 0.125)
    out.reshape(B, T, D).backward(dy.float())
    sc = torch.einsum('bthd,bshd->bhts', q.float().view(B, T, H, 64), k.float().view(B, S, H, 64)) * 0.125
    sc = sc.masked_fill(torch.ones(T, S, dtype=torch.bool).triu(1 + S - T), float('-inf'))
    o, lse = F.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, None, True, 0.125)
    dq, dk, dv = F.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o, dy.cuda(), lse, H, None, True, 0.125)
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
    assert (lse.cpu() - torch.logsumexp(sc, -1)).abs().max().item() < 2e-3 * max(1.0, torch.logsumexp(sc, -1).abs().max().item())
    for name, got, want in (('o', o, out.detach().reshape(B, T, D)), ('dq', dq, qf.grad), ('dk', dk, kf.grad), ('dv', dv, vf.grad)):
        assert torch.isfinite(got.float()).all(), name
        assert rel_err(got, want) < tol, (name, rel_err(got, want))


@pytest.mark.parametrize('B,H,T,S', [(2, 2, 300, 300), (2, 1, 130, 700)])
def test_long_kernels_causal_with_key_padding(F, B, H, T, S):
    """both masks at once (the reference adds them, modules.py:654-677): padding keys inside the causal window, the last batch
    row with its first 40 keys masked (its first queries see no key at all: zero rows, zero lse)"""
    g = torch.Generator().manual_seed(T + S)
    D = H * 64
    q, k, v, dy = (torch.randn(B, n, D, generator=g).bfloat16() for n in (T, S, S, T))
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[0, 5::7] = True
    pad[B - 1, :40] = True
    qf, kf, vf = (x.float().clone().requires_grad_() for x in (q, k, v))
    out, _ = O.attention_core(qf.view(B, T, H, 64), kf.view(B, S, H, 64), vf.view(B, S, H, 64), pad, True, 0.125)
    out.reshape(B, T, D).backward(dy.float())
    o, lse = F.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, pad.cuda(), True, 0.125)
    dq, dk, dv = F.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o, dy.cuda(), lse, H, pad.cuda(), True, 0.125)
    for name, got, want in (('o', o, out.detach().reshape(B, T, D)), ('dq', dq, qf.grad), ('dk', dk, kf.grad), ('dv', dv, vf.grad)):
        assert torch.isfinite(got.float()).all(), name
        assert rel_err(got, want) < 2.5e-2, (name, rel_err(got, want))
    if S == T:  # queries 0..39 of the last row see only masked keys
        assert (o[B - 1, :40] == 0).all() and (lse[B - 1, :, :40] == 0).all()
