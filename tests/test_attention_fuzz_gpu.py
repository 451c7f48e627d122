"""Randomised attention problems (sizes around the tile boundaries of every kernel variant, heads of 64 and 128, causal /
key-padding masks, packed-projection strides, all three dtypes) against the oracle's explicit softmax attention."""
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
    # floor on the scale: with a single visible key the softmax is constant and dq, dk are exactly zero, while the
    # kernels compute them as dO.v - rowsum(dO*O) — a difference of two O(sqrt(hd)) sums, i.e. round-off of unit inputs
    return ((a - b).abs().max() / b.abs().max().clamp_min(0.05)).item()


@pytest.mark.parametrize('seed', range(5))
def test_attention_fuzz(F, seed):
    rs = np.random.RandomState(2000 + seed)
    lens_choice = [1, 2, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 160, 255, 256, 257, 300]
    for case in range(14):
        dtype = [torch.bfloat16, torch.float16, torch.float32][rs.randint(3)]
        hd = int(rs.choice([64, 128]))
        B, H = int(rs.randint(1, 4)), int(rs.randint(1, 4))
        causal = bool(rs.randint(2))
        T = int(rs.choice(lens_choice))
        S = T if (causal and rs.randint(2)) else int(rs.choice(lens_choice))
        if causal and S < T:
            S = T  # (the reference's causal mask assumes the queries are the last T positions of the keys)
        D = H * hd
        packed = (S == T) and bool(rs.randint(2))  # self-attention on the packed (B, T, 3D) projection output
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        if packed:
            qkv = torch.randn(B, T, 3 * D, generator=g).to(dtype)
            q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
            dq_, dk_, dv_ = qkv.cuda()[..., :D], qkv.cuda()[..., D:2 * D], qkv.cuda()[..., 2 * D:]
        else:
            q, k, v = (torch.randn(B, L, D, generator=g).to(dtype) for L in (T, S, S))
            dq_, dk_, dv_ = q.cuda(), k.cuda(), v.cuda()
        dy = torch.randn(B, T, D, generator=g).to(dtype)
        key_pad = None
        if not causal and rs.randint(2):
            lens = torch.from_numpy(rs.randint(1, S + 1, size=B))
            lens[rs.randint(B)] = S
            key_pad = O.len_to_mask(lens, S)
        scale = 1.0 / np.sqrt(hd)
        qf, kf, vf = (t.float().clone().requires_grad_() for t in (q, k, v))
        out, _ = O.attention_core(qf.view(B, T, H, hd), kf.view(B, S, H, hd), vf.view(B, S, H, hd), key_pad, causal, scale)
        out = out.reshape(B, T, D)
        out.backward(dy.float())
        kp = key_pad.cuda() if key_pad is not None else None
        o, lse = F.attn_fwd(dq_, dk_, dv_, H, kp, causal, scale)
        dq, dk, dv = F.attn_bwd(dq_, dk_, dv_, o, dy.cuda(), lse, H, kp, causal, scale)
        what = (seed, case, str(dtype), hd, B, H, T, S, causal, packed, key_pad is not None)
        tol = 2e-5 if dtype == torch.float32 else (2.5e-2 if dtype == torch.bfloat16 else 4e-3)
        for name, got, want in (('o', o, out.detach()), ('dq', dq, qf.grad), ('dk', dk, kf.grad), ('dv', dv, vf.grad)):
            assert torch.isfinite(got.float()).all(), (what, name)
            assert rel_err(got, want) < tol, (what, name, rel_err(got, want))
