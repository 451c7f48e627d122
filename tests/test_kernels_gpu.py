"""Kernel-level parity tests (GPU): every C-ABI entry point against the CPU oracle / an fp64 restatement on the same
seeded inputs.  fp32 kernels: tight tolerance; bf16 kernels: tolerance of bf16 storage (2^-8 relative), stated per test.
"""
import math

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
    a = a.double().cpu()
    b = b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rnd(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


GEMM_SHAPES = [
    (128, 128, 64), (256, 384, 512), (130, 70, 100), (1, 8032, 512), (300, 8032, 128), (1000, 512, 2048),
    (64, 64, 8), (257, 129, 65),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('a_col,b_col', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
def test_gemm_layouts(F, dtype, a_col, b_col, M, N, K):
    A = rnd((M, K), 1, dtype)
    B = rnd((N, K), 2, dtype)
    ref = A.double() @ B.double().t()
    a = (A.t().contiguous() if a_col else A).cuda()
    b = (B.t().contiguous() if b_col else B).cuda()
    out = F.gemm(a, b, a_col=a_col, b_col=b_col)
    torch.cuda.synchronize()
    # fp32: exact-fp32 MFMA fma chain; bf16: inputs exact, fp32 accumulate, output rounded to bf16 (2^-9 rel)
    tol = 2e-6 * math.sqrt(K) if dtype == torch.float32 else 6e-3
    assert rel_err(out, ref) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gemm_asymmetric_identity(F, dtype):
    """A = I with an asymmetric B catches a transposed C write (cdna guide §3)"""
    n = 128
    A = torch.eye(n, dtype=dtype)
    B = (torch.arange(n)[:, None] * 3 + torch.arange(n)[None, :] % 7).to(dtype)  # B[i][j] != B[j][i]
    out = F.gemm(A.cuda(), B.cuda())  # C = A Bᵀ = Bᵀ
    assert torch.equal(out.cpu().float(), B.t().float())
    out = F.gemm(A.cuda(), B.cuda(), b_col=True)  # C[m,n] = sum_k A[m,k] B[k,n] = B
    assert torch.equal(out.cpu().float(), B.float())
    out = F.gemm(B.cuda(), A.cuda(), a_col=True)  # C[m,n] = sum_k B[k,m] A[n,k] = B[n,m]... = Bᵀ
    assert torch.equal(out.cpu().float(), B.t().float())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('act', ['none', 'relu', 'gelu', 'gelu_tanh', 'silu'])
def test_gemm_epilogues(F, dtype, act):
    M, N, K = 200, 264, 96
    x = rnd((M, K), 3, dtype)
    w = rnd((N, K), 4, dtype, K ** -0.5)
    bias = rnd((N,), 5, dtype)
    res = rnd((M, N), 6, dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    pre_ref = O.linear(x.float(), w.float(), bias.float())
    ref = O.activation('swiglu' if act == 'silu' else act, pre_ref) if act != 'none' else pre_ref
    pre = torch.empty(M, N, dtype=dtype, device='cuda')
    out = F.gemm(x.cuda(), w.cuda(), bias=bias.cuda(), act=act, preact=pre)
    assert rel_err(out, ref) < tol
    assert rel_err(pre, pre_ref) < tol
    out = F.gemm(x.cuda(), w.cuda(), bias=bias.cuda(), act=act, aux=res.cuda(), mode=1, alpha=0.5)
    ref1 = O.activation('swiglu' if act == 'silu' else act, 0.5 * (x.float() @ w.float().t()) + bias.float()) \
        if act != 'none' else 0.5 * (x.float() @ w.float().t()) + bias.float()
    assert rel_err(out, ref1 + res.float()) < tol
    # mode 2: v * act'(aux): compare with autograd of the oracle's activation
    z = res.float().clone().requires_grad_()
    y = O.activation('swiglu' if act == 'silu' else act, z) if act != 'none' else z * 1.0
    y.backward(torch.ones_like(y))
    out = F.gemm(x.cuda(), w.cuda(), act=act, aux=res.cuda(), mode=2)
    assert rel_err(out, (x.float() @ w.float().t()) * z.grad) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('splitk', [2, 5, 16])
def test_gemm_splitk(F, dtype, splitk):
    M, N, K = 192, 160, 4100  # weight-gradient shape: small output, long contraction (K not a multiple of BK)
    A = rnd((K, M), 7, dtype)
    B = rnd((K, N), 8, dtype)
    acc = rnd((M, N), 9, dtype)
    ref = A.double().t() @ B.double() + acc.double()
    out = F.gemm(A.cuda(), B.cuda(), a_col=True, b_col=True, aux=acc.cuda(), mode=1, splitk=splitk)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(out, ref) < tol


def test_gemm_strided_views(F):
    """q/k/v-style column slices and a padded leading dimension"""
    M, N, K = 96, 64, 128
    big = rnd((M, 3 * K), 10, torch.bfloat16).cuda()
    w = rnd((N, K), 11, torch.bfloat16).cuda()
    out = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device='cuda')
    F.gemm(big[:, K:2 * K], w, out=out[:, N:])
    ref = big[:, K:2 * K].double().cpu() @ w.double().cpu().t()
    assert rel_err(out[:, N:], ref) < 6e-3
    assert out[:, :N].abs().max().item() == 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,d', [(7, 128), (1000, 512), (33, 1024), (5, 2048), (3, 520)])
def test_layernorm_fwd_bwd(F, dtype, rows, d):
    x = rnd((rows, d), 20, dtype)
    res = rnd((rows, d), 21, dtype)
    gamma = (1 + 0.1 * rnd((d,), 22, torch.float32)).to(dtype)
    beta = rnd((d,), 23, dtype, 0.1)
    dy = rnd((rows, d), 24, dtype)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    # oracle
    xz = (x.float() + res.float()).to(dtype).float().requires_grad_()  # z is a tensor of `dtype` in the reference
    g32, b32 = gamma.float().requires_grad_(), beta.float().requires_grad_()
    y_ref = O.layer_norm(xz, g32, b32, 1e-5)
    y_ref.backward(dy.float())
    y, z, mean, rstd = F.residual_ln_fwd(x.cuda(), res.cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    assert rel_err(z, xz.detach()) < tol
    assert rel_err(y, y_ref.detach()) < tol
    dres, dx, dgamma, dbeta = F.residual_ln_bwd(dy.cuda(), None, z, gamma.cuda(), mean, rstd, want_dres=True,
                                                want_dx=False, want_param_grads=True)
    assert rel_err(dres, xz.grad) < tol
    assert rel_err(dgamma, g32.grad) < tol
    assert rel_err(dbeta, b32.grad) < tol
    # plain LN (no residual), and the pre-norm residual-only mode with an extra incoming gradient
    y2, _, _, _ = F.residual_ln_fwd(x.cuda(), None, gamma.cuda(), beta.cuda(), 1e-5, want_z=False)
    assert rel_err(y2, O.layer_norm(x.float(), gamma.float(), beta.float(), 1e-5)) < tol
    _, z3, _, _ = F.residual_ln_fwd(x.cuda(), res.cuda(), None, None, 1e-5)
    assert rel_err(z3, x.float() + res.float()) < tol
    extra = rnd((rows, d), 25, dtype)
    dres2, _, _, _ = F.residual_ln_bwd(dy.cuda(), extra.cuda(), z, gamma.cuda(), mean, rstd, want_dres=True,
                                       want_dx=False, want_param_grads=False)
    assert rel_err(dres2, xz.grad + extra.float()) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_residual_ln_mask_consistency(F, dtype):
    """dropout inside the fused kernel: keep-rate ~ 1-p, kept values scaled by 1/(1-p), and the backward pass
    regenerates the very same mask from (seed, offset)"""
    rows, d, p = 512, 512, 0.1
    x = torch.ones(rows, d, dtype=dtype).cuda()
    res = torch.zeros(rows, d, dtype=dtype).cuda()
    _, z, _, _ = F.residual_ln_fwd(x, res, None, None, 1e-5, drop_p=p, seed=1234, offset=7)
    kept = z != 0
    rate = kept.float().mean().item()
    assert abs(rate - (1 - p)) < 5e-3
    assert torch.allclose(z[kept].float(), torch.full_like(z[kept].float(), 1 / (1 - p)), rtol=1e-2)
    dz = torch.ones(rows, d, dtype=dtype).cuda()
    _, dx, _, _ = F.residual_ln_bwd(None, dz, None, None, None, None, want_dres=False, want_dx=True,
                                    want_param_grads=False, drop_p=p, seed=1234, offset=7)
    assert torch.equal(dx != 0, kept)
    _, z2, _, _ = F.residual_ln_fwd(x, res, None, None, 1e-5, drop_p=p, seed=1234, offset=8)
    assert not torch.equal(z2 != 0, kept)  # a different offset gives an independent mask
