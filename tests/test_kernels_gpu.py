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


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
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
    tol = 2e-6 * math.sqrt(K) if dtype == torch.float32 else (6e-3 if dtype == torch.bfloat16 else 8e-4)  # fp16: 2^-11
    assert rel_err(out, ref) < tol


@pytest.mark.parametrize('act', ['none', 'relu', 'gelu'])
@pytest.mark.parametrize('M,N,K', [(1, 512, 512), (7, 1536, 512), (64, 512, 2048), (64, 8032, 512), (100, 520, 64),
                                   (256, 2048, 512), (129, 36, 192), (1024, 512, 2048), (1000, 512, 512), (777, 264, 1024),
                                   (16000, 64, 1024), (3001, 40, 128)])
def test_gemm_few_rows_decoding_shapes(F, act, M, N, K):
    """M <= 256 (<= 1024 for outputs of <= 512 columns; any M for outputs of <= 64 columns: an adapter's down-projection),
    row-form bf16 operands, K % 64 == 0: the latency-shaped kernel of a decoding step / a short decoder batch (gemm_skinny.hip);
    bias + activation (+ residual) epilogues, ragged M and N tails, a strided input (q columns of a packed projection)"""
    x3 = rnd((M, 3 * K), 11, torch.bfloat16).cuda()
    x = x3[:, K:2 * K]  # row stride 3K, like q = qkv[:, :D]
    w = rnd((N, K), 12, torch.bfloat16, K ** -0.5).cuda()
    bias = rnd((N,), 13, torch.bfloat16).cuda()
    res = rnd((M, N), 14, torch.bfloat16).cuda()
    pre = x.float() @ w.float().t() + bias.float()
    ref = O.activation(act, pre.cpu()).cuda() if act != 'none' else pre
    out = F.gemm(x, w, bias=bias, act=act)
    assert rel_err(out, ref) < 1e-2
    out = F.gemm(x, w, bias=bias, act=act, aux=res, mode=1)
    assert rel_err(out, ref + res.float()) < 1e-2
    out = F.gemm(x, w)
    assert rel_err(out, x.float() @ w.float().t()) < 1e-2


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
@pytest.mark.parametrize('M,N,K', [(200, 264, 96), (384, 512, 128)])  # ragged edge tiles; whole tiles only (the lean epilogues)
def test_gemm_epilogues(F, dtype, act, M, N, K):
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


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('a_col,b_col', [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize('M,N,K,splitk', [(130, 70, 100, 1), (48, 512, 512, 1), (1024, 1024, 512, 1),
                                          (512, 512, 2048, 4), (200, 264, 1100, 2)])
def test_gemm_accumulates_in_place(F, dtype, a_col, b_col, M, N, K, splitk):
    """C = A·B + C with aux aliasing out (the fused loss accumulates dW across row chunks, the gated FFN adds its
    second dX): every kernel variant (skinny, 128- and 256-tiles, split-K reduce) must read an element of aux in the
    thread that writes it, and the result must not depend on when other workgroups store"""
    A = rnd((K, M) if a_col else (M, K), 31, dtype)
    B = rnd((K, N) if b_col else (N, K), 32, dtype)
    acc = rnd((M, N), 33, dtype)
    ref = (A.double().t() if a_col else A.double()) @ (B.double() if b_col else B.double().t()) + acc.double()
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    out = acc.cuda()
    got = F.gemm(A.cuda(), B.cuda(), a_col=a_col, b_col=b_col, aux=out, mode=1, out=out, splitk=splitk)
    assert got.data_ptr() == out.data_ptr()
    assert rel_err(out, ref) < tol
    sep = F.gemm(A.cuda(), B.cuda(), a_col=a_col, b_col=b_col, aux=acc.cuda(), mode=1, splitk=splitk)
    assert torch.equal(sep, out)  # bitwise the same as with separate buffers


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('splitk', [1, 7])
@pytest.mark.parametrize('M,N,K', [(192, 160, 4100), (2048, 512, 3000), (130, 70, 100)])
def test_gemm_fused_bias_gradient(F, dtype, splitk, M, N, K):
    """weight-gradient GEMM dW = dYᵀ X that also emits db = colsum(dY) (asum_out)"""
    dY = rnd((K, M), 12, dtype)
    X = rnd((K, N), 13, dtype)
    db = torch.empty(M, dtype=dtype, device='cuda')
    dW = F.gemm(dY.cuda(), X.cuda(), a_col=True, b_col=True, splitk=splitk, asum_out=db)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(dW, dY.double().t() @ X.double()) < tol
    assert rel_err(db, dY.double().sum(0)) < tol


@pytest.mark.parametrize('M,pad', [(1030, 2), (2051, 5), (262, 10)])
def test_gemm_col_form_rows_not_a_multiple_of_8(F, M, pad):
    """weight gradient of the vocabulary projection: A = dlogits (rows, V) in col form with V % 8 != 0 (NLLB: 256206),
    rows padded to a 16-byte multiple.  The 256-tile kernel reads the chunk that straddles the edge whole; the pad
    columns (poisoned here with NaN) only feed output rows that are not stored"""
    K, N = 4096, 512
    buf = torch.full((K, M + pad), float('nan'), dtype=torch.bfloat16)
    A = rnd((K, M), 51, torch.bfloat16)
    buf[:, :M] = A
    B = rnd((K, N), 52, torch.bfloat16)
    a = buf.cuda()[:, :M]
    assert a.stride(0) == M + pad and (M + pad) % 8 == 0 and M % 8 != 0
    ref = A.double().t() @ B.double()
    for sk in (1, 32):  # 128-tile kernel / 256-tile kernel with split-K
        out = F.gemm(a, B.cuda(), a_col=True, b_col=True, splitk=sk)
        assert torch.isfinite(out.float()).all()
        assert rel_err(out, ref) < 6e-3


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
@pytest.mark.parametrize('rows,d', [(7, 128), (1000, 512), (33, 1024), (5, 2048), (3, 520), (2100, 1280), (9, 4096)])
def test_layernorm_fwd_bwd(F, dtype, rows, d):
    if dtype == torch.float32 and d > 2048:
        pytest.skip('the backward kernel keeps a row in registers: fp32 rows up to 2048 (rejected loudly beyond)')
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


def test_wave_reductions_match_their_shuffle_statements(tmp_path):
    """wave_sum / wave_max / lanes8_sum / half_wave_max / half_wave_swap (common.h: permlane swaps + DPP instead of
    ds_bpermute round trips) bit for bit against the `__shfl_xor` statements they replace — every LayerNorm, softmax and
    attention-epilogue result depends on them.  tools/wave_reduce_check.hip includes the library's own header."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    exe = str(tmp_path / 'wave_reduce_check')
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', os.path.join(root, 'tools', 'wave_reduce_check.hip'), '-o', exe],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(': 0 mismatches') == 5, out.stdout


def test_specialised_layernorm_kernels_match_the_general_ones():
    """residual_ln_fwd16_kernel / residual_ln_bwd16_kernel (16-bit rows of 512 / 1024: template flags, packed arithmetic, a
    second register set for the next row) against the general kernels on the same inputs — bf16 and f16, every flag
    combination the dispatch takes, ragged row count: tools/ln_bench.py --compare runs the general kernels in a child process
    (PK_LN_FWD16=0 PK_LN_BWD16=0).  Parameter gradients and z are identical; the rest within one 16-bit ulp of a few elements."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'ln_bench.py'), '--compare'], capture_output=True, text=True,
                         cwd=root, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count('max|diff|') >= 100, out.stdout[-2000:]
    for line in out.stdout.splitlines():
        m = re.search(r'(dgamma|dbeta| z  ) +max\|diff\| ([0-9.e+-]+)', line)
        if m:
            assert float(m.group(2)) == 0.0, line
    worst = float(re.search(r'worst relative difference ([0-9.e+-]+)', out.stdout).group(1))
    assert worst < 8e-3, out.stdout[-2000:]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('rows,d', [(7, 128), (1000, 512), (5, 2048), (3, 520)])
def test_rmsnorm_fwd_bwd(F, dtype, rows, d):
    """RMSNorm = the same kernels with mean == NULL (pasero/models/modules.py:192-202)"""
    x = rnd((rows, d), 26, dtype) + 0.5  # a non-zero mean, so that a wrongly centred kernel fails
    gamma = (1 + 0.1 * rnd((d,), 27, torch.float32)).to(dtype)
    dy = rnd((rows, d), 28, dtype)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    x32, g32 = x.float().requires_grad_(), gamma.float().requires_grad_()
    y_ref = O.rms_norm(x32, g32, 1e-6)
    y_ref.backward(dy.float())
    y, _, mean, rstd = F.residual_ln_fwd(x.cuda(), None, gamma.cuda(), None, 1e-6, want_z=False, rms=True)
    assert mean is None
    assert rel_err(y, y_ref.detach()) < tol
    dx, _, dgamma, dbeta = F.residual_ln_bwd(dy.cuda(), None, x.cuda(), gamma.cuda(), None, rstd, want_dres=True,
                                             want_dx=False, want_param_grads=True, has_beta=False)
    assert dbeta is None
    assert rel_err(dx, x32.grad) < tol
    assert rel_err(dgamma, g32.grad) < tol


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


# ------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------
def _attn_oracle(q, k, v, H, key_pad, causal, scale, dy):
    B, T, D = q.shape
    hd = D // H
    q = q.clone().requires_grad_()
    k = k.clone().requires_grad_()
    v = v.clone().requires_grad_()
    out, _ = O.attention_core(q.view(B, T, H, hd), k.view(B, -1, H, hd), v.view(B, -1, H, hd), key_pad, causal, scale)
    out = out.reshape(B, T, D)
    out.backward(dy)
    return out.detach(), q.grad, k.grad, v.grad


ATTN_CASES = [
    # B, H, T, S, causal, ragged
    (2, 2, 5, 7, False, True),
    (3, 8, 64, 64, False, True),
    (2, 4, 128, 128, True, False),
    (2, 2, 100, 100, True, False),
    (1, 2, 130, 257, False, True),     # several key tiles, ragged tails
    (2, 2, 1, 9, True, False),          # incremental decoding step: one query, S > T
    (2, 2, 3, 70, True, False),         # causal with offset S - T
    (2, 1, 200, 200, True, False),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,T,S,causal,ragged', ATTN_CASES)
def test_attention_fwd_bwd(F, dtype, B, H, T, S, causal, ragged):
    D = H * 64
    q = rnd((B, T, D), 30, dtype)
    k = rnd((B, S, D), 31, dtype)
    v = rnd((B, S, D), 32, dtype)
    dy = rnd((B, T, D), 33, dtype)
    key_pad = None
    if ragged:
        lens = torch.randint(1, S + 1, (B,), generator=torch.Generator().manual_seed(34))
        lens[0] = S
        key_pad = O.len_to_mask(lens, S)
    scale = 1.0 / 8.0
    o_ref, dq_ref, dk_ref, dv_ref = _attn_oracle(q.float(), k.float(), v.float(), H, key_pad, causal, scale, dy.float())
    kp = key_pad.cuda() if key_pad is not None else None
    o, lse = F.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, kp, causal, scale)
    # fp32: exact arithmetic, different summation order.  bf16: P and dS are rounded to bf16 for the MFMA
    # (relative 2^-8), outputs rounded to bf16.
    tol = 1e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)  # fp16: 2^-11 storage
    assert rel_err(o, o_ref) < tol
    dq, dk, dv = F.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o, dy.cuda(), lse, H, kp, causal, scale)
    assert rel_err(dq, dq_ref) < tol
    assert rel_err(dk, dk_ref) < tol
    assert rel_err(dv, dv_ref) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,T,S,causal,ragged', [(2, 2, 70, 90, False, True), (2, 3, 128, 128, True, False),
                                                     (1, 2, 1, 37, True, False), (2, 1, 200, 130, False, True)])
def test_attention_head_dim_128(F, dtype, B, H, T, S, causal, ragged):
    """transformer_small / nllb_3b3 (4 or 16 heads of 128): the same kernels instantiated for head_dim 128 (forward
    and dQ with four d-tiles per wave; dV and dK in two launches), against the oracle's explicit softmax attention"""
    D = H * 128
    q, k, v, dy = (rnd((B, n, D), 20 + i, dtype) for i, n in enumerate((T, S, S, T)))
    key_pad = None
    if ragged:
        lens = torch.tensor([S] + [max(1, S - 17 * (i + 1)) for i in range(B - 1)])
        key_pad = O.len_to_mask(lens, S)
    scale = 128 ** -0.5
    refs = _attn_oracle(q.float(), k.float(), v.float(), H, key_pad, causal, scale, dy.float())
    kp = key_pad.cuda() if key_pad is not None else None
    o, lse = F.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, kp, causal, scale)
    dq, dk, dv = F.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o, dy.cuda(), lse, H, kp, causal, scale)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for got, want, name in zip((o, dq, dk, dv), refs, 'o dq dk dv'.split()):
        assert rel_err(got, want) < tol, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,T,S,causal,ragged,hd', [(2, 2, 70, 90, False, True, 64), (2, 4, 128, 128, True, False, 64),
                                                        (3, 2, 100, 128, False, True, 64), (1, 2, 130, 257, False, True, 64),
                                                        (2, 2, 64, 100, False, True, 128), (2, 4, 500, 500, False, True, 64),
                                                        (1, 2, 200, 333, True, False, 64)])
def test_attention_probability_dropout(F, dtype, B, H, T, S, causal, ragged, hd):
    """dropout_p of F.scaled_dot_product_attention (modules.py:707-720; attention_dropout 0.1 in the IWSLT2023 recipes):
    the forward kernel draws the keep bits and stores them; with THOSE bits the outputs and all three gradients must
    equal o = (softmax(s) * M / (1 - p)) v computed explicitly — the fused single-workgroup backward (T, S <= 128), the
    general dQ / dKdV kernels and head_dim 128 all read the same mask.  Plus: keep rate, determinism, lse unaffected."""
    pdrop, D = 0.25, H * hd
    q, k, v, dy = (rnd((B, n, D), 50 + i, dtype) for i, n in enumerate((T, S, S, T)))
    key_pad = None
    if ragged:
        lens = torch.tensor([S] + [max(2, S - 13 * (i + 1)) for i in range(B - 1)])
        key_pad = O.len_to_mask(lens, S)
    scale = hd ** -0.5
    kp = key_pad.cuda() if key_pad is not None else None
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    o, lse, mask = F.attn_fwd(qc, kc, vc, H, kp, causal, scale, pdrop, 77, 5)
    o0, lse0 = F.attn_fwd(qc, kc, vc, H, kp, causal, scale)
    assert torch.equal(lse, lse0)  # the softmax statistics do not see the dropout
    def unpack(m):  # (B,H,T,pitch) bytes -> (B,H,T,S) bool; bytes past ceil(S/8) are padding
        return ((m[..., :, None] >> torch.arange(8, device='cuda', dtype=torch.uint8)) & 1).flatten(-2)[..., :S].bool()
    bits = unpack(mask)
    assert bits.shape == (B, H, T, S)
    live = torch.ones(B, 1, T, S, dtype=torch.bool, device='cuda')  # bits of masked (query, key) pairs are never used
    if kp is not None:
        live = live & ~kp[:, None, None, :]
    if causal:
        live = live & ~torch.ones(T, S, device='cuda', dtype=torch.bool).triu(1 + S - T)
    live = live.expand(B, H, T, S)
    assert abs(bits[live].float().mean().item() - (1 - pdrop)) < 1e-2
    _, _, mask_again = F.attn_fwd(qc, kc, vc, H, kp, causal, scale, pdrop, 77, 5)
    _, _, mask_other = F.attn_fwd(qc, kc, vc, H, kp, causal, scale, pdrop, 77, 6)
    assert torch.equal(bits[live], unpack(mask_again)[live]) and not torch.equal(bits[live], unpack(mask_other)[live])
    # explicit reference with the kernel's own bits
    qf, kf, vf = (t.float().cuda().view(B, -1, H, hd).transpose(1, 2).requires_grad_() for t in (q, k, v))
    sc = (qf @ kf.transpose(-1, -2)) * scale
    if key_pad is not None:
        sc = sc.masked_fill(kp[:, None, None, :], float('-inf'))
    if causal:
        sc = sc.masked_fill(torch.ones(T, S, device='cuda', dtype=torch.bool).triu(1 + S - T), float('-inf'))
    pr = torch.softmax(sc, -1)
    ref = ((pr * bits / (1 - pdrop)) @ vf).transpose(1, 2).reshape(B, T, D)
    ref.backward(dy.float().cuda())
    back = lambda g: g.transpose(1, 2).reshape(B, -1, D)
    dq, dk, dv = F.attn_bwd(qc, kc, vc, o, dy.cuda(), lse, H, kp, causal, scale, drop_p=pdrop, drop_mask=mask)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(o, ref.detach()) < tol
    for got, want, name in ((dq, qf.grad, 'dq'), (dk, kf.grad, 'dk'), (dv, vf.grad, 'dv')):
        assert rel_err(got, back(want)) < tol, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_fully_masked_rows_and_strided_qkv(F, dtype):
    """a batch row whose keys are all padding outputs zeros (reference: nan_to_num, modules.py:765); q/k/v are
    column slices of one fused (B, T, 3D) projection buffer"""
    B, H, T = 2, 2, 12
    D = H * 64
    qkv = rnd((B, T, 3 * D), 35, dtype).cuda()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    key_pad = torch.zeros(B, T, dtype=torch.bool)
    key_pad[1, :] = True
    o, lse = F.attn_fwd(q, k, v, H, key_pad.cuda(), False, 0.125)
    assert o[1].abs().max().item() == 0
    o_ref, *_ = _attn_oracle(q.float().cpu(), k.float().cpu(), v.float().cpu(), H, key_pad, False, 0.125,
                             torch.zeros(B, T, D))
    assert rel_err(o, o_ref) < (1e-5 if dtype == torch.float32 else 2e-2)
    dq, dk, dv = F.attn_bwd(q, k, v, o, torch.ones_like(o), lse, H, key_pad.cuda(), False, 0.125)
    assert torch.isfinite(dq.float()).all() and dq[1].abs().max().item() == 0
    assert dk[1].abs().max().item() == 0 and dv[1].abs().max().item() == 0


def test_attention_online_softmax_rescale_branch(F):
    """force the running max to jump in a later key tile (cdna guide rule 26): spike one key against one query"""
    B, H, T, S = 1, 1, 40, 200
    q = rnd((B, T, 64), 36, torch.bfloat16)
    k = rnd((B, S, 64), 37, torch.bfloat16)
    v = rnd((B, S, 64), 38, torch.bfloat16)
    k[0, 150] = q[0, 7] * 4  # huge score for query 7 in the third key tile
    o_ref, *_ = _attn_oracle(q.float(), k.float(), v.float(), H, None, False, 0.125, torch.zeros(B, T, 64))
    o, _ = F.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, None, False, 0.125)
    assert rel_err(o, o_ref) < 2e-2
    assert (o[0, 7].float().cpu() - v[0, 150].float()).abs().max() < 0.05


# ------------------------------------------------------------------------------------------------------------
# embedding, cross-entropy, helpers
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_embedding_fwd_bwd(F, dtype):
    V, d, B, T = 101, 128, 5, 17
    E = rnd((V, d), 40, dtype, d ** -0.5)
    pos = O.sinusoidal_table(64, d, 2).to(dtype)
    ids = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(41))
    ids[0, 3] = 1
    ids[2, :5] = 7  # repeated ids -> several rows add into the same gradient row
    dout = rnd((B, T, d), 42, dtype)
    scale = math.sqrt(d)
    E32 = E.float().requires_grad_()
    ref = E32[ids] * scale + pos.float()[2 + 3: 2 + 3 + T][None]
    ref.backward(dout.float())
    gref = E32.grad.clone()
    gref[1] = 0  # padding_idx row receives no gradient (nn.Embedding(padding_idx=1), modules.py:905)
    out = F.embed_fwd(ids.cuda(), E.cuda(), pos.cuda(), scale, 2 + 3)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert rel_err(out, ref.detach()) < tol
    dE = F.embed_bwd(ids.cuda(), dout.cuda(), V, 1, scale)
    assert rel_err(dE, gref) < (1e-5 if dtype == torch.float32 else 8e-3)
    assert dE[1].abs().max().item() == 0
    # dropout: same mask forward and backward
    out_d = F.embed_fwd(ids.cuda(), E.cuda(), None, 1.0, 0, drop_p=0.3, seed=5, offset=11)
    kept = (out_d != 0) | (E.cuda()[ids.cuda()] == 0)
    assert abs(kept.float().mean().item() - 0.7) < 0.03
    ones = torch.ones(B, T, d, dtype=dtype).cuda()
    ids_u = torch.arange(B * T).view(B, T) % V  # B*T <= V: every token its own row
    if B * T <= V:
        out_u = F.embed_fwd(ids_u.cuda(), E.cuda(), None, 1.0, 0, drop_p=0.3, seed=5, offset=11)
        dE_u = F.embed_bwd(ids_u.cuda(), ones, V, -1, 1.0, drop_p=0.3, seed=5, offset=11)
        fw_kept = (out_u != 0).view(B * T, d)
        nz = E.cuda()[ids_u.cuda().view(-1)] != 0
        assert torch.equal((dE_u[:B * T] != 0) & nz, fw_kept & nz)


@pytest.mark.parametrize('dtype,V,d,ntok', [(torch.bfloat16, 8032, 512, 32768), (torch.float32, 1000, 1024, 3000),
                                            (torch.bfloat16, 70376, 2048, 5000), (torch.float16, 300, 136, 777)])
def test_embedding_backward_is_deterministic_and_exact(F, dtype, V, d, ntok):
    """the gradient of the lookup is a fixed-order sum per vocabulary row (stable sort of the token positions by id, no
    float atomics): two runs are bitwise equal, heavy rows (one id on a quarter of the tokens, like EOS / a language
    tag) and a pad id included, and the values match an fp32 index_add; widths of 1, 2 and 4 column blocks"""
    g = torch.Generator().manual_seed(V + d)
    ids = torch.randint(0, V, (ntok,), generator=g)
    ids[torch.rand(ntok, generator=g) < 0.25] = 2      # a heavy hitter
    ids[torch.rand(ntok, generator=g) < 0.05] = 1      # padding positions
    ids[:3] = torch.tensor([-5, V + 7, V - 1])         # out-of-range ids clamp like the forward kernel
    dout = (torch.randn(ntok, d, generator=g) * 0.5).to(dtype)
    runs = [F.embed_bwd(ids.cuda(), dout.cuda(), V, 1, 1.5, drop_p=0.1, seed=3, offset=9) for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    dE = F.embed_bwd(ids.cuda(), dout.cuda(), V, 1, 1.5)
    ref = torch.zeros(V, d, dtype=torch.float64)
    keep = ids != 1
    ref.index_add_(0, ids.clamp(0, V - 1)[keep], dout.double()[keep] * 1.5)
    ref[1] = 0
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(dE, ref.float()) < tol
    assert dE[1].abs().max().item() == 0
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[ids.clamp(0, V - 1)] = False
    assert dE[untouched.cuda()].abs().max().item() == 0 if untouched.any() else True
    empty = F.embed_bwd(ids[:0].cuda(), dout[:0].cuda(), V, 1, 1.0)
    assert empty.shape == (V, d) and empty.abs().max().item() == 0


@pytest.mark.parametrize('counts', [
    [640],                              # one id on every token: a run through ten workgroup ranges
    [64, 64, 64],                       # runs that end exactly on the range boundaries
    [63, 1, 64, 65, 1, 62],             # one off on either side of a boundary
    [1] * 70 + [17, 16, 200, 3],        # short runs, the long-run threshold, a run over three ranges, a ragged tail
    [5, 0, 130, 1],                     # an id without tokens in between
    [100, 300],                         # the pad id (row 1 below) on a run that crosses ranges
])
def test_embedding_backward_runs_across_workgroup_ranges(F, counts):
    """the sorted token positions are cut into ranges of 64 per workgroup and a vocabulary row seen more often than that is
    summed in pieces (partials of the cut runs added in range order by a second kernel): run layouts around every edge
    of that scheme, against an fp64 index_add, bitwise reproducible"""
    V, d = len(counts) + 2, 256
    ids = torch.cat([torch.full((c,), i, dtype=torch.long) for i, c in enumerate(counts)])
    ids = ids[torch.randperm(ids.numel(), generator=torch.Generator().manual_seed(len(counts)))]
    dout = torch.randn(ids.numel(), d, generator=torch.Generator().manual_seed(7)).bfloat16()
    pad = 1 if counts == [100, 300] else V - 1
    a = F.embed_bwd(ids.cuda(), dout.cuda(), V, pad, 0.5)
    b = F.embed_bwd(ids.cuda(), dout.cuda(), V, pad, 0.5)
    assert torch.equal(a, b)
    ref = torch.zeros(V, d, dtype=torch.float64)
    ref.index_add_(0, ids, dout.double() * 0.5)
    ref[pad] = 0
    assert rel_err(a, ref.float()) < 6e-3
    assert a[pad].abs().max().item() == 0


@pytest.mark.parametrize('dtype,V,d,ntok', [(torch.bfloat16, 8032, 512, 32768), (torch.float32, 1000, 1024, 3000),
                                            (torch.float16, 300, 136, 777), (torch.bfloat16, 50, 256, 700)])
def test_embedding_backward_added_into_an_existing_gradient(F, dtype, V, d, ntok):
    """pk_embed_bwd_acc (the tied table: the lookup's rows added into the projection's dense dW): every row with tokens =
    fp32(existing) + the fixed-order sum, rounded once — within one rounding of existing + pk_embed_bwd's own result —, rows
    without tokens and the pad row bit for bit what they were, cut runs (a heavy hitter across workgroup ranges) included,
    reproducible bit for bit, and the empty batch leaves everything alone"""
    g = torch.Generator().manual_seed(V * 3 + d)
    ids = torch.randint(0, V, (ntok,), generator=g)
    ids[torch.rand(ntok, generator=g) < 0.25] = 2
    ids[torch.rand(ntok, generator=g) < 0.05] = 1
    dout = (torch.randn(ntok, d, generator=g) * 0.5).to(dtype).cuda()
    base = torch.randn(V, d, generator=g).to(dtype).cuda()
    outs = []
    for _ in range(2):
        acc = base.clone()
        assert F.embed_bwd(ids.cuda(), dout, V, 1, 1.5, drop_p=0.1, seed=3, offset=9, into=acc) is acc
        outs.append(acc)
    assert torch.equal(outs[0], outs[1])
    acc = base.clone()
    F.embed_bwd(ids.cuda(), dout, V, 1, 1.5, into=acc)
    ref = base.double().cpu()
    keep = ids != 1
    ref.index_add_(0, ids[keep], dout.double().cpu()[keep] * 1.5)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(acc, ref.float()) < tol
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[ids] = False
    untouched[1] = True
    assert torch.equal(acc[untouched.cuda()], base[untouched.cuda()])
    alone = F.embed_bwd(ids.cuda(), dout, V, 1, 1.5)
    two_roundings = (base.float() + alone.float()).to(dtype)
    ulp = 2.0 ** -23 if dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
    scale = torch.maximum(torch.maximum(base.float().abs(), alone.float().abs()), acc.float().abs()).clamp_min(1e-3)
    assert ((acc.float() - two_roundings.float()).abs() <= 2.5 * ulp * scale).all()
    same = base.clone()
    F.embed_bwd(ids[:0].cuda(), dout[:0], V, 1, 1.0, into=same)
    assert torch.equal(same, base)


@pytest.mark.parametrize('ntok,V', [(1, 50), (2, 8032), (1000, 8032), (4097, 600), (32768, 8032), (32768, 70376), (8192, 256206),
                                    (32767, 131072), (40000, 8032)])
def test_embedding_backward_sort_sizes(F, ntok, V):
    """token counts from 1 to 40 000 over vocabularies of 50 to 256 206 rows (the key widths of the position sort), one hot row,
    the first and the last id present — against index_add in fp64, twice for the bits.  (Written for a single-workgroup
    bitonic sort that was measured and dropped: 120 stages over 128 KiB of LDS are bound by one CU's LDS bandwidth, 270-350 us
    against the radix sort's 45.)"""
    g = torch.Generator().manual_seed(ntok + V)
    d = 64
    ids = torch.randint(0, V, (ntok,), generator=g)
    ids[: ntok // 3] = int(ids[0])  # one hot row
    if ntok > 2:
        ids[-1], ids[-2] = V - 1, 0
    dout = torch.randn(ntok, d, generator=g).bfloat16()
    a = F.embed_bwd(ids.cuda(), dout.cuda(), V, -1, 1.0)
    b = F.embed_bwd(ids.cuda(), dout.cuda(), V, -1, 1.0)
    assert torch.equal(a, b)
    ref = torch.zeros(V, d, dtype=torch.float64)
    ref.index_add_(0, ids, dout.double())
    assert rel_err(a, ref.float()) < 6e-3
    touched = torch.zeros(V, dtype=torch.bool)
    touched[ids] = True
    assert a[~touched.cuda()].abs().max().item() == 0 if (~touched).any() else True


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('eps', [0.0, 0.1])
@pytest.mark.parametrize('V', [8032, 101, 70376])
def test_cross_entropy_rows(F, dtype, eps, V):
    rows = 37
    logits = rnd((rows, V), 50, dtype, 2.0)
    target = torch.randint(4, V, (rows,), generator=torch.Generator().manual_seed(51))
    target[5] = 1
    target[20:23] = 1
    x = logits.float().requires_grad_()
    loss, nll, ntok = O.label_smoothed_ce(x, target, 1, eps)
    loss.backward()
    row_loss = torch.empty(rows, device='cuda')
    row_nll = torch.empty(rows, device='cuda')
    dl = torch.empty(rows, V, dtype=dtype, device='cuda')
    F.ce_rows(logits.cuda(), target.cuda(), 1, eps, row_loss, row_nll, dlogits=dl)
    sums = F.ce_finalize(row_loss, row_nll, target.cuda(), 1).cpu()
    assert abs(sums[0].item() - loss.item()) <= 2e-6 * abs(loss.item())
    assert abs(sums[1].item() - nll.item()) <= 2e-6 * abs(nll.item())
    assert int(sums[2].item()) == int(ntok)
    assert rel_err(dl, x.grad) < (1e-5 if dtype == torch.float32 else 8e-3)
    assert dl[5].abs().max().item() == 0
    # in-place variant (dlogits aliases logits)
    lg = logits.cuda().clone()
    F.ce_rows(lg, target.cuda(), 1, eps, row_loss, row_nll, dlogits=lg)
    assert torch.equal(lg, dl)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('V,ld', [(1000, 1000), (8192, 8192), (32000, 32000), (50004, 50008), (98304, 98304), (98312, 98312),
                                   (256206, 256208), (131080, 131080), (300000, 300000)])
def test_cross_entropy_rows_register_resident(F, dtype, V, ld):
    """16-bit rows of up to 98 304 columns stay in registers between the statistics and the gradient pass (`ce_reg_kernel`:
    4 vectors x 256 threads, 4 / 12 vectors x 1024 threads); wider rows (NLLB's 256 206) take the two-pass kernel as ONE
    1024-thread workgroup per CU (round 5: the rows in flight then fit the Infinity Cache), fp32 the 256-thread two-pass kernel.
    Every instantiation and both sides of each boundary, a row pitch padded to 8 with a scalar tail (50 004 in rows of 50 008,
    256 206 in 256 208: the pad columns must come back as zeros), padding targets, the gradient written over the logits —
    against the oracle's label-smoothed cross-entropy in fp32 (transformer.py:324-380)."""
    rows, eps = 9, 0.1
    g = torch.Generator().manual_seed(V)
    buf = torch.full((rows, ld), float('nan'), dtype=dtype)
    buf[:, :V] = (torch.randn(rows, V, generator=g) * 2.0).to(dtype)
    target = torch.randint(4, V, (rows,), generator=g)
    target[2] = 1           # padding position: zero loss, zero gradient row
    target[3] = V - 1       # the last column (inside the scalar tail where there is one)
    target[4] = 0
    x = buf[:, :V].float().requires_grad_()
    loss, nll, ntok = O.label_smoothed_ce(x, target, 1, eps)
    loss.backward()
    lg = buf.cuda()
    row_loss, row_nll = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    F.ce_rows(lg[:, :V], target.cuda(), 1, eps, row_loss, row_nll, dlogits=lg[:, :V])
    sums = F.ce_finalize(row_loss, row_nll, target.cuda(), 1).cpu()
    assert abs(sums[0].item() - loss.item()) <= 2e-6 * abs(loss.item())
    assert abs(sums[1].item() - nll.item()) <= 2e-6 * abs(nll.item())
    assert int(sums[2].item()) == int(ntok)
    assert rel_err(lg[:, :V], x.grad) < (8e-3 if dtype == torch.bfloat16 else 1e-3)
    assert lg[2, :V].abs().max().item() == 0
    if ld > V:
        assert (lg[:, V:(V + 7) // 8 * 8] == 0).all()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,act', [(16000, 'relu'), (2048, 'relu'), (777, 'gelu'), (2048, 'none')])
def test_few_rows_kernel_with_the_activation_derivative_epilogue(F, dtype, M, act):
    """an adapter's dA = s (dY W_up) * act'(a) — a contraction over d into 64 columns — on the few-rows kernel (pk_gemm mode 2
    there since round 5): against fp64, bit for bit the tiled kernel's result for the (row, col) form of the same product,
    and the launch sampling says which kernel ran (tag 64)"""
    import ctypes
    from pasero_amd import lib
    g = torch.Generator().manual_seed(M)
    dy = (torch.randn(M, 1024, generator=g) * 0.5).to(dtype).cuda()
    w_up = (torch.randn(1024, 64, generator=g) * 0.05).to(dtype).cuda()
    aux = torch.randn(M, 64, generator=g).to(dtype).cuda()
    if act == 'relu':
        aux = aux.clamp_min(0)
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(8, 1), 'start')
    kw = dict(act=act, aux=aux, mode=2) if act != 'none' else {}
    out = F.gemm(dy, w_up.t().contiguous(), alpha=0.5, **kw)
    n = L.pk_gemm_timing_stop()
    ints = [ctypes.c_int() for _ in range(5)]
    fl, ms = ctypes.c_double(), ctypes.c_float()
    lib.check(L.pk_gemm_timing_read(0, *[ctypes.byref(x) for x in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
    assert n == 1 and ints[0].value == 64, (n, ints[0].value)
    tiled = F.gemm(dy, w_up, b_col=True, alpha=0.5, **kw)
    ref = 0.5 * (dy.double() @ w_up.double())
    if act == 'relu':
        ref = ref * (aux > 0)
    elif act == 'gelu':
        x = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -9
    assert (out.double() - ref).abs().max().item() <= tol * ref.abs().max().item()
    assert (out.double() - tiled.double()).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,L,C,k,stride', [(3, 50, 64, 3, 2), (2, 301, 512, 3, 1), (2, 100, 40, 5, 2), (1, 37, 12, 5, 2), (4, 64, 6, 3, 2)])
def test_col2im1d(F, dtype, B, L, C, k, stride):
    """pk_col2im1d (the input gradient of the channels-last Conv1d, pasero/models/modules.py:793-799 backward): the 16-byte
    chunk kernel (C a multiple of 8 / 4) and the scalar one (C = 12 in bf16, 6) against an explicit scatter of the windows in fp64"""
    pad = k // 2
    Lout = (L + 2 * pad - k) // stride + 1
    R = -(-(L + 2 * pad) // stride)
    g = torch.Generator().manual_seed(B * L + C)
    dA = torch.randn(B * R, k * C, generator=g).to(dtype)
    ref = torch.zeros(B, L + 2 * pad + stride * R, C, dtype=torch.float64)
    for r in range(Lout):
        for j in range(k):
            ref[:, r * stride + j] += dA.view(B, R, k, C)[:, r, j].double()
    ref = ref[:, pad:pad + L]
    out = F.col2im1d(dA.cuda(), B, L, C, R, Lout, k, stride, pad)
    assert out.shape == (B, L, C)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert (out.double().cpu() - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,L,C,chans,ks,strides', [
    (3, 300, 80, (96, 64), (3, 3), (1, 2)),        # Whisper's stack: conv 0 writes straight into the padded input of conv 1
    (2, 301, 40, (64, 64), (3, 3), (1, 2)),        # odd length
    (2, 120, 32, (48, 40), (3, 5), (1, 1)),        # the second conv's windows reach past the end of its input (k 5, stride 1)
    (2, 100, 32, (48, 40, 24), (3, 3, 3), (1, 1, 2)),
    (2, 90, 32, (48, 40), (3, 3), (2, 2)),         # no straight write (stride 2 in front): the copies stay
])
def test_conv_stack_is_the_chain_of_convs_bit_for_bit(dtype, B, L, C, chans, ks, strides):
    """autograd.ConvStackFn (one node: conv i's GEMM writes into the padded input of conv i + 1, pk_col2im1d writes the gradient in
    the row layout the conv below contracts) against the chain of Conv1dChannelsLastFn nodes it replaces: the same GEMMs on the
    same operands, so outputs and every gradient are equal bit for bit — and against torch's conv1d in fp64"""
    from pasero_amd.autograd import Conv1dChannelsLastFn, ConvStackFn
    g = torch.Generator().manual_seed(L + C)
    x0 = torch.randn(B, L, C, generator=g).to(dtype).cuda()
    ws, bs, cin = [], [], C
    for O, k in zip(chans, ks):
        ws.append((torch.randn(O, cin, k, generator=g) / (cin * k) ** 0.5).to(dtype).cuda())
        bs.append((0.1 * torch.randn(O, generator=g)).to(dtype).cuda())
        cin = O
    outs = []
    for stack in (True, False):
        x = x0.clone().requires_grad_()
        w = [t.clone().requires_grad_() for t in ws]
        b = [t.clone().requires_grad_() for t in bs]
        if stack:
            y = ConvStackFn.apply(x, 'gelu', tuple((s, k // 2) for s, k in zip(strides, ks)), *[t for p in zip(w, b) for t in p])
        else:
            y = x
            for wi, bi, s, k in zip(w, b, strides, ks):
                y = Conv1dChannelsLastFn.apply(y, wi, bi, s, k // 2, 'gelu')
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dtype).cuda()
        y.backward(dy)
        outs.append([y.detach(), x.grad] + [t.grad for t in w] + [t.grad for t in b])
    for a, r in zip(*outs):
        assert a.shape == r.shape and torch.equal(a, r)
    x = x0.double().cpu().transpose(1, 2).requires_grad_()
    y = x
    for wi, bi, s, k in zip(ws, bs, strides, ks):
        y = torch.nn.functional.gelu(torch.nn.functional.conv1d(y, wi.double().cpu(), bi.double().cpu(), stride=s, padding=k // 2))
    y.transpose(1, 2).backward(dy.double().cpu())
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert rel_err(outs[0][0], y.transpose(1, 2)) < tol
    assert rel_err(outs[0][1], x.grad.transpose(1, 2)) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N', [(1000, 512), (7, 2048), (33000, 1536), (50, 100), (3, 9)])
def test_colsum(F, dtype, M, N):
    x = rnd((M, N), 60, dtype)
    out = F.colsum(x.cuda())
    ref = x.double().sum(0)
    assert ((out.double().cpu() - ref).abs().max() / ref.abs().max()).item() < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_and_scale(F, dtype):
    n = 100003
    x = rnd((n,), 70, dtype).cuda()
    y = F.dropout(x, 0.25, 9, 3)
    kept = y != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    assert rel_err(y[kept], x[kept].float() / 0.75) < 1e-2
    y2 = F.dropout(x, 0.25, 9, 3)
    assert torch.equal(y, y2)
    s = torch.tensor([3.0], device='cuda')
    z = F.scale(x, s, 0.5)
    assert rel_err(z, x.float() * 1.5) < (1e-6 if dtype == torch.float32 else 8e-3)


# ------------------------------------------------------------------------------------------------------------
# K8: Whisper log-mel
# ------------------------------------------------------------------------------------------------------------
def test_log_mel_vs_golden_and_oracle(F):
    """golden = transformers.WhisperFeatureExtractor (the third-party code the reference calls); oracle = numpy fp64.
    fp32 DFT on the matrix cores vs fp64: features are O(1) on a log10 scale -> 5e-4 absolute"""
    from conftest import load_golden
    g = load_golden('logmel')
    w0, w1 = g['wav0'], g['wav1']
    n = max(len(w0), len(w1))
    wav = torch.zeros(2, n)
    wav[0, :len(w0)] = torch.from_numpy(w0)
    wav[1, :len(w1)] = torch.from_numpy(w1)
    lens = torch.tensor([len(w0), len(w1)])
    out = F.log_mel(wav.cuda(), lens.cuda()).cpu().numpy()
    assert out.shape == (2, 3000, 80)
    for i, w in enumerate((w0, w1)):
        ref = O.log_mel(w)
        assert np.abs(out[i] - ref).max() < 5e-4
        assert np.abs(out[i, :240] - g['feats'][i]).max() < 5e-4
        assert np.abs(out[i, -4:] - g['feats_tail'][i]).max() < 5e-4


def test_log_mel_full_30s_clip_and_truncation(F):
    """maximum size: a clip longer than 30 s is truncated to 480000 samples; reflect padding at both ends"""
    gen = torch.Generator().manual_seed(80)
    wav = 0.1 * torch.randn(2, 500000, generator=gen)
    out = F.log_mel(wav.cuda()).cpu().numpy()
    ref = O.log_mel(wav[1].numpy())
    assert np.abs(out[1] - ref).max() < 5e-4
    assert np.isfinite(out).all()
    silent = torch.zeros(1, 16000)
    o2 = F.log_mel(silent.cuda()).cpu().numpy()
    assert np.allclose(o2, (np.log10(1e-10) + 4) / 4)  # all-zero input: every bin clamps at 1e-10


# ------------------------------------------------------------------------------------------------------------
# "next" row: fused clip + Adam
# ------------------------------------------------------------------------------------------------------------
def test_fused_adam_and_clip_vs_reference():
    """optimization.clip_grad_norm_ + optimization.Adam.step of the reference (golden: adam_step), 3 steps"""
    from conftest import load_golden
    from pasero_amd.optim import Adam
    g = load_golden('adam_step')
    n, steps = int(g['n']), int(g['steps'])
    params = [torch.nn.Parameter(torch.from_numpy(g[f'p0:{i}']).cuda()) for i in range(n)]
    opt = Adam(params, lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    for s in range(steps):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(g[f'g{s}:{i}']).cuda()
        gnorm = opt.fused_step(scale=1.0, max_norm=1.0)
        assert abs(gnorm.item() - float(g[f'gnorm{s}'])) <= 1e-5 * float(g[f'gnorm{s}'])
        for i, p in enumerate(params):
            assert rel_err(p.detach(), torch.from_numpy(g[f'p{s + 1}:{i}'])) < 1e-5


def test_fused_adam_bf16_scale_and_state_layout():
    """bf16 parameters: fp32 moments, update through fp32, one rounding (optimization.py:88-149); `scale` = the
    Trainer's dp_size / num_tokens factor (training.py:455-470)"""
    from pasero_amd.optim import Adam
    p0 = rnd((300, 70), 90, torch.float32)
    g0 = rnd((300, 70), 91, torch.float32)
    pb = torch.nn.Parameter(p0.bfloat16().cuda())
    pb.grad = g0.bfloat16().cuda()
    opt = Adam([pb], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0)
    gnorm = opt.fused_step(scale=0.25, max_norm=0.0)
    gq = g0.bfloat16().float() * 0.25
    ref, m, v = O.adam_step(p0.bfloat16().float(), gq, torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-2, 0.9, 0.98,
                            1e-8, 0.0)
    assert abs(gnorm.item() - gq.double().norm().item()) <= 1e-4 * gq.norm().item()
    assert torch.equal(pb.detach().cpu(), ref.bfloat16())
    st = opt.state[pb]
    assert set(st) == {'step', 'exp_avg', 'exp_avg_sq'} and st['step'] == 1
    assert st['exp_avg'].dtype == torch.float32 and rel_err(st['exp_avg'], m) < 1e-6 and rel_err(st['exp_avg_sq'], v) < 1e-6


def test_fused_adam_one_global_norm_and_per_parameter_steps():
    """two parameter groups, fp32 and bf16 parameters, one parameter without a gradient at the second step: the clip
    coefficient comes from ONE norm over every gradient (optimization.py:404-424), a parameter that skipped a step keeps
    its own `state['step']` and bias correction (optimization.py:72-76,120-127), its moments untouched"""
    from pasero_amd.optim import Adam
    shapes = [(64, 40), (300,), (128, 96), (1000,)]
    dts = [torch.float32, torch.bfloat16, torch.bfloat16, torch.float32]
    p0 = [rnd(s, 200 + i, torch.float32).to(dt) for i, (s, dt) in enumerate(zip(shapes, dts))]
    params = [torch.nn.Parameter(p.clone().cuda()) for p in p0]
    opt = Adam([{'params': params[:2], 'lr': 1e-2}, {'params': params[2:], 'lr': 5e-3, 'weight_decay': 0.0}],
               betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    ref_p = [p.float() for p in p0]
    ref_m = [torch.zeros_like(p) for p in ref_p]
    ref_v = [torch.zeros_like(p) for p in ref_p]
    ref_step = [0] * 4
    lr, wd = [1e-2, 1e-2, 5e-3, 5e-3], [0.01, 0.01, 0.0, 0.0]
    for s in range(3):
        active = [0, 1, 2, 3] if s != 1 else [0, 2, 3]  # parameter 1 gets no gradient at step 1
        grads = {i: rnd(shapes[i], 300 + 10 * s + i, torch.float32).to(dts[i]) * 3.0 for i in active}
        for i, p in enumerate(params):
            p.grad = grads[i].cuda() if i in grads else None
        gnorm = opt.fused_step(scale=0.5, max_norm=1.0)
        total = torch.sqrt(sum((grads[i].double() * 0.5).pow(2).sum() for i in active))
        assert abs(gnorm.item() - total.item()) <= 1e-5 * total.item()
        coef = 0.5 * min(1.0, 1.0 / (total.item() + 1e-6))
        for i in active:
            ref_step[i] += 1
            ref_p[i], ref_m[i], ref_v[i] = O.adam_step(ref_p[i], grads[i].float() * coef, ref_m[i], ref_v[i],
                                                       ref_step[i], lr[i], 0.9, 0.98, 1e-8, wd[i])
            ref_p[i] = ref_p[i].to(dts[i]).float()
        for i, p in enumerate(params):
            st = opt.state[p]
            assert st.get('step', 0) == ref_step[i], (s, i)
            tol = 1e-5 if dts[i] == torch.float32 else 1e-2
            assert rel_err(p.detach().float(), ref_p[i]) < tol, (s, i)
            if ref_step[i]:
                assert rel_err(st['exp_avg'], ref_m[i]) < 2e-5 and rel_err(st['exp_avg_sq'], ref_v[i]) < 2e-5, (s, i)


def test_mixed_dtypes_and_wrong_sizes_are_refused_on_the_host(F):
    """the kernels trust the element type and sizes they are told: an fp32 LayerNorm weight next to bf16 activations,
    int32 token ids or 16-bit optimizer moments would be read with the wrong stride or past their end — refused before
    any launch"""
    x = torch.randn(8, 128, device='cuda').bfloat16()
    g32 = torch.ones(128, device='cuda')
    with pytest.raises(TypeError):
        F.residual_ln_fwd(x, None, g32, None, 1e-5)
    with pytest.raises(AssertionError):
        F.residual_ln_fwd(x, None, g32.bfloat16()[:64].contiguous(), None, 1e-5)
    q = torch.randn(2, 4, 64, device='cuda').bfloat16()
    with pytest.raises(TypeError):
        F.attn_fwd(q, q.float(), q, 1, None, False, 0.125)
    logits = torch.randn(4, 32, device='cuda').bfloat16()
    rl, rn = torch.empty(4, device='cuda'), torch.empty(4, device='cuda')
    with pytest.raises(AssertionError):
        F.ce_rows(logits, torch.zeros(4, dtype=torch.int32, device='cuda'), 1, 0.1, rl, rn)
    with pytest.raises(AssertionError):
        F.ce_rows(logits, torch.zeros(4, dtype=torch.int64, device='cuda'), 1, 0.1, rl[:3], rn)
    from pasero_amd.optim import Adam
    p = torch.nn.Parameter(torch.randn(64, 64, device='cuda').bfloat16())
    opt = Adam([p], lr=1e-3)
    p.grad = torch.randn_like(p)
    opt.step()
    opt.state[p]['exp_avg'] = opt.state[p]['exp_avg'].bfloat16()  # what torch's load_state_dict would have done
    with pytest.raises(RuntimeError, match='fp32'):
        opt.step()
