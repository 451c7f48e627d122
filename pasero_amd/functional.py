"""Thin, non-differentiable wrappers over the C ABI (one Python function per kernel entry point).
They only do argument checking, output allocation and the ctypes call; autograd lives in `pasero_amd.autograd`.
"""
import ctypes
from typing import Optional

import torch
from torch import Tensor

from . import lib
from .lib import ACT, check, dtype_code, ptr, require_gpu, stream_ptr


def _same(ref: Tensor, *tensors, what: str) -> None:
    """The kernels index every tensor of a call with ONE element type and trust the sizes they are given: a tensor of
    another dtype would be read with the wrong stride — or past its end (an fp32 LayerNorm weight next to bf16
    activations, optimizer moments cast to 16 bits).  Checked here, on the host, for every multi-tensor entry point."""
    for t in tensors:
        if t is not None and t.dtype != ref.dtype:
            raise TypeError(f'pasero_amd: {what}: tensors of dtype {t.dtype} and {ref.dtype} in one kernel call '
                            f'(cast the module with .to(dtype), or run under torch.autocast)')


def _ld(t: Tensor) -> int:
    assert t.dim() == 2 and (t.stride(1) == 1 or t.size(1) == 1), 'operand must be 2-D with unit inner stride'
    return t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0))


def gemm(a: Tensor, b: Tensor, *, a_col: bool = False, b_col: bool = False, bias: Optional[Tensor] = None,
         aux: Optional[Tensor] = None, act: str = 'none', mode: int = 0, alpha: float = 1.0,
         out: Optional[Tensor] = None, preact: Optional[Tensor] = None, splitk: int = 1,
         asum_out: Optional[Tensor] = None, pad_n: bool = False, pad_k: bool = False) -> Tensor:
    """C[m,n] = epi(alpha * sum_k A(m,k) B(n,k)).  `a` is [M,K] (or [K,M] if a_col), `b` is [N,K] (or [K,N] if b_col).
    See include/pasero_hip.h:pk_gemm for the epilogue modes; pad_n / pad_k: the promises of pk_gemm_ex (the rows of `out`
    have room for N rounded up to 8 / the rows of a row-form `a` continue with zeros up to K rounded up to 8)."""
    require_gpu(a, b, bias, aux, out, preact)
    M, K = (a.size(1), a.size(0)) if a_col else (a.size(0), a.size(1))
    N, Kb = (b.size(1), b.size(0)) if b_col else (b.size(0), b.size(1))
    assert K == Kb, f'contraction mismatch {K} vs {Kb}'
    assert a.dtype == b.dtype
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    assert out.shape == (M, N) and out.dtype == a.dtype
    if bias is not None:
        assert bias.dtype == a.dtype and bias.numel() == N and bias.is_contiguous()
    if aux is not None:
        assert aux.dtype == a.dtype and aux.shape == (M, N)
    if preact is not None:
        assert preact.dtype == a.dtype and preact.shape == (M, N)
    if asum_out is not None:
        assert a_col and asum_out.dtype == a.dtype and asum_out.numel() == M and asum_out.is_contiguous()
    ws, ws_bytes = None, 0
    if splitk > 1:
        ws_bytes = 2 * splitk * M * (N + 1) * 4  # room for the 256-tile kernel's (up to 2x) larger split factor
        ws = lib.workspace(ws_bytes, a.device, 'splitk')
    L = lib.load()
    if pad_n or pad_k:
        flags = (lib.PK_GEMM_PAD_N if pad_n else 0) | (lib.PK_GEMM_PAD_K if pad_k else 0)
        check(L.pk_gemm_ex(ptr(a), ptr(b), ptr(out), ptr(bias), ptr(aux), ptr(preact), M, N, K, _ld(a), _ld(b), _ld(out),
                           _ld(aux) if aux is not None else 0, _ld(preact) if preact is not None else 0,
                           int(a_col), int(b_col), ACT[act], mode, float(alpha), dtype_code(a), int(splitk),
                           ptr(ws), ws_bytes, ptr(asum_out), flags, stream_ptr()), 'pk_gemm_ex')
        return out
    check(L.pk_gemm(ptr(a), ptr(b), ptr(out), ptr(bias), ptr(aux), ptr(preact), M, N, K, _ld(a), _ld(b), _ld(out),
                    _ld(aux) if aux is not None else 0, _ld(preact) if preact is not None else 0,
                    int(a_col), int(b_col), ACT[act], mode, float(alpha), dtype_code(a), int(splitk),
                    ptr(ws), ws_bytes, ptr(asum_out), stream_ptr()), 'pk_gemm')
    return out


def relu_bits_eligible(x: Tensor, w1: Tensor) -> bool:
    """can the ReLU feed-forward over x [M, K] with fc1 weight w1 [f, K] keep its mask as one bit per element
    (pk_gemm_relu_bits: fc1 forward writes it, the dH GEMM of backward reads it instead of the activations)?"""
    if not (x.is_cuda and w1.is_cuda) or x.dtype != w1.dtype or x.dtype not in (torch.bfloat16, torch.float16):
        return False
    if x.dim() != 2 or w1.dim() != 2 or x.size(1) != w1.size(1) or x.stride(1) != 1 or w1.stride(1) != 1 or w1.size(0) % 32:
        return False
    M, K, N = x.size(0), x.size(1), w1.size(0)
    L = lib.load()
    # forward shape (row-form W1) and backward shape (dZ [M, K] times col-form W2 [K, N]): same sizes, other pointers
    return bool(L.pk_gemm_relu_bits_eligible(ptr(x), ptr(w1), ptr(x), None, M, N, K, _ld(x), _ld(w1), N, N // 8, 0, 0,
                                             dtype_code(x))) and \
        bool(L.pk_gemm_relu_bits_eligible(ptr(x), ptr(w1), ptr(x), None, M, N, K, _ld(x), N, N, N // 8, 1, 2, dtype_code(x)))


def gemm_relu_bits(x: Tensor, w1: Tensor, bias: Optional[Tensor]):
    """h = relu(x w1ᵀ + bias) and its mask as bits [M, f / 8] uint8"""
    require_gpu(x, w1, bias)
    _same(x, w1, bias, what='gemm_relu_bits')
    M, K = x.shape
    N = w1.size(0)
    h = torch.empty(M, N, dtype=x.dtype, device=x.device)
    bits = torch.empty(M, N // 8, dtype=torch.uint8, device=x.device)
    check(lib.load().pk_gemm_relu_bits(ptr(x), ptr(w1), ptr(h), ptr(bias), ptr(bits), M, N, K, _ld(x), _ld(w1), N, N // 8, 0, 0,
                                       1.0, dtype_code(x), stream_ptr()), 'pk_gemm_relu_bits')
    return h, bits


def gemm_mask_bits(dy: Tensor, w2: Tensor, bits: Tensor) -> Tensor:
    """dh = (dy w2) where the bit is set, 0 elsewhere; dy [M, K], w2 [K, f] (fc2's weight: col form), bits [M, f / 8]"""
    require_gpu(dy, w2, bits)
    _same(dy, w2, what='gemm_mask_bits')
    M, K = dy.shape
    N = w2.size(1)
    assert w2.size(0) == K and bits.shape == (M, N // 8) and bits.dtype == torch.uint8 and bits.is_contiguous()
    dh = torch.empty(M, N, dtype=dy.dtype, device=dy.device)
    check(lib.load().pk_gemm_relu_bits(ptr(dy), ptr(w2), ptr(dh), None, ptr(bits), M, N, K, _ld(dy), _ld(w2), N, N // 8, 1, 2,
                                       1.0, dtype_code(dy), stream_ptr()), 'pk_gemm_relu_bits')
    return dh


def _wgrad_problem(dy: Tensor, x: Tensor, dw: Optional[Tensor], db: Optional[Tensor]) -> 'lib.PkWgradProblem':
    return lib.PkWgradProblem(ptr(dy), ptr(x), ptr(dw), ptr(db), dy.size(1), x.size(1), dy.size(0), _ld(dy), _ld(x),
                              _ld(dw) if dw is not None else x.size(1))


def wgrad_group_eligible(dy: Tensor, x: Tensor) -> bool:
    """Can dW = dyᵀ·x (dy [rows, N_out], x [rows, K_in]) ride in a grouped launch (pk_gemm_wgrad_group)?  The output is a
    fresh contiguous tensor, so only the operands decide."""
    if not (dy.is_cuda and x.is_cuda) or dy.dtype != x.dtype or dy.dtype not in (torch.bfloat16, torch.float16):
        return False
    if dy.dim() != 2 or x.dim() != 2 or dy.size(0) != x.size(0) or dy.stride(1) != 1 or x.stride(1) != 1:
        return False
    q = _wgrad_problem(dy, x, None, None)
    q.C = 16  # (placeholder: any 16-byte aligned non-null address; the real output is allocated at launch time)
    return bool(lib.load().pk_gemm_wgrad_group_eligible(ctypes.byref(q), dtype_code(dy)))


def wgrad_group(entries):
    """entries: [(dy [rows, N_out], x [rows, K_in], want_bias)], every one `wgrad_group_eligible` and of one dtype.
    Returns [(dW [N_out, K_in], db [N_out] or None)]: all weight gradients in ONE GEMM launch + one reduction launch per
    PK_WGRAD_MAX problems (include/pasero_hip.h:pk_gemm_wgrad_group)."""
    L = lib.load()
    out = []
    for i in range(0, len(entries), lib.PK_WGRAD_MAX):
        part = entries[i:i + lib.PK_WGRAD_MAX]
        ref = part[0][0]
        require_gpu(*[t for e in part for t in e[:2]])
        _same(ref, *[t for e in part for t in e[:2]], what='wgrad_group')
        res = []
        for dy, x, want_b in part:
            dw = torch.empty(dy.size(1), x.size(1), dtype=dy.dtype, device=dy.device)
            db = torch.empty(dy.size(1), dtype=dy.dtype, device=dy.device) if want_b else None
            res.append((dw, db))
        arr = (lib.PkWgradProblem * len(part))(*[_wgrad_problem(dy, x, dw, db)
                                                 for (dy, x, _), (dw, db) in zip(part, res)])
        ws_bytes = L.pk_gemm_wgrad_group_workspace(arr, len(part))
        ws = lib.workspace(ws_bytes, ref.device, 'splitk') if ws_bytes else None
        check(L.pk_gemm_wgrad_group(arr, len(part), dtype_code(ref), ptr(ws), ws_bytes, stream_ptr()),
              'pk_gemm_wgrad_group')
        out.extend(res)
    return out


def fwd_split(M: int, N: int, K: int, dtype) -> int:
    """csrc/layer.cpp: fwd_split — forward GEMMs are not split as a rule; the exception is a projection back to d from a long
    contraction at a few thousand rows (NLLB-1.3B's fc2, 8192 -> 1024, at the IWSLT recipe's 2048-row decoder batch: 32 tiles of
    256 x 256, 123 us on the 128-tile kernel, ~40 as K-slabs + reduction).  16-bit types only (fp32 is the parity path).
    The rule gates on M, and pk_gemm re-derives its own slab count from the output's tile count: the number of partial sums of
    a row's contraction therefore depends on the batch's row count (one chain at 4096 rows, 8 slabs of the 256-tile kernel at
    2048, 4 of the 128-tile kernel at 1024).  The same row in batches of different sizes agrees to the fp32 round-off of the
    accumulation — at most one ulp of the 16-bit output — never bitwise by construction
    (tests/test_native_layer_gpu.py::test_forward_split_boundary_moves_rows_by_round_off_only pins that deviation).  The
    native layer applies the same rule (the two paths stay bit for bit equal)."""
    if dtype == torch.float32:
        return 1
    return 4 if (K >= 4096 and K % 512 == 0 and N <= 1024 and N % 256 == 0 and 512 <= M <= 2048 and M % 256 == 0) else 1


def choose_splitk(M: int, N: int, K: int, target_blocks: int = 512) -> int:
    """Split the contraction when the output has too few 128x128 tiles to fill 256 CUs (weight-gradient GEMMs)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > 256 or K < 1024:
        # an output of 80..159 256-tiles (C5: 8192 x 1024 = 128) fills half the chip with the 256-tile kernel and runs
        # as one low-occupancy round of 128-tiles: two K slices on the 256 kernel are faster (8192 x 1024 x 8192, cold
        # operands: weight gradient 201 -> 171 us, dX 185 -> 166 us, split-K reduction included)
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        # (K % 8: the phase-interleaved kernel zero-fills a partial last K-tile — the vocabulary dX GEMMs, K = V.  With
        # K % 64 required, a 6144- or 8192-row chunk of transformer_big's dX GEMM ran unsplit on 96 / 128 of the 256 CUs:
        # 1.5 ms instead of 0.8)
        return 2 if (80 <= t256 < 160 and K >= 2048 and M % 8 == 0 and N % 8 == 0 and K % 8 == 0) else 1
    # tiles * s must stay within one round of 2 workgroups per CU (512): 560 workgroups run as two rounds and take
    # 1.6x as long as 504.  256 tiles are still split in two — one workgroup per CU has nothing to overlap its
    # prologue / epilogue / DMA waits with (transformer_big's 4096 x 1024 weight gradients: 642 -> 363 us)
    return max(1, min(target_blocks // tiles, K // 512))


def residual_ln_fwd(x: Tensor, residual: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor],
                    eps: float, drop_p: float = 0.0, seed: int = 0, offset: int = 0, want_z: bool = True,
                    rms: bool = False):
    """z = residual + dropout(x); y = LN(z), or RMSNorm(z) with `rms` (then mean is None and beta must be None).
    Returns (y or None, z or None, mean, rstd)."""
    require_gpu(x, residual, gamma, beta)
    d = x.size(-1)
    rows = x.numel() // d
    assert x.is_contiguous() and (residual is None or (residual.is_contiguous() and residual.shape == x.shape))
    _same(x, residual, gamma, beta, what='residual_ln_fwd')
    assert all(t is None or (t.numel() == d and t.is_contiguous()) for t in (gamma, beta))
    y = torch.empty_like(x) if gamma is not None else None
    z = torch.empty_like(x) if (want_z or gamma is None) else None
    mean = rstd = None
    if gamma is not None:
        assert not (rms and beta is not None)
        mean = None if rms else torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L = lib.load()
    check(L.pk_residual_ln_fwd(ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(z), ptr(y), ptr(mean), ptr(rstd),
                               rows, d, float(eps), float(drop_p), int(seed), int(offset), dtype_code(x),
                               stream_ptr()), 'pk_residual_ln_fwd')
    return y, z, mean, rstd


def gemm_ln_eligible(x: Tensor, weight: Tensor) -> bool:
    """can `LN(residual + dropout(x Wᵀ + b))` run as ONE kernel (pk_gemm_ln_fwd)?  x [M, K], weight [512, K], 16-bit"""
    if not (x.is_cuda and weight.is_cuda) or x.dtype != weight.dtype or x.dtype not in (torch.bfloat16, torch.float16):
        return False
    if x.dim() != 2 or weight.dim() != 2 or x.size(1) != weight.size(1) or x.stride(1) != 1 or weight.stride(1) != 1:
        return False
    if x.data_ptr() % 16 or weight.data_ptr() % 16 or x.size(0) == 0:
        return False
    return bool(lib.load().pk_gemm_ln_eligible(x.size(0), weight.size(0), x.size(1), _ld(x), _ld(weight), dtype_code(x)))


def gemm_ln_eligible_shape(rows: int, weight: Tensor) -> bool:
    """the same question for a contiguous [rows, K] input that does not exist yet"""
    if not weight.is_cuda or weight.dtype not in (torch.bfloat16, torch.float16) or weight.dim() != 2:
        return False
    if weight.stride(1) != 1 or weight.data_ptr() % 16 or rows == 0:
        return False
    return bool(lib.load().pk_gemm_ln_eligible(rows, weight.size(0), weight.size(1), weight.size(1), _ld(weight),
                                               dtype_code(weight)))


def gemm_ln_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Optional[Tensor], gamma: Tensor,
                beta: Optional[Tensor], eps: float, drop_p: float = 0.0, seed: int = 0, offset: int = 0,
                want_z: bool = True, rms: bool = False):
    """z = residual + dropout(x Wᵀ + bias); y = LN(z) (RMSNorm with `rms`) in one kernel.  x [M, K], weight [N, K],
    residual [M, N].  Returns (y, z or None, mean or None, rstd) like residual_ln_fwd."""
    require_gpu(x, weight, bias, residual, gamma, beta)
    _same(x, weight, bias, residual, gamma, beta, what='gemm_ln_fwd')
    M, K = x.shape
    N = weight.size(0)
    assert weight.size(1) == K and (residual is None or (residual.shape == (M, N) and residual.stride(1) == 1))
    assert all(t is None or (t.numel() == N and t.is_contiguous()) for t in (bias, gamma, beta))
    assert not (rms and beta is not None)
    y = torch.empty(M, N, dtype=x.dtype, device=x.device)
    z = torch.empty(M, N, dtype=x.dtype, device=x.device) if want_z else None
    mean = None if rms else torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib.load().pk_gemm_ln_fwd(ptr(x), ptr(weight), ptr(bias), ptr(residual), ptr(gamma), ptr(beta), ptr(z), ptr(y),
                                    ptr(mean), ptr(rstd), M, N, K, _ld(x), _ld(weight),
                                    _ld(residual) if residual is not None else 0, float(eps), float(drop_p), int(seed),
                                    int(offset), dtype_code(x), stream_ptr()), 'pk_gemm_ln_fwd')
    return y, z, mean, rstd


def residual_ln_bwd(dy: Optional[Tensor], dz_extra: Optional[Tensor], z: Optional[Tensor], gamma: Optional[Tensor],
                    mean: Optional[Tensor], rstd: Optional[Tensor], *, want_dres: bool, want_dx: bool,
                    want_param_grads: bool, has_beta: bool = True, drop_p: float = 0.0, seed: int = 0,
                    offset: int = 0):
    """Returns (dres, dx, dgamma, dbeta); see include/pasero_hip.h:pk_residual_ln_bwd"""
    ref = dy if dy is not None else dz_extra
    require_gpu(dy, dz_extra, z, gamma)
    d = ref.size(-1)
    rows = ref.numel() // d
    for t in (dy, dz_extra, z):
        assert t is None or (t.is_contiguous() and t.numel() == ref.numel())
    _same(ref, dy, dz_extra, z, gamma, what='residual_ln_bwd')
    assert gamma is None or (gamma.numel() == d and gamma.is_contiguous())
    for t in (mean, rstd):
        assert t is None or (t.dtype == torch.float32 and t.numel() == rows and t.is_contiguous())
    dres = torch.empty_like(ref) if want_dres else None
    dx = torch.empty_like(ref) if want_dx else None
    dgamma = dbeta = None
    ws, ws_bytes = None, 0
    L = lib.load()
    if gamma is not None and want_param_grads:
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma) if has_beta else None
        ws_bytes = L.pk_residual_ln_bwd_workspace(rows, d)
        ws = lib.workspace(ws_bytes, ref.device, 'ln')
    check(L.pk_residual_ln_bwd(ptr(dy), ptr(dz_extra), ptr(z), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                               ptr(dgamma), ptr(dbeta), ptr(ws), ws_bytes, rows, d, float(drop_p), int(seed),
                               int(offset), dtype_code(ref), stream_ptr()), 'pk_residual_ln_bwd')
    return dres, dx, dgamma, dbeta


def _bs_rs(t: Tensor):
    """(batch stride, row stride) of a (B, L, H*64) view whose last dim is contiguous"""
    assert t.dim() == 3 and (t.stride(2) == 1 or t.size(2) == 1)
    return t.stride(0), t.stride(1)


def _rope_args(rope, T: int, S: int, hd: int):
    """rope = (cos_t, sin_t, q_pos0, k_pos0): fp32 [max_pos, hd / 2] tables and the positions of query 0 / key 0"""
    cos_t, sin_t, q0, k0 = rope
    require_gpu(cos_t, sin_t)
    assert cos_t.dtype == torch.float32 and sin_t.dtype == torch.float32 and cos_t.is_contiguous() and sin_t.is_contiguous()
    assert cos_t.shape == sin_t.shape and cos_t.size(1) * 2 == hd  # (the library checks the positions against the table)
    return ptr(cos_t), ptr(sin_t), cos_t.size(0), int(q0), int(k0)


def attn_fwd(q: Tensor, k: Tensor, v: Tensor, num_heads: int, key_pad: Optional[Tensor], causal: bool,
             scale: float, drop_p: float = 0.0, seed: int = 0, offset: int = 0, rope=None):
    """q (B,T,D), k/v (B,S,D) (views with arbitrary batch/row strides), D = num_heads * head_dim (64 or 128).
    Returns (o (B,T,D) contiguous, lse (B,H,T) fp32) — and, with drop_p > 0 (attention-probability dropout), the keep-bit
    mask (B,H,T,8*ceil(S/64)) uint8 that attn_bwd needs."""
    require_gpu(q, k, v, key_pad)
    B, T, D = q.shape
    S = k.size(1)
    hd = D // num_heads
    _same(q, k, v, what='attn_fwd')
    assert k.shape == (B, S, D) and v.shape == (B, S, D)
    o = torch.empty(B, T, D, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=q.device)
    mask = torch.empty(B, num_heads, T, 8 * ((S + 63) // 64), dtype=torch.uint8, device=q.device) if drop_p > 0 else None
    if key_pad is not None:
        assert key_pad.dtype == torch.bool and key_pad.shape == (B, S) and key_pad.is_contiguous()
    L = lib.load()
    if rope is not None:  # rotary positions inside the kernel: q, k are the unrotated projection
        check(L.pk_attn_fwd_rope(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(key_pad), B, num_heads, T, S, hd,
                                 *_bs_rs(q), *_bs_rs(k), *_bs_rs(v), *_bs_rs(o), int(causal), float(scale), float(drop_p),
                                 int(seed), int(offset), ptr(mask), *_rope_args(rope, T, S, hd), dtype_code(q),
                                 stream_ptr()), 'pk_attn_fwd_rope')
        return (o, lse, mask) if drop_p > 0 else (o, lse)
    check(L.pk_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(key_pad), B, num_heads, T, S, hd,
                        *_bs_rs(q), *_bs_rs(k), *_bs_rs(v), *_bs_rs(o), int(causal), float(scale), float(drop_p),
                        int(seed), int(offset), ptr(mask), dtype_code(q), stream_ptr()), 'pk_attn_fwd')
    return (o, lse, mask) if drop_p > 0 else (o, lse)


def attn_bwd(q, k, v, o, d_o, lse, num_heads: int, key_pad, causal: bool, scale: float,
             dq: Optional[Tensor] = None, dk: Optional[Tensor] = None, dv: Optional[Tensor] = None,
             drop_p: float = 0.0, drop_mask: Optional[Tensor] = None, rope=None):
    require_gpu(q, k, v, o, d_o, lse, key_pad, drop_mask)
    B, T, D = q.shape
    S = k.size(1)
    hd = D // num_heads
    if d_o.stride(2) != 1:
        d_o = d_o.contiguous()
    _same(q, k, v, o, d_o, dq, dk, dv, what='attn_bwd')
    assert k.shape == (B, S, D) and v.shape == (B, S, D) and o.shape == (B, T, D) and d_o.shape == (B, T, D)
    assert lse.dtype == torch.float32 and lse.shape == (B, num_heads, T) and lse.is_contiguous()
    if key_pad is not None:
        assert key_pad.dtype == torch.bool and key_pad.shape == (B, S) and key_pad.is_contiguous()
    if drop_p > 0:
        assert drop_mask is not None and drop_mask.dtype == torch.uint8 and drop_mask.is_contiguous() \
            and drop_mask.shape == (B, num_heads, T, 8 * ((S + 63) // 64))
    dq = torch.empty(B, T, D, dtype=q.dtype, device=q.device) if dq is None else dq
    dk = torch.empty(B, S, D, dtype=q.dtype, device=q.device) if dk is None else dk
    dv = torch.empty(B, S, D, dtype=q.dtype, device=q.device) if dv is None else dv
    delta = torch.empty(B, num_heads, T, dtype=torch.float32, device=q.device)
    L = lib.load()
    if rope is not None:  # dq, dk: gradients of the UNROTATED q, k
        check(L.pk_attn_bwd_rope(ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
                                 ptr(key_pad), B, num_heads, T, S, hd, *_bs_rs(q), *_bs_rs(k), *_bs_rs(v), *_bs_rs(o),
                                 *_bs_rs(d_o), *_bs_rs(dq), *_bs_rs(dk), *_bs_rs(dv), int(causal), float(scale),
                                 float(drop_p), ptr(drop_mask), *_rope_args(rope, T, S, hd), dtype_code(q), stream_ptr()),
              'pk_attn_bwd_rope')
        return dq, dk, dv
    check(L.pk_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
                        ptr(key_pad), B, num_heads, T, S, hd, *_bs_rs(q), *_bs_rs(k), *_bs_rs(v), *_bs_rs(o),
                        *_bs_rs(d_o), *_bs_rs(dq), *_bs_rs(dk), *_bs_rs(dv), int(causal), float(scale),
                        float(drop_p), ptr(drop_mask), dtype_code(q), stream_ptr()), 'pk_attn_bwd')
    return dq, dk, dv


def attn_probs(q: Tensor, k: Tensor, num_heads: int, key_pad: Optional[Tensor], causal: bool, scale: float) -> Tensor:
    """attention weights (B, T, H, S) of q (B,T,D) against k (B,S,D) — the `return_attn` output of the reference"""
    require_gpu(q, k, key_pad)
    B, T, D = q.shape
    S = k.size(1)
    _same(q, k, what='attn_probs')
    assert k.shape == (B, S, D)
    if key_pad is not None:
        assert key_pad.dtype == torch.bool and key_pad.shape == (B, S) and key_pad.is_contiguous()
    probs = torch.empty(B, T, num_heads, S, dtype=q.dtype, device=q.device)
    L = lib.load()
    check(L.pk_attn_probs(ptr(q), ptr(k), ptr(probs), ptr(key_pad), B, num_heads, T, S, D // num_heads, *_bs_rs(q),
                          *_bs_rs(k), int(causal), float(scale), dtype_code(q), stream_ptr()), 'pk_attn_probs')
    return probs


def embed_fwd(ids: Tensor, E: Tensor, pos: Optional[Tensor], scale: float, pos_start: int, drop_p: float = 0.0,
              seed: int = 0, offset: int = 0) -> Tensor:
    """ids (B,T) int64 -> (B,T,d); pos: (P,d) table in E's dtype or None; rows pos_start.. are added"""
    require_gpu(ids, E, pos)
    B, T = ids.shape
    V, d = E.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous() and E.is_contiguous()
    if pos is not None:
        assert pos.dtype == E.dtype and pos.is_contiguous() and pos.size(1) == d and pos_start + T <= pos.size(0), \
            f'input sequence is too long: {T}, positional embedding size: {pos.size(0)}'
    out = torch.empty(B, T, d, dtype=E.dtype, device=E.device)
    L = lib.load()
    check(L.pk_embed_fwd(ptr(ids), ptr(E), ptr(pos), ptr(out), B * T, T, d, V, float(scale), int(pos_start),
                         float(drop_p), int(seed), int(offset), dtype_code(E), stream_ptr()), 'pk_embed_fwd')
    return out


def add_positions(x: Tensor, pos: Optional[Tensor], scale: float, pos_start: int, drop_p: float = 0.0, seed: int = 0,
                  offset: int = 0) -> Tensor:
    """x (B,T,d) dense features -> dropout(x * scale + pos[pos_start : pos_start+T])"""
    require_gpu(x, pos)
    B, T, d = x.shape
    assert x.is_contiguous()
    if pos is not None:
        assert pos.dtype == x.dtype and pos.is_contiguous() and pos_start + T <= pos.size(0), \
            f'input sequence is too long: {T}, positional embedding size: {pos.size(0)}'
    out = torch.empty_like(x)
    L = lib.load()
    check(L.pk_embed_fwd(None, ptr(x), ptr(pos), ptr(out), B * T, T, d, B * T, float(scale), int(pos_start),
                         float(drop_p), int(seed), int(offset), dtype_code(x), stream_ptr()), 'pk_embed_fwd')
    return out


def embed_bwd(ids: Tensor, dout: Tensor, V: int, pad_idx: int, scale: float, drop_p: float = 0.0, seed: int = 0,
              offset: int = 0, into: Optional[Tensor] = None) -> Tensor:
    """dE (V, d) of the lookup; `into`: add the rows into that (V, d) gradient instead (pk_embed_bwd_acc: no zero fill, no
    dense pass — the tied projection's dW that the table already received) and return it"""
    require_gpu(ids, dout, into)
    d = dout.size(-1)
    assert dout.is_contiguous() and ids.is_contiguous() and ids.dtype == torch.int64
    assert dout.numel() == ids.numel() * d
    if into is not None:
        assert into.shape == (V, d) and into.dtype == dout.dtype and into.is_contiguous()
    dE = into if into is not None else torch.empty(V, d, dtype=dout.dtype, device=dout.device)
    ws_bytes = lib.load().pk_embed_bwd_workspace(ids.numel(), V, d)
    ws = lib.workspace(ws_bytes, dout.device, 'embed')
    L = lib.load()
    fn = L.pk_embed_bwd_acc if into is not None else L.pk_embed_bwd
    check(fn(ptr(ids), ptr(dout), ptr(dE), ptr(ws), ws_bytes, ids.numel(), d, V, int(pad_idx),
             float(scale), float(drop_p), int(seed), int(offset), dtype_code(dout), stream_ptr()),
          'pk_embed_bwd')
    return dE


def ce_rows(logits: Tensor, target: Tensor, pad_idx: int, eps: float, row_loss: Tensor, row_nll: Tensor,
            dlogits: Optional[Tensor] = None, row_lse: Optional[Tensor] = None):
    """logits (rows, V) [row stride allowed]; writes row_loss/row_nll (rows,) fp32 and optionally dlogits"""
    require_gpu(logits, target, dlogits)
    rows, V = logits.shape
    assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == rows
    _same(logits, dlogits, what='ce_rows')
    assert dlogits is None or dlogits.shape == logits.shape
    for t in (row_loss, row_nll, row_lse):
        assert t is None or (t.dtype == torch.float32 and t.numel() == rows and t.is_contiguous() and t.is_cuda)
    L = lib.load()
    check(L.pk_ce_rows(ptr(logits), _ld(logits), ptr(target), ptr(dlogits), _ld(dlogits) if dlogits is not None else 0,
                       ptr(row_loss), ptr(row_nll), ptr(row_lse), rows, V, int(pad_idx), float(eps or 0.0),
                       dtype_code(logits), stream_ptr()), 'pk_ce_rows')


def ce_finalize(row_loss: Tensor, row_nll: Tensor, target: Tensor, pad_idx: int) -> Tensor:
    n = target.numel()
    assert target.dtype == torch.int64 and target.is_contiguous()
    for t in (row_loss, row_nll):
        assert t.dtype == torch.float32 and t.numel() == n and t.is_contiguous()
    sums = torch.empty(3, dtype=torch.float32, device=row_loss.device)
    L = lib.load()
    check(L.pk_ce_finalize(ptr(row_loss), ptr(row_nll), ptr(target), target.numel(), int(pad_idx), ptr(sums),
                           stream_ptr()), 'pk_ce_finalize')
    return sums


def colsum(x: Tensor) -> Tensor:
    """x (M, N) -> (N,) column sums in x's dtype (fp32 accumulation)"""
    require_gpu(x)
    M, N = x.shape
    out = torch.empty(N, dtype=x.dtype, device=x.device)
    L = lib.load()
    ws_bytes = L.pk_colsum_workspace(M, N)
    ws = lib.workspace(ws_bytes, x.device, 'colsum')
    check(L.pk_colsum(ptr(x), _ld(x), ptr(out), M, N, ptr(ws), ws_bytes, dtype_code(x), stream_ptr()), 'pk_colsum')
    return out


def dropout(x: Tensor, p: float, seed: int, offset: int) -> Tensor:
    require_gpu(x)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    L = lib.load()
    check(L.pk_dropout(ptr(x), ptr(out), x.numel(), float(p), int(seed), int(offset), dtype_code(x), stream_ptr()),
          'pk_dropout')
    return out


def scale(x: Tensor, dev_scalar: Optional[Tensor], host_scalar: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    require_gpu(x, dev_scalar)
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    assert out.dtype == x.dtype and out.numel() == x.numel() and out.is_contiguous()
    if dev_scalar is not None:
        assert dev_scalar.dtype == torch.float32 and dev_scalar.numel() == 1
    L = lib.load()
    check(L.pk_scale(ptr(x), ptr(out), x.numel(), ptr(dev_scalar), float(host_scalar), dtype_code(x), stream_ptr()),
          'pk_scale')
    return out


def act_fwd(x: Tensor, act: str) -> Tensor:
    require_gpu(x)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    check(lib.load().pk_act_fwd(ptr(x), ptr(out), x.numel(), ACT[act], dtype_code(x), stream_ptr()), 'pk_act_fwd')
    return out


def act_bwd(dy: Tensor, x: Tensor, act: str) -> Tensor:
    require_gpu(dy, x)
    assert x.is_contiguous() and dy.is_contiguous() and dy.numel() == x.numel()
    _same(x, dy, what='act_bwd')
    out = torch.empty_like(x)
    check(lib.load().pk_act_bwd(ptr(dy), ptr(x), ptr(out), x.numel(), ACT[act], dtype_code(x), stream_ptr()),
          'pk_act_bwd')
    return out


def glu_fwd(x: Tensor) -> Tensor:
    require_gpu(x)
    assert x.is_contiguous() and x.size(-1) % 2 == 0
    C = x.size(-1) // 2
    out = torch.empty(*x.shape[:-1], C, dtype=x.dtype, device=x.device)
    check(lib.load().pk_glu_fwd(ptr(x), ptr(out), x.numel() // (2 * C), C, dtype_code(x), stream_ptr()), 'pk_glu_fwd')
    return out


def glu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    require_gpu(dy, x)
    assert x.is_contiguous() and dy.is_contiguous() and 2 * dy.numel() == x.numel()
    _same(x, dy, what='glu_bwd')
    C = x.size(-1) // 2
    dx = torch.empty_like(x)
    check(lib.load().pk_glu_bwd(ptr(dy), ptr(x), ptr(dx), x.numel() // (2 * C), C, dtype_code(x), stream_ptr()),
          'pk_glu_bwd')
    return dx


def col2im1d(dA: Tensor, B: int, L: int, C: int, R: int, Lout: int, ksize: int, stride: int, pad: int) -> Tensor:
    require_gpu(dA)
    assert dA.is_contiguous() and dA.shape == (B * R, ksize * C)
    dx = torch.empty(B, L, C, dtype=dA.dtype, device=dA.device)
    check(lib.load().pk_col2im1d(ptr(dA), ptr(dx), B, L, C, R, Lout, ksize, stride, pad, dtype_code(dA),
                                 stream_ptr()), 'pk_col2im1d')
    return dx


def log_mel(wav: Tensor, wav_len: Optional[Tensor] = None) -> Tensor:
    """wav (B, n<=480000) fp32 on the GPU (16 kHz) -> Whisper log-mel features (B, 3000, 80) fp32.
    `wav_len` (B,) int64: valid samples per clip (default: all n)."""
    require_gpu(wav, wav_len)
    assert wav.dim() == 2 and wav.dtype == torch.float32 and wav.stride(1) == 1
    B, n = wav.shape
    if wav_len is None:
        wav_len = torch.full((B,), min(n, 480000), dtype=torch.int64, device=wav.device)
    assert wav_len.dtype == torch.int64 and wav_len.is_contiguous() and wav_len.numel() == B
    out = torch.empty(B, 3000, 80, dtype=torch.float32, device=wav.device)
    L = lib.load()
    ws_bytes = L.pk_logmel_workspace(B)
    ws = lib.workspace(ws_bytes, wav.device, 'logmel')
    check(L.pk_logmel(ptr(wav), ptr(wav_len), wav.stride(0), ptr(out), ptr(ws), ws_bytes, B, stream_ptr()),
          'pk_logmel')
    return out


def rope(x: Tensor, cos_t: Tensor, sin_t: Tensor, ncols: int, pos_offset: int, inverse: bool = False) -> Tensor:
    """x (B, T, C) contiguous packed projection; rotates the first `ncols` columns (heads of 2 * cos_t.size(1) = 64 or
    128), copies the rest"""
    require_gpu(x, cos_t, sin_t)
    B, T, C = x.shape
    assert x.is_contiguous() and cos_t.dtype == torch.float32 and cos_t.is_contiguous() and sin_t.is_contiguous()
    assert cos_t.shape == sin_t.shape and cos_t.size(1) in (32, 64)
    y = torch.empty_like(x)
    check(lib.load().pk_rope(ptr(x), ptr(y), B * T, T, C, int(ncols), C, ptr(cos_t), ptr(sin_t), cos_t.size(0),
                             int(pos_offset), int(inverse), 2 * cos_t.size(1), dtype_code(x), stream_ptr()), 'pk_rope')
    return y


def gated_act_bwd(dh: Tensor, z: Tensor, u: Tensor, act: str):
    """h = act(z) * u  ->  (dz, du)"""
    require_gpu(dh, z, u)
    assert dh.is_contiguous() and z.is_contiguous() and u.is_contiguous() and dh.numel() == z.numel() == u.numel()
    _same(z, dh, u, what='gated_act_bwd')
    dz, du = torch.empty_like(z), torch.empty_like(u)
    check(lib.load().pk_gated_act_bwd(ptr(dh), ptr(z), ptr(u), ptr(dz), ptr(du), z.numel(), ACT[act], dtype_code(z),
                                      stream_ptr()), 'pk_gated_act_bwd')
    return dz, du


def argmax_rows(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """index of the first maximum of every row of x (rows, n) [row stride allowed] -> int64 (rows,), or written into
    `out`, a 1-D int64 view with any stride (e.g. a column of the (B, T) token buffer): the greedy choice of
    pasero/decoding.py:1196-1205 without an intermediate tensor"""
    require_gpu(x, out)
    rows, n = x.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=x.device)
    assert out.dtype == torch.int64 and out.dim() == 1 and out.numel() == rows
    L = lib.load()
    check(L.pk_argmax_rows(ptr(x), rows, n, _ld(x), ptr(out), out.stride(0) if rows > 1 else 1, dtype_code(x),
                           stream_ptr()), 'pk_argmax_rows')
    return out


def mt_copy_plan(numels, device):
    """chunk list (device tensors) of a multi-tensor copy over tensors of `numels` elements (pk_mt_copy), plus the
    pointer table's home: a device tensor and two pinned host buffers that alternate between calls (the upload is an
    asynchronous copy on the launch stream; no allocation and no host synchronisation per call)"""
    chunk = lib.load().pk_mt_chunk_size()
    ct, cs = [], []
    for i, n in enumerate(numels):
        for start in range(0, n, chunk):
            ct.append(i)
            cs.append(start)
    n = len(numels)
    table = torch.empty(3 * n, dtype=torch.int64, device=device)
    pinned = [torch.empty(3 * n, dtype=torch.int64, pin_memory=True) for _ in range(2)]
    for buf in pinned:
        buf[2 * n:] = torch.tensor(list(numels), dtype=torch.int64)
    return {'ct': torch.tensor(ct, dtype=torch.int32, device=device), 'cs': torch.tensor(cs, dtype=torch.int64, device=device),
            'numels': list(numels), 'table': table, 'pinned': pinned, 'turn': 0, 'uploaded': [None, None]}


def mt_copy(srcs, dsts, plan) -> None:
    """dsts[i] <- srcs[i] for every pair, in one launch (the gradient pack of a data-parallel bucket, ddp.py)"""
    if not srcs:
        return
    require_gpu(*srcs, *dsts)
    numels, n = plan['numels'], len(srcs)
    for s, d, k in zip(srcs, dsts, numels):
        assert s.dtype == dsts[0].dtype and d.dtype == dsts[0].dtype and s.numel() == k and d.numel() == k
        assert s.is_contiguous() and d.is_contiguous()
    plan['turn'] ^= 1
    turn = plan['turn']
    host = plan['pinned'][turn]
    if plan['uploaded'][turn] is not None:
        # the upload issued from this pinned buffer two calls ago must have been READ by the device before the host
        # rewrites it (a host that runs steps ahead of the GPU — lazy logs, a bench loop — would otherwise hand the
        # kernel another call's addresses); by now it almost always has: the wait is a query
        plan['uploaded'][turn].synchronize()
    host.numpy()[:2 * n] = [s.data_ptr() for s in srcs] + [d.data_ptr() for d in dsts]
    plan['table'].copy_(host, non_blocking=True)
    ev = plan['uploaded'][turn] or torch.cuda.Event()
    ev.record()
    plan['uploaded'][turn] = ev
    check(lib.load().pk_mt_copy(ptr(plan['table']), n, ptr(plan['ct']), ptr(plan['cs']), plan['ct'].numel(),
                                dtype_code(dsts[0]), stream_ptr()), 'pk_mt_copy')
