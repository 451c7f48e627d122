"""Thin, non-differentiable wrappers over the C ABI (one Python function per kernel entry point).
They only do argument checking, output allocation and the ctypes call; autograd lives in `pasero_amd.autograd`.
"""
from typing import Optional

import torch
from torch import Tensor

from . import lib
from .lib import ACT, check, dtype_code, ptr, require_gpu, stream_ptr


def _ld(t: Tensor) -> int:
    assert t.dim() == 2 and (t.stride(1) == 1 or t.size(1) == 1), 'operand must be 2-D with unit inner stride'
    return t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0))


def gemm(a: Tensor, b: Tensor, *, a_col: bool = False, b_col: bool = False, bias: Optional[Tensor] = None,
         aux: Optional[Tensor] = None, act: str = 'none', mode: int = 0, alpha: float = 1.0,
         out: Optional[Tensor] = None, preact: Optional[Tensor] = None, splitk: int = 1) -> Tensor:
    """C[m,n] = epi(alpha * sum_k A(m,k) B(n,k)).  `a` is [M,K] (or [K,M] if a_col), `b` is [N,K] (or [K,N] if b_col).
    See include/pasero_hip.h:pk_gemm for the epilogue modes."""
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
    ws, ws_bytes = None, 0
    if splitk > 1:
        ws_bytes = splitk * M * N * 4
        ws = lib.workspace(ws_bytes, a.device, 'splitk')
    L = lib.load()
    check(L.pk_gemm(ptr(a), ptr(b), ptr(out), ptr(bias), ptr(aux), ptr(preact), M, N, K, _ld(a), _ld(b), _ld(out),
                    _ld(aux) if aux is not None else 0, _ld(preact) if preact is not None else 0,
                    int(a_col), int(b_col), ACT[act], mode, float(alpha), dtype_code(a), int(splitk),
                    ptr(ws), ws_bytes, stream_ptr()), 'pk_gemm')
    return out


def choose_splitk(M: int, N: int, K: int, target_blocks: int = 512) -> int:
    """Split the contraction when the output has too few 128x128 tiles to fill 256 CUs (weight-gradient GEMMs)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 256 or K < 1024:
        return 1
    s = max(1, min(target_blocks // tiles, K // 512))
    return s


def residual_ln_fwd(x: Tensor, residual: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor],
                    eps: float, drop_p: float = 0.0, seed: int = 0, offset: int = 0, want_z: bool = True):
    """z = residual + dropout(x); y = LN(z).  Returns (y or None, z or None, mean, rstd)."""
    require_gpu(x, residual, gamma, beta)
    d = x.size(-1)
    rows = x.numel() // d
    assert x.is_contiguous() and (residual is None or (residual.is_contiguous() and residual.shape == x.shape))
    y = torch.empty_like(x) if gamma is not None else None
    z = torch.empty_like(x) if (want_z or gamma is None) else None
    mean = rstd = None
    if gamma is not None:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L = lib.load()
    check(L.pk_residual_ln_fwd(ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(z), ptr(y), ptr(mean), ptr(rstd),
                               rows, d, float(eps), float(drop_p), int(seed), int(offset), dtype_code(x),
                               stream_ptr()), 'pk_residual_ln_fwd')
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
        assert t is None or t.is_contiguous()
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
