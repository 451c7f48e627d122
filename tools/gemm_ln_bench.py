#!/usr/bin/env python3
"""The fused block end (pk_gemm_ln_fwd) against pk_gemm + pk_residual_ln_fwd, same box, same process, C2 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    for K in (512, 2048):
        for p in (0.0, 0.1):
            x = torch.randn(M, K, device='cuda').bfloat16()
            w = (torch.randn(512, K, device='cuda') / K ** 0.5).bfloat16()
            b = torch.randn(512, device='cuda').bfloat16()
            res = torch.randn(M, 512, device='cuda').bfloat16()
            g, bt = torch.ones(512, device='cuda').bfloat16(), torch.zeros(512, device='cuda').bfloat16()
            t_g = timeit(lambda: F.gemm(x, w, bias=b))
            v = F.gemm(x, w, bias=b)
            t_l = timeit(lambda: F.residual_ln_fwd(v, res, g, bt, 1e-5, p, 1, 2))
            t_f = timeit(lambda: F.gemm_ln_fwd(x, w, b, res, g, bt, 1e-5, p, 1, 2))
            fl = 2.0 * M * 512 * K
            print(f'M={M} K={K} p={p}: gemm {t_g:6.1f} us ({fl / t_g / 1e6:5.0f} TF) + LN {t_l:6.1f} us = {t_g + t_l:6.1f} us   '
                  f'fused {t_f:6.1f} us ({fl / t_f / 1e6:5.0f} TF)', flush=True)


if __name__ == '__main__':
    main()
