#!/usr/bin/env python3
"""Micro-benchmark of the residual + LayerNorm kernels over the model widths of the presets (bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    p = float(os.environ.get('LN_BENCH_P', '0'))
    for d in (512, 1024, 1280, 2048, 4096):
        rows = (32768 * 512) // d
        x = torch.randn(rows, d, device='cuda').bfloat16()
        res = torch.randn(rows, d, device='cuda').bfloat16()
        g = torch.ones(d, device='cuda').bfloat16()
        b = torch.zeros(d, device='cuda').bfloat16()
        dy = torch.randn(rows, d, device='cuda').bfloat16()
        y, z, mean, rstd = F.residual_ln_fwd(x, res, g, b, 1e-5)
        nbytes = rows * d * 2
        t_f = bench(lambda: F.residual_ln_fwd(x, res, g, b, 1e-5, p, 5, 7))       # reads x, res; writes z, y
        t_b = bench(lambda: F.residual_ln_bwd(dy, None, z, g, mean, rstd, want_dres=True,
                                              want_param_grads=True, want_dx=p > 0, drop_p=p, seed=5, offset=7) if p > 0 else
                    F.residual_ln_bwd(dy, None, z, g, mean, rstd, want_dres=True, want_dx=False, want_param_grads=True))
        print(f'd={d:5d} rows={rows:6d}  fwd {t_f:6.1f} us ({4 * nbytes / t_f / 1e6:5.2f} TB/s)   '
              f'bwd {t_b:6.1f} us ({3 * nbytes / t_b / 1e6:5.2f} TB/s)', flush=True)


if __name__ == '__main__':
    main()
