#!/usr/bin/env python3
"""Cross-entropy rows micro-benchmark (pk_ce_rows with the gradient written in place of a copy): us per launch and TB/s of
its algorithmic bytes (logits read once by the register-resident kernels, twice by the two-pass one, gradient written once).

    python tools/ce_bench.py [--rows 8192] [--iters 50]      (PK_CE_NO_REG=1: the two-pass kernel everywhere)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=8192)
    ap.add_argument('--iters', type=int, default=50)
    args = ap.parse_args()
    for V in (8032, 51865, 70376, 98304, 256206):
        rows = args.rows if V < 200000 else args.rows // 4
        g = torch.Generator(device='cuda').manual_seed(V)
        vpad = (V + 7) // 8 * 8  # (the models' logits live in rows of a pitch that is a multiple of 8: pk_gemm_ex, PK_GEMM_PAD_K)
        logits = (3 * torch.randn(rows, vpad, device='cuda', generator=g)).bfloat16()[:, :V]
        target = torch.randint(0, V, (rows,), device='cuda', generator=g)
        dl = torch.empty(rows, vpad, device='cuda', dtype=torch.bfloat16)[:, :V]
        rl = torch.empty(rows, device='cuda')
        rn = torch.empty(rows, device='cuda')
        for what, out in (('copy', dl), ('in place', logits)):  # (the models' vocabulary loss writes the gradient over the logits)
            fn = lambda: F.ce_rows(logits, target, 1, 0.1, rl, rn, out)
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / args.iters
            print(f'rows={rows} V={V} {what:8s}: {us:8.1f} us  ({2 * rows * V * 2 / us / 1e6:.2f} TB/s of read + write)')


if __name__ == '__main__':
    main()
