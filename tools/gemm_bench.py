#!/usr/bin/env python3
"""Micro-benchmark of pk_gemm on the hot-path shapes (random bf16 data): the phase-interleaved kernel (gemm8p.hip) and
the one-barrier-per-K-tile kernel (gemm256.hip) in alternating rounds inside ONE process, next to torch.matmul
(hipBLASLt) on the same data as the known-good reference; every result is also checked against an fp32 product.
usage: tools/gemm_bench.py [--only NAME ...] [--iters N] [--rounds R] [--no-torch] [--cold]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402

SHAPES = {
    # name: (M, N, K, a_col, b_col, splitk)
    'qkv_fwd': (32768, 1536, 512, False, False, 1),
    'out_fwd': (32768, 512, 512, False, False, 1),
    'fc1_fwd': (32768, 2048, 512, False, False, 1),
    'fc2_fwd': (32768, 512, 2048, False, False, 1),
    'vocab_fwd': (8192, 8032, 512, False, False, 1),
    'fc1_dx': (32768, 512, 2048, False, True, 1),
    'fc2_dx': (32768, 2048, 512, False, True, 1),
    'qkv_dx': (32768, 512, 1536, False, True, 1),
    'out_dx': (32768, 512, 512, False, True, 1),
    'fc1_dw': (2048, 512, 32768, True, True, 0),
    'out_dw': (512, 512, 32768, True, True, 0),
    'out_dw_sk8': (512, 512, 32768, True, True, 8),
    'out_dw_sk16': (512, 512, 32768, True, True, 16),
    'out_dw_sk32': (512, 512, 32768, True, True, 32),
    'out_dw_sk64': (512, 512, 32768, True, True, 64),
    'qkv_dw': (1536, 512, 32768, True, True, 0),
    'fc2_dw': (512, 2048, 32768, True, True, 0),
    'big_4k': (4096, 4096, 4096, False, False, 1),
    'big_qkv_fwd': (32768, 3072, 1024, False, False, 1),
    'big_out_fwd': (32768, 1024, 1024, False, False, 1),
    'big_fc1_fwd': (32768, 4096, 1024, False, False, 1),
    'big_fc2_fwd': (32768, 1024, 4096, False, False, 1),
    'big_fc1_dx': (32768, 1024, 4096, False, True, 1),
    'big_fc2_dx': (32768, 4096, 1024, False, True, 1),
    # transformer_big (d = 1024, f = 4096) weight gradients
    'big_fc1_dw_sk1': (4096, 1024, 32768, True, True, 1),
    'big_fc1_dw_sk2': (4096, 1024, 32768, True, True, 2),
    'big_fc1_dw_sk4': (4096, 1024, 32768, True, True, 4),
    'big_qkv_dw_sk2': (3072, 1024, 32768, True, True, 2),
    'big_qkv_dw_sk4': (3072, 1024, 32768, True, True, 4),
    'big_out_dw_sk8': (1024, 1024, 32768, True, True, 8),
    'big_out_dw_sk16': (1024, 1024, 32768, True, True, 16),
    # NLLB-1.3B shapes at C5 (d = 1024, f = 8192, 8192 rows per step)
    'nllb_fc1_dw_sk1': (8192, 1024, 8192, True, True, 1),
    'nllb_fc1_dw_sk2': (8192, 1024, 8192, True, True, 2),
    'nllb_fc2_dw_sk1': (1024, 8192, 8192, True, True, 1),
    'nllb_fc2_dw_sk2': (1024, 8192, 8192, True, True, 2),
    'nllb_fc1_dx': (8192, 1024, 8192, False, True, 1),
    'nllb_fc1_dx_sk2': (8192, 1024, 8192, False, True, 2),
    'nllb_fc2_fwd': (8192, 1024, 8192, False, False, 1),
    'nllb_fc2_fwd_sk2': (8192, 1024, 8192, False, False, 2),
}


_flush = None


def bench_cold(fn, iters):
    """like bench(), but a 1 GiB write between launches evicts L2 / Infinity Cache (the in-model situation)"""
    global _flush
    if _flush is None:
        _flush = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    tot = 0.0
    for i in range(iters + 2):
        _flush.fill_(i & 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += s.elapsed_time(e)
    return tot / iters * 1e3


def bench(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    from pasero_amd import lib
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3, help='interleaved rounds per variant (median reported)')
    ap.add_argument('--no-torch', action='store_true')
    ap.add_argument('--cold', action='store_true', help='evict caches between launches')
    args = ap.parse_args()
    L = lib.load()
    for name, (M, N, K, a_col, b_col, splitk) in SHAPES.items():
        if args.only and name not in args.only:
            continue
        A = torch.randn(M, K, device='cuda').bfloat16()
        B = torch.randn(N, K, device='cuda').bfloat16()
        a = A.t().contiguous() if a_col else A
        b = B.t().contiguous() if b_col else B
        sk = F.choose_splitk(M, N, K) if splitk == 0 else splitk
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        timer = bench_cold if args.cold else bench
        run = lambda: F.gemm(a, b, a_col=a_col, b_col=b_col, splitk=sk, out=out)  # noqa: E731
        # correctness of both kernels on a row / column sample against fp32
        rows = torch.arange(0, M, max(1, M // 257), device='cuda')
        ref = A[rows].float() @ B.float().t()
        errs = {}
        for mode in (1, 0):
            L.pk_gemm_use_8p(mode)
            out.zero_()
            run()
            errs[mode] = ((out[rows].float() - ref).abs().max() / ref.abs().max()).item()
        times = {1: [], 0: []}
        for _ in range(args.rounds):
            for mode in (1, 0):
                L.pk_gemm_use_8p(mode)
                times[mode].append(timer(run, args.iters))
        L.pk_gemm_use_8p(1)
        med = {m: sorted(v)[len(v) // 2] for m, v in times.items()}
        tf = {m: 2.0 * M * N * K / med[m] / 1e6 for m in med}
        line = (f'{name:14s} M={M:6d} N={N:5d} K={K:6d} {"col" if a_col else "row"},{"col" if b_col else "row"} sk={sk:2d}  '
                f'8p {med[1]:7.1f} us {tf[1]:7.1f} TF (err {errs[1]:.1e}) | 256 {med[0]:7.1f} us {tf[0]:7.1f} TF (err {errs[0]:.1e})')
        if not args.no_torch:
            if a_col and b_col:
                tref = lambda: torch.matmul(a.t(), b)  # noqa: E731
            elif b_col:
                tref = lambda: torch.matmul(a, b)  # noqa: E731
            else:
                tref = lambda: torch.matmul(a, b.t())  # noqa: E731
            us2 = timer(tref, args.iters)
            line += f' | torch {us2:7.1f} us {2.0 * M * N * K / us2 / 1e6:7.1f} TF'
        print(line, flush=True)


if __name__ == '__main__':
    main()
