#!/usr/bin/env python3
"""A/B of the B-stationary kernel (gemmbs.hip) against the tiled kernels on the K = 512 shapes of the training step, in
alternating rounds inside ONE process (pk_gemm_use_bs toggles the dispatch), next to torch.matmul (hipBLASLt) on the same
data; every output of the new kernel is compared BITWISE with the tiled kernel's (same MFMA shape and k order, same
epilogue arithmetic) and, on a row sample, with an fp32 product.
usage: tools/gemmbs_bench.py [--only NAME ...] [--iters N] [--rounds R] [--no-torch] [--cold]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402
from gemm_bench import bench, bench_cold  # noqa: E402

SHAPES = {
    # name: (M, N, K, b_col, bias, act)
    'qkv_fwd': (32768, 1536, 512, False, True, 'none'),
    'crossq_fwd': (32768, 512, 512, False, True, 'none'),
    'crosskv_fwd': (32768, 1024, 512, False, True, 'none'),
    'fc1_fwd': (32768, 2048, 512, False, True, 'relu'),
    'out_dx': (32768, 512, 512, True, False, 'none'),
    'fc2_dx_nomask': (32768, 2048, 512, True, False, 'none'),
    'fc2_dx': (32768, 2048, 512, True, False, 'mask'),
    'whisper_fc1': (24000, 2048, 512, False, True, 'relu'),
    'ragged': (30000 + 8, 1032, 512, False, True, 'none'),
}


def main():
    from pasero_amd import lib
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--no-torch', action='store_true')
    ap.add_argument('--cold', action='store_true')
    args = ap.parse_args()
    L = lib.load()
    timer = bench_cold if args.cold else bench
    for name, (M, N, K, b_col, has_bias, act) in SHAPES.items():
        if args.only and name not in args.only:
            continue
        A = torch.randn(M, K, device='cuda').bfloat16()
        B = torch.randn(N, K, device='cuda').bfloat16()
        bias = torch.randn(N, device='cuda').bfloat16() if has_bias else None
        b = B.t().contiguous() if b_col else B
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        aux = torch.randn(M, N, device='cuda').bfloat16() if act == 'mask' else None
        if act == 'mask':
            run = lambda: F.gemm(A, b, b_col=b_col, aux=aux, act='relu', mode=2, out=out)  # noqa: E731
        else:
            run = lambda: F.gemm(A, b, b_col=b_col, bias=bias, act=act, out=out)  # noqa: E731
        res = {}
        for mode in (1, 0):
            L.pk_gemm_use_bs(mode)
            out.fill_(float('nan'))
            run()
            res[mode] = out.clone()
        same = torch.equal(res[1].view(torch.int16), res[0].view(torch.int16))
        rows = torch.arange(0, M, max(1, M // 257), device='cuda')
        ref = A[rows].float() @ B.float().t()
        if bias is not None:
            ref = ref + bias.float()
        if act == 'relu':
            ref = ref.relu()
        if act == 'mask':
            ref = ref * (aux[rows].float() > 0)
        err = ((res[1][rows].float() - ref).abs().max() / ref.abs().max()).item()
        times = {1: [], 0: []}
        for _ in range(args.rounds):
            for mode in (1, 0):
                L.pk_gemm_use_bs(mode)
                times[mode].append(timer(run, args.iters))
        L.pk_gemm_use_bs(1)
        med = {m: sorted(v)[len(v) // 2] for m, v in times.items()}
        tf = {m: 2.0 * M * N * K / med[m] / 1e6 for m in med}
        line = (f'{name:14s} M={M:6d} N={N:5d} K={K:4d} row,{"col" if b_col else "row"} {act:4s}  bs {med[1]:7.1f} us '
                f'{tf[1]:7.1f} TF | tiled {med[0]:7.1f} us {tf[0]:7.1f} TF | bitwise equal: {same}  err vs fp32 {err:.1e}')
        if not args.no_torch:
            tref = (lambda: torch.matmul(A, b)) if b_col else (lambda: torch.matmul(A, b.t()))
            us2 = timer(tref, args.iters)
            line += f' | torch {us2:7.1f} us {2.0 * M * N * K / us2 / 1e6:7.1f} TF'
        print(line, flush=True)


if __name__ == '__main__':
    main()
