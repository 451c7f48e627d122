#!/usr/bin/env python3
"""The weight gradients of one layer: the grouped launch (pk_gemm_wgrad_group) against the one-by-one pk_gemm calls it
replaces, same box, same process.  Usage: python tools/wgrad_group_bench.py [d f rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
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
    d, f, rows = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 32768)
    def prob(n_out, k_in):
        return (torch.randn(rows, n_out, device='cuda').bfloat16(), torch.randn(rows, k_in, device='cuda').bfloat16(), True)
    layers = {'encoder layer (qkv, out, fc1, fc2)': [prob(3 * d, d), prob(d, d), prob(f, d), prob(d, f)],
              'decoder layer (+ cross q, kv, out)': [prob(3 * d, d), prob(d, d), prob(d, d), prob(2 * d, d), prob(d, d),
                                                     prob(f, d), prob(d, f)]}
    for name, entries in layers.items():
        flops = sum(2.0 * dy.size(0) * dy.size(1) * x.size(1) for dy, x, _ in entries)
        def one_by_one():
            for dy, x, _ in entries:
                db = torch.empty(dy.size(1), dtype=dy.dtype, device='cuda')
                F.gemm(dy, x, a_col=True, b_col=True, splitk=F.choose_splitk(dy.size(1), x.size(1), rows), asum_out=db)
        t1 = timeit(one_by_one)
        t2 = timeit(lambda: F.wgrad_group(entries))
        print(f'd={d} f={f} rows={rows} {name}: one by one {t1:7.1f} us ({flops / t1 / 1e6:6.0f} TFLOP/s)   grouped '
              f'{t2:7.1f} us ({flops / t2 / 1e6:6.0f} TFLOP/s)', flush=True)


if __name__ == '__main__':
    main()
