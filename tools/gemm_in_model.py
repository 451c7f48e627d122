#!/usr/bin/env python3
"""Per-shape GEMM timings INSIDE the training step (C2 by default): HIP events around every pk_gemm main kernel for a few
steps (include/pasero_hip.h: pk_gemm_timing_*), grouped by kernel / operand layout / split factor / 2MNK.  Compare with
tools/gemm_bench.py (the same shapes in isolation, operands warm) to see which launches lose time to their context."""
import argparse
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--preset', default='TransformerConfig')
    ap.add_argument('--vocab', type=int, default=8032)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--len', type=int, default=128)
    args = ap.parse_args()
    import paramgen
    from pasero_amd import config as C, lib, rng
    from pasero_amd.transformer import Transformer
    cfg = getattr(C, args.preset)()
    torch.manual_seed(0)
    model = Transformer(cfg, C.DistributedConfig(), C.SyntheticTask(args.vocab)).to(torch.bfloat16).cuda().train()
    rng.manual_seed(1)
    batch = {k: torch.from_numpy(v).cuda() for k, v in
             paramgen.make_text_batch(1, args.batch, args.len, args.len, args.vocab, ragged=False).items()}

    def step():
        for p in model.parameters():
            p.grad = None
        loss, _ = model(**batch)
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(20000, 1), 'pk_gemm_timing_start')
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    n = L.pk_gemm_timing_stop()
    ints = [ctypes.c_int() for _ in range(5)]
    flops, ms = ctypes.c_double(), ctypes.c_float()
    agg = collections.OrderedDict()
    for i in range(n):
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(flops), ctypes.byref(ms)),
                  'pk_gemm_timing_read')
        key = tuple(x.value for x in ints[:4]) + (int(flops.value),)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms.value
    print(f'{n} launches over {args.steps} steps; per step:')
    print('kernel a_col b_col splitk      2MNK (GF)  launches/step   avg us    TFLOP/s   ms/step')
    total = 0.0
    for key, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        k, ac, bc, sk, fl = key
        total += t / args.steps
        print(f'{k:6d} {ac:5d} {bc:5d} {sk:6d} {fl / 1e9:14.2f} {cnt / args.steps:14.1f} {1e3 * t / cnt:8.1f} '
              f'{fl / (t / cnt * 1e-3) / 1e12:10.1f} {t / args.steps:9.3f}')
    print(f'GEMM main kernels: {total:.2f} ms/step')


if __name__ == '__main__':
    main()
