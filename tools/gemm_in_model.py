#!/usr/bin/env python3
"""GEMM census of a training step: HIP events around every pk_gemm main kernel for a few steps (include/pasero_hip.h:
pk_gemm_timing_*), grouped by kernel / operand layout / split factor / (M, N, K).  `--workload` takes bench.py's workloads
(c2_base_bf16, c3_big, c4_whisper, c5_nllb_1b3, c4_iwslt: the same model, batch and step as the bench line); `--preset` a
text configuration at a batch of its own.  Compare with tools/gemm_bench.py (the same shapes in isolation, operands warm) to
see which launches lose time to their context.

Kernel tags: 128 = gemm_kernel (128 x 128 tiles), 64 = the few-rows kernel, 256 = gemm256, 8 | flags = gemm8p (0x10 general
epilogue, 0x20 partial last K-tile, 0x400 the 128 x 256 tile, 0x40 the grouped weight gradients, 0x80 GEMM + LayerNorm,
0x800 gemmpw.hip's persistent 128 x 256 kernel, 0x4000 the persistent walk of 256 x 256 tiles), 0x200 | ... = the B-stationary kernel."""
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
    ap.add_argument('--workload', default=None, help="one of bench.py's workloads (overrides --preset)")
    ap.add_argument('--preset', default='TransformerConfig')
    ap.add_argument('--vocab', type=int, default=8032)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--len', type=int, default=128)
    ap.add_argument('--min-share', type=float, default=0.0, help='print only rows of at least this share of the GEMM time')
    args = ap.parse_args()
    import paramgen
    from pasero_amd import config as C, lib, rng
    from pasero_amd.transformer import Transformer
    wav = None
    if args.workload:
        import bench
        cfg, model, batch, wav = bench.build_workload(args.workload, torch.bfloat16, torch.device('cuda:0'))
    else:
        cfg = getattr(C, args.preset)()
        torch.manual_seed(0)
        model = Transformer(cfg, C.DistributedConfig(), C.SyntheticTask(args.vocab)).to(torch.bfloat16).cuda().train()
        rng.manual_seed(1)
        batch = {k: torch.from_numpy(v).cuda() for k, v in
                 paramgen.make_text_batch(1, args.batch, args.len, args.len, args.vocab, ragged=False).items()}

    def step():
        for p in model.parameters():
            p.grad = None
        if wav is not None:
            from pasero_amd import functional as PF
            batch['encoder_input'] = PF.log_mel(wav).to(torch.bfloat16)
        loss, _ = model(**batch)
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(40000, 1), 'pk_gemm_timing_start')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.steps):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    n = L.pk_gemm_timing_stop()
    ints = [ctypes.c_int() for _ in range(5)]
    mnk = [ctypes.c_longlong() for _ in range(3)]
    flops, ms = ctypes.c_double(), ctypes.c_float()
    agg = collections.OrderedDict()
    for i in range(n):
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(flops), ctypes.byref(ms)),
                  'pk_gemm_timing_read')
        lib.check(L.pk_gemm_timing_shape(i, *[ctypes.byref(x) for x in mnk]), 'pk_gemm_timing_shape')
        key = tuple(x.value for x in ints[:4]) + tuple(x.value for x in mnk) + (int(flops.value),)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms.value
    step_ms = ev[0].elapsed_time(ev[1]) / args.steps
    print(f'{n} launches over {args.steps} steps ({step_ms:.2f} ms per step with the event pairs in it); per step:')
    print('kernel a_col b_col splitk        M        N        K   2MNK (GF)  launches/step   avg us    TFLOP/s   ms/step  share')
    total = sum(t for _, t in agg.values()) / args.steps
    by_kernel = collections.defaultdict(float)
    for key, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        k, ac, bc, sk, M, N, K, fl = key
        by_kernel[k] += t / args.steps
        if t / args.steps < args.min_share * total:
            continue
        print(f'{k:#6x} {ac:5d} {bc:5d} {sk:6d} {M:8d} {N:8d} {K:8d} {fl / 1e9:11.2f} {cnt / args.steps:14.1f} '
              f'{1e3 * t / cnt:8.1f} {fl / (t / cnt * 1e-3) / 1e12:10.1f} {t / args.steps:9.3f} {100 * t / args.steps / total:5.1f}%')
    print(f'GEMM main kernels: {total:.2f} ms/step; by kernel tag: '
          + ', '.join(f'{k:#x}: {v:.2f}' for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])))


if __name__ == '__main__':
    main()
