#!/usr/bin/env python3
"""End-to-end soak: N optimizer steps of the C2 workload (bf16, dropout 0.1, fused clip + Adam) on one fixed synthetic
batch — the loss per target token must fall monotonically-ish towards 0 (memorisation), stay finite, and the allocator's
peak must stop growing after the first steps.  usage: tools/train_soak.py [--steps 100]"""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--lr', type=float, default=5e-4)
    args = ap.parse_args()
    import bench
    from pasero_amd import config as C, rng
    from pasero_amd.optim import Adam
    from pasero_amd.transformer import Transformer
    V = 8032
    torch.manual_seed(0)
    model = Transformer(C.TransformerConfig(), C.DistributedConfig(), C.SyntheticTask(V)).bfloat16().cuda().train()
    rng.manual_seed(1)
    opt = Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.98), eps=1e-8)
    batch = bench.synthetic_batch(args.batch, 128, 128, V, seed=1, device=torch.device('cuda'))
    peaks, t0 = [], time.perf_counter()
    for step in range(1, args.steps + 1):
        model.zero_grad(set_to_none=True)
        loss, logs = model(**batch)
        loss.backward()
        lr = args.lr * min(1.0, step / 20)  # short warm-up
        for g in opt.param_groups:
            g['lr'] = lr
        gnorm = opt.fused_step(scale=1.0 / logs['num_tokens'], max_norm=1.0)
        peaks.append(torch.cuda.max_memory_allocated() >> 20)
        if step % 10 == 0 or step == 1:
            per_tok = loss.item() / logs['num_tokens']
            assert math.isfinite(per_tok) and math.isfinite(gnorm.item())
            print(f'step {step:4d}  loss/token {per_tok:7.4f} (ln V = {math.log(V):.3f})  gnorm {gnorm.item():8.3f}  '
                  f'peak {peaks[-1]} MiB  {1e3 * (time.perf_counter() - t0) / step:6.1f} ms/step', flush=True)
    assert peaks[-1] == peaks[min(10, len(peaks) - 1)], 'allocator peak keeps growing'
    print('ok')


if __name__ == '__main__':
    main()
