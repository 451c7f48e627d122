#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the hot-path shapes (bf16, packed qkv like the model)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402


def bench(fn, iters=20):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    for name, B, H, T, S, causal in [('enc_self', 256, 8, 128, 128, False), ('dec_self', 256, 8, 128, 128, True),
                                     ('cross', 256, 8, 128, 128, False), ('whisper_enc', 16, 8, 1500, 1500, False)]:
        D = H * 64
        qkv = torch.randn(B, T, 3 * D, device='cuda').bfloat16()
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        if name == 'cross':
            q = torch.randn(B, T, D, device='cuda').bfloat16()
            kv = torch.randn(B, S, 2 * D, device='cuda').bfloat16()
            k, v = kv[..., :D], kv[..., D:]
        pad = None
        if not causal:
            lens = torch.full((B,), S, device='cuda')
            pad = torch.arange(S, device='cuda')[None] >= lens[:, None]
        o, lse = F.attn_fwd(q, k, v, H, pad, causal, 0.125)
        do = torch.randn_like(o)
        fwd = bench(lambda: F.attn_fwd(q, k, v, H, pad, causal, 0.125), args.iters)
        bwd = bench(lambda: F.attn_bwd(q, k, v, o, do, lse, H, pad, causal, 0.125), args.iters)
        fl = 4.0 * B * H * T * S * 64 * (0.5 if causal else 1.0)
        byt_f = (B * T * D * 2 + 2 * B * S * D) * 2
        byt_b = (B * T * D * 4 + 4 * B * S * D + 2 * B * T * D) * 2
        print(f'{name:12s} fwd {fwd:7.1f} us ({fl / fwd / 1e6:6.1f} TF, {byt_f / fwd / 1e6:5.2f} TB/s)   '
              f'bwd {bwd:7.1f} us ({2.5 * fl / bwd / 1e6:6.1f} TF, {byt_b / bwd / 1e6:5.2f} TB/s)', flush=True)


if __name__ == '__main__':
    main()
