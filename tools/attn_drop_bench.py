#!/usr/bin/env python3
"""Attention kernels with and without attention-probability dropout at the shapes of the IWSLT2023 recipe (B = 32, 16 heads,
500 encoder positions, T = 64): forward and backward, bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, B, H, T, S, causal in [('iwslt_enc', 32, 16, 500, 500, False), ('iwslt_cross', 32, 16, 64, 500, False), ('iwslt_dec', 32, 16, 64, 64, True)]:
    D = H * 64
    q = torch.randn(B, T, D, device='cuda').bfloat16(); k = torch.randn(B, S, D, device='cuda').bfloat16(); v = torch.randn(B, S, D, device='cuda').bfloat16()
    for pd in (0.0, 0.1):
        if pd:
            o, lse, mask = F.attn_fwd(q, k, v, H, None, causal, 0.125, pd, 7, 3)
            fwd = bench(lambda: F.attn_fwd(q, k, v, H, None, causal, 0.125, pd, 7, 3))
            do = torch.randn_like(o)
            bwd = bench(lambda: F.attn_bwd(q, k, v, o, do, lse, H, None, causal, 0.125, drop_p=pd, drop_mask=mask))
        else:
            o, lse = F.attn_fwd(q, k, v, H, None, causal, 0.125)
            fwd = bench(lambda: F.attn_fwd(q, k, v, H, None, causal, 0.125))
            do = torch.randn_like(o)
            bwd = bench(lambda: F.attn_bwd(q, k, v, o, do, lse, H, None, causal, 0.125))
        print(f'{name:12s} p={pd}: fwd {fwd:7.1f} us  bwd {bwd:7.1f} us', flush=True)
