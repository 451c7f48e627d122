"""the long attention kernels, the same launch 40 times: outputs and gradients bitwise equal every time (an LDS-DMA / barrier
race would show as a difference), at shapes that exercise ragged tiles, masks, dropout and the causal forms"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F
torch.manual_seed(0)
bad = 0
for B, H, T, S, causal, drop in [(16, 8, 1500, 1500, False, 0.0), (8, 16, 500, 500, False, 0.1), (4, 8, 1500, 1500, True, 0.0), (16, 8, 64, 1500, False, 0.0),
                                 (8, 4, 333, 777, False, 0.0), (64, 8, 128, 128, False, 0.0), (64, 8, 128, 128, True, 0.0), (4, 8, 700, 700, True, 0.1)]:
    D = H * 64
    qkv = torch.randn(B, max(T, S), 3 * D, device='cuda').bfloat16()
    q, k, v = qkv[:, :T, :D], qkv[:, :S, D:2 * D], qkv[:, :S, 2 * D:]
    pad = None
    if not causal:
        lens = torch.randint(S // 2, S + 1, (B,), device='cuda'); pad = torch.arange(S, device='cuda')[None] >= lens[:, None]
    do = torch.randn(B, T, D, device='cuda').bfloat16()
    ref = None
    for it in range(int(os.environ.get("PK_DET_ITERS", "40"))):
        if drop:
            o, lse, mask = F.attn_fwd(q, k, v, H, pad, causal, 0.125, drop, 7, 3)
            g = F.attn_bwd(q, k, v, o, do, lse, H, pad, causal, 0.125, drop_p=drop, drop_mask=mask)
            cur = (o, lse, mask) + tuple(g)
        else:
            o, lse = F.attn_fwd(q, k, v, H, pad, causal, 0.125)
            g = F.attn_bwd(q, k, v, o, do, lse, H, pad, causal, 0.125)
            cur = (o, lse) + tuple(g)
        cur = [c.clone() for c in cur]
        if ref is None: ref = cur
        else:
            for i, (a, b_) in enumerate(zip(ref, cur)):
                if drop and causal and i == 2: continue  # (keep words of tiles in the causal future are never written nor read)
                if not torch.equal(a, b_):
                    bad += 1; print('DIFF', (B, H, T, S, causal, drop), 'tensor', i, 'iteration', it, flush=True)
    print((B, H, T, S, causal, drop), 'ok' if not bad else 'differences so far: %d' % bad, flush=True)
print('done, differences:', bad)
sys.exit(1 if bad else 0)
