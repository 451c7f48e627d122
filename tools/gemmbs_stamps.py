#!/usr/bin/env python3
"""Where a workgroup of the B-stationary GEMM (csrc/gemmbs.hip) spends its time: B panel into registers / ring priming /
the walk / store drain, and the shader clock the chip holds inside the walk.

Needs a DIAGNOSTIC build with -DPKBS_STAMPS (tools/gemmbs_ablate.sh builds libpasero_hip_STAMPS.so; the shipped build has
no stamps):   PASERO_HIP_LIB=pasero_amd/libpasero_hip_STAMPS.so python tools/gemmbs_stamps.py
Thread 0 of every workgroup writes s_memrealtime (100 MHz) and s_memtime (shader clock) at five seams into a buffer of its
own (PK8P_STAMP_PTR)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(1024 * 64, dtype=torch.int64, device='cuda')
os.environ['PK8P_STAMP_PTR'] = hex(buf.data_ptr())
from pasero_amd import functional as F  # noqa: E402


def main():
    for (M, N, b_col) in [(32768, 2048, False), (32768, 2048, True), (32768, 1536, False), (32768, 512, False), (32768, 512, True)]:
        a = torch.randn(M, 512, device='cuda').bfloat16()
        w = torch.randn(N, 512, device='cuda').bfloat16()
        b = w.t().contiguous() if b_col else w
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        for _ in range(3):
            F.gemm(a, b, b_col=b_col, out=out)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        F.gemm(a, b, b_col=b_col, out=out)
        torch.cuda.synchronize()
        s = buf.view(1024, 64).cpu()
        nb = int((s[:, 0] != 0).sum())
        if nb == 0:
            raise SystemExit('no stamps: is PASERO_HIP_LIB a -DPKBS_STAMPS build?')
        t = s[:nb, :5].double() / 100.0   # us
        c = s[:nb, 32:37].double()
        t0 = t[:, 0].min()
        steps = -(-M // 32) / (nb / -(-N // 256))
        loop = (t[:, 3] - t[:, 2])
        ghz = ((c[:, 3] - c[:, 2]) / loop / 1e3).mean().item()
        print(f'M={M} N={N} B {"col" if b_col else "row"}: {nb} workgroups x {steps:.1f} steps, kernel span {float(t.max() - t0):6.1f} us '
              f'(starts within {float(t[:, 0].max() - t0):4.1f}) | per workgroup: B panel {float((t[:, 1] - t[:, 0]).mean()):5.2f}  '
              f'ring priming {float((t[:, 2] - t[:, 1]).mean()):5.2f}  walk {float(loop.mean()):6.2f} '
              f'({float(loop.mean()) / (steps * 8) * 1e3:5.1f} ns per K-tile, clock {ghz:4.2f} GHz)  drain '
              f'{float((t[:, 4] - t[:, 3]).mean()):5.2f} us', flush=True)


if __name__ == '__main__':
    main()
