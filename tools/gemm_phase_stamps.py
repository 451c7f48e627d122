#!/usr/bin/env python3
"""Where a tile of the phase-interleaved GEMM (csrc/gemm8p.hip) spends its time: prologue / K loop / epilogue + store
drain per workgroup, and the shader clock the chip holds inside the K loop.

Needs a DIAGNOSTIC build of the library with -DPK8P_STAMPS (the shipped build has no stamps):
    cd pasero_amd/csrc && cp *.o /tmp/st/ && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPK8P_STAMPS \
        -c gemm8p.hip -o /tmp/st/gemm8p.o && hipcc -shared -fPIC --offload-arch=gfx950 /tmp/st/*.o -o /tmp/libpasero_st.so
    PASERO_HIP_LIB=/tmp/libpasero_st.so python tools/gemm_phase_stamps.py
(add -DPK8P_ABL_NODMA / -DPK8P_ABL_NOMFMA for the ablation builds quoted in DESIGN.md §4).  Thread 0 of every workgroup
writes s_memrealtime (100 MHz) and s_memtime (shader clock) at four seams — tile start, first K-tile landed, K loop done,
epilogue stores acknowledged — into a buffer of its own (PK8P_STAMP_PTR); the in-kernel clock of the K loop is
delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(1024 * 64, dtype=torch.int64, device='cuda')
os.environ['PK8P_STAMP_PTR'] = hex(buf.data_ptr())  # read by the library at its first gemm8p launch
from pasero_amd import functional as F  # noqa: E402


def main():
    for (M, N, K) in [(32768, 2048, 512), (32768, 1536, 512), (32768, 4096, 1024), (32768, 512, 2048), (4096, 4096, 4096)]:
        a = torch.randn(M, K, device='cuda').bfloat16()
        b = torch.randn(N, K, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        for _ in range(3):
            F.gemm(a, b, out=out)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        F.gemm(a, b, out=out)
        torch.cuda.synchronize()
        s = buf.view(1024, 64).cpu()
        nb = int((s[:, 0] != 0).sum())
        if nb == 0:
            raise SystemExit('no stamps: is PASERO_HIP_LIB a -DPK8P_STAMPS build?')
        t = s[:nb, :4].double() / 100.0   # us
        c = s[:nb, 32:36].double()        # shader clock ticks
        t0 = t[:, 0].min()
        ghz = ((c[:, 2] - c[:, 1]) / (t[:, 2] - t[:, 1]) / 1e3).mean().item()
        print(f'M={M} N={N} K={K}: {nb} workgroups, kernel span {float(t.max() - t0):6.1f} us | per workgroup: prologue '
              f'{float((t[:, 1] - t[:, 0]).mean()):5.2f}  K loop {float((t[:, 2] - t[:, 1]).mean()):6.2f} '
              f'({float((t[:, 2] - t[:, 1]).mean()) / (K / 64):4.2f} us per K-tile, clock {ghz:4.2f} GHz)  '
              f'epilogue + drain {float((t[:, 3] - t[:, 2]).mean()):5.2f} us', flush=True)


if __name__ == '__main__':
    main()
