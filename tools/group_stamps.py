#!/usr/bin/env python3
"""Per-workgroup timing of the grouped weight-gradient launch (csrc/gemm8p.hip: gemm8p_group_kernel) at the C2 layer shapes:
start skew, K-loop duration per (problem, K-slab) unit and per XCD — what showed in round 4 that the workgroups carrying the
fused bias gradient ran 25 % longer than the others and set the launch's duration.

Needs a DIAGNOSTIC build of the library with -DPK8P_STAMPS (the shipped build has no stamps), as tools/gemm_phase_stamps.py:
    cd pasero_amd/csrc && mkdir -p /tmp/st && cp *.o /tmp/st/ && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPK8P_STAMPS \
        -c gemm8p.hip -o /tmp/st/gemm8p.o && hipcc -shared -fPIC --offload-arch=gfx950 /tmp/st/*.o -o /tmp/libpasero_st.so
    PASERO_HIP_LIB=/tmp/libpasero_st.so python tools/group_stamps.py        (NOBIAS=1: without the bias gradients;
    PK_WGRAD_MAP=0: the contiguous launch map instead of the XCD-packed one)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(2048 * 64, dtype=torch.int64, device='cuda')
os.environ['PK8P_STAMP_PTR'] = hex(buf.data_ptr())
from pasero_amd import functional as F
d, f, rows = 512, 2048, 32768
def prob(n_out, k_in):
    return (torch.randn(rows, n_out, device='cuda').bfloat16(), torch.randn(rows, k_in, device="cuda").bfloat16(), os.environ.get("NOBIAS") is None)
layers = {'enc': [prob(3 * d, d), prob(d, d), prob(f, d), prob(d, f)],
          'dec': [prob(3 * d, d), prob(d, d), prob(d, d), prob(2 * d, d), prob(d, d), prob(f, d), prob(d, f)]}
for name, entries in layers.items():
    for _ in range(3): F.wgrad_group(entries)
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    F.wgrad_group(entries); torch.cuda.synchronize()
    s = buf.view(2048, 64).cpu()
    live = (s[:, 0] != 0)
    idx = live.nonzero().flatten()
    t = s[idx, :4].double() / 100.0
    t0 = t[:, 0].min()
    print(f'{name}: {len(idx)} workgroups, span {float(t.max()-t0):.1f} us; start skew: max {float((t[:,0]-t0).max()):.2f} us;'
          f' first tile landed after {float((t[:,1]-t[:,0]).mean()):.2f} (max {float((t[:,1]-t[:,0]).max()):.2f});'
          f' K loop mean {float((t[:,2]-t[:,1]).mean()):.1f} min {float((t[:,2]-t[:,1]).min()):.1f} max {float((t[:,2]-t[:,1]).max()):.1f};'
          f' epilogue {float((t[:,3]-t[:,2]).mean()):.2f}; loop-end spread {float(t[:,2].max()-t[:,2].min()):.1f}')
    for x in range(8):
        m = (idx % 8 == x)
        st = (t[m, 0] - t0)
        print(f'   xcd {x}: n {int(m.sum())} start {float(st.min()):.2f}..{float(st.max()):.2f} us  loop end {float((t[m,2]-t0).min()):.1f}..{float((t[m,2]-t0).max()):.1f}')
    # per problem / slab
    import ctypes
    from pasero_amd import lib
    L = lib.load()
    shapes = [(dy.size(1), x.size(1), dy.size(0)) for dy, x, _ in entries]
    arr = (lib.PkWgradProblem * len(shapes))(*[lib.PkWgradProblem(16, 16, 16, None, M, N, K, M, N, N) for M, N, K in shapes])
    out = (ctypes.c_int * (2 * 4096))()
    grid = L.pk_gemm_wgrad_group_map(arr, len(shapes), out, 4096)
    import collections, math
    byp = collections.defaultdict(list)
    for row, b in enumerate(idx.tolist()):
        p, lin = out[2 * b], out[2 * b + 1]
        M, N, K = shapes[p]
        tiles = math.ceil(M / 256) * math.ceil(N / 256)
        byp[(p, lin // tiles)].append(float(t[row, 2] - t[row, 1]))
    for k in sorted(byp):
        v = byp[k]
        print(f'   problem {k[0]} {shapes[k[0]][:2]} slab {k[1]}: n {len(v)} K loop mean {sum(v)/len(v):.1f} min {min(v):.1f} max {max(v):.1f}')
