#!/usr/bin/env python3
"""LayerNorm backward / forward micro-benchmark: us per launch and TB/s of algorithmic bytes at the training shapes, and
(with --compare) the specialised 16-bit backward kernel against the general one on the same inputs (PK_LN_BWD16=0 in a
child process): largest difference of every output.

    python tools/ln_bench.py [--rows 32768] [--iters 200] [--compare | --widths]"""
import argparse
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402


def inputs(rows, d, dtype, extra):
    g = torch.Generator(device='cuda').manual_seed(rows + d)
    r = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = r(rows, d).to(dtype)
    gamma = (1 + 0.1 * r(d)).to(dtype)
    beta = (0.1 * r(d)).to(dtype)
    y, _, mean, rstd = F.residual_ln_fwd(x, None, gamma, beta, 1e-5, want_z=False)
    dy = r(rows, d).to(dtype)
    ex = r(rows, d).to(dtype) if extra else None
    return x, gamma, mean, rstd, dy, ex


def run(rows, d, dtype, extra, p):
    x, gamma, mean, rstd, dy, ex = inputs(rows, d, dtype, extra)
    return F.residual_ln_bwd(dy, ex, x, gamma, mean, rstd, want_dres=True, want_dx=p > 0, want_param_grads=True,
                             drop_p=p, seed=1234, offset=77)


def run_fwd(rows, d, dtype, mode, p):
    g = torch.Generator(device='cuda').manual_seed(rows * 3 + d)
    r = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x, res = r(rows, d).to(dtype), r(rows, d).to(dtype)
    gamma, beta = (1 + 0.1 * r(d)).to(dtype), (0.1 * r(d)).to(dtype)
    if mode == 'ln':       # LayerNorm alone (pre-norm blocks)
        return F.residual_ln_fwd(x, None, gamma, beta, 1e-5, want_z=False)
    if mode == 'block':    # post-norm block end
        return F.residual_ln_fwd(x, res, gamma, beta, 1e-5, p, 1234, 77)
    return F.residual_ln_fwd(x, res, None, None, 0.0, p, 1234, 77)  # residual + dropout only


FWD_CASES = [(d, dt, mode, p) for d in (512, 1024) for dt in (torch.bfloat16, torch.float16)
             for mode, p in (('ln', 0.0), ('block', 0.0), ('block', 0.1), ('res', 0.0), ('res', 0.1))]


def timeit(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=32768)
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--compare', action='store_true')
    ap.add_argument('--dump', default=None)
    ap.add_argument('--widths', action='store_true',
                    help='the round-2 sweep: forward and backward over the model widths of the presets (512 ... 4096, bf16)')
    args = ap.parse_args()
    if args.widths:
        for d in (512, 1024, 1280, 2048, 4096):
            rows = (32768 * 512) // d
            x = torch.randn(rows, d, device='cuda').bfloat16()
            res = torch.randn(rows, d, device='cuda').bfloat16()
            g = torch.ones(d, device='cuda').bfloat16()
            b = torch.zeros(d, device='cuda').bfloat16()
            dy = torch.randn(rows, d, device='cuda').bfloat16()
            y, z, mean, rstd = F.residual_ln_fwd(x, res, g, b, 1e-5)
            nbytes = rows * d * 2
            t_f = timeit(lambda: F.residual_ln_fwd(x, res, g, b, 1e-5), args.iters)       # reads x, res; writes z, y
            t_b = timeit(lambda: F.residual_ln_bwd(dy, None, z, g, mean, rstd, want_dres=True, want_dx=False,
                                                   want_param_grads=True), args.iters)     # reads dy, z; writes dres
            print(f'd={d:5d} rows={rows:6d}  fwd {t_f:6.1f} us ({4 * nbytes / t_f / 1e6:5.2f} TB/s)   '
                  f'bwd {t_b:6.1f} us ({3 * nbytes / t_b / 1e6:5.2f} TB/s)', flush=True)
        return
    cases = [(d, dt, extra, p) for d in (512, 1024) for dt in (torch.bfloat16, torch.float16)
             for extra, p in ((False, 0.1), (True, 0.0), (True, 0.1), (False, 0.0))]
    if args.dump:  # child of --compare: outputs of every case to a file
        out = {}
        for d, dt, extra, p in cases:
            out[(d, str(dt), extra, p)] = [None if t is None else t.float().cpu() for t in run(4096 + 3, d, dt, extra, p)]
        for d, dt, mode, p in FWD_CASES:
            out[('fwd', d, str(dt), mode, p)] = [None if t is None else t.float().cpu() for t in run_fwd(4096 + 3, d, dt, mode, p)]
        torch.save(out, args.dump)
        return
    if args.compare:
        path = '/tmp/ln_bench_general.pt'
        env = dict(os.environ, PK_LN_BWD16='0', PK_LN_FWD16='0')
        subprocess.run([sys.executable, os.path.abspath(__file__), '--dump', path], env=env, check=True)
        ref = torch.load(path)
        worst = 0.0
        for d, dt, extra, p in cases:
            got = [None if t is None else t.float().cpu() for t in run(4096 + 3, d, dt, extra, p)]
            for name, a, b in zip(('dres', 'dx', 'dgamma', 'dbeta'), got, ref[(d, str(dt), extra, p)]):
                if a is None:
                    assert b is None
                    continue
                diff = (a - b).abs().max().item()
                rel = diff / max(b.abs().max().item(), 1e-9)
                nz = (a != b).float().mean().item()
                worst = max(worst, rel)
                print(f'd={d} {str(dt)[6:]:8s} extra={int(extra)} p={p}: {name:6s} max|diff| {diff:.3e} (rel {rel:.2e}), '
                      f'{100 * nz:.3f} % of elements differ')
        for d, dt, mode, p in FWD_CASES:
            got = [None if t is None else t.float().cpu() for t in run_fwd(4096 + 3, d, dt, mode, p)]
            for name, a, b in zip(('y', 'z', 'mean', 'rstd'), got, ref[('fwd', d, str(dt), mode, p)]):
                if a is None:
                    assert b is None, (d, dt, mode, p, name)
                    continue
                diff = (a - b).abs().max().item()
                rel = diff / max(b.abs().max().item(), 1e-9)
                worst = max(worst, rel)
                print(f'fwd d={d} {str(dt)[6:]:8s} {mode:5s} p={p}: {name:5s} max|diff| {diff:.3e} (rel {rel:.2e}), '
                      f'{100 * (a != b).float().mean().item():.3f} % of elements differ')
        print('worst relative difference', worst)
        return
    for d, dt, extra, p in cases:
        if dt is torch.float16:
            continue
        x, gamma, mean, rstd, dy, ex = inputs(args.rows, d, dt, extra)
        fn = lambda: F.residual_ln_bwd(dy, ex, x, gamma, mean, rstd, want_dres=True, want_dx=p > 0, want_param_grads=True,
                                       drop_p=p, seed=1234, offset=77)
        us = timeit(fn, args.iters)
        streams = 3 + int(extra) + int(p > 0)
        gb = streams * args.rows * d * 2 / 1e9
        print(f'bwd rows={args.rows} d={d} extra={int(extra)} p={p}: {us:7.1f} us  ({gb / us * 1e6 / 1e3:.2f} TB/s of {streams} streams; '
              f'incl. the parameter-gradient reduction launch)')
    for rows in sorted({args.rows, 8192}):
        for d in (512, 1024):
            for mode, p in (('ln', 0.0), ('res', 0.1)):
                g = torch.Generator(device='cuda').manual_seed(d)
                x = torch.randn(rows, d, device='cuda', generator=g).bfloat16()
                res = torch.randn(rows, d, device='cuda', generator=g).bfloat16()
                gamma, beta = torch.ones(d, device='cuda').bfloat16(), torch.zeros(d, device='cuda').bfloat16()
                if mode == 'ln':
                    fn = lambda: F.residual_ln_fwd(x, None, gamma, beta, 1e-5, want_z=False)
                else:
                    fn = lambda: F.residual_ln_fwd(x, res, None, None, 0.0, p, 1234, 77)
                us = timeit(fn, args.iters)
                streams = 2 if mode == 'ln' else 3
                print(f'fwd {mode:3s} rows={rows} d={d} p={p}: {us:7.1f} us  ({streams * rows * d * 2 / us / 1e6:.2f} TB/s of {streams} streams)')
    for d in (512, 1024):
        g = torch.Generator(device='cuda').manual_seed(d)
        x = torch.randn(args.rows, d, device='cuda', generator=g).bfloat16()
        res = torch.randn(args.rows, d, device='cuda', generator=g).bfloat16()
        gamma = torch.ones(d, device='cuda').bfloat16()
        beta = torch.zeros(d, device='cuda').bfloat16()
        for p in (0.0, 0.1):
            fn = lambda: F.residual_ln_fwd(x, res, gamma, beta, 1e-5, p, 1234, 77)
            us = timeit(fn, args.iters)
            gb = 4 * args.rows * d * 2 / 1e9
            print(f'fwd rows={args.rows} d={d} p={p}: {us:7.1f} us  ({gb / us * 1e6 / 1e3:.2f} TB/s of 4 streams)')


if __name__ == '__main__':
    main()
