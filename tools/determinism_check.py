#!/usr/bin/env python3
"""Race hunt: the forward pass has no atomics, so repeated runs on the same inputs must be BITWISE identical.
Runs a fixture's no-grad evaluation (and train-mode forward) many times with a forward hook on every sub-module; the
first module whose output bits differ from iteration 0 is reported.
Usage: python tools/determinism_check.py [--iters N] [--dtype f32|bf16] [fixture ...]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))


def digest(t: torch.Tensor):
    t = t.detach().contiguous()
    v = t.view(torch.int16 if t.element_size() == 2 else torch.int32) if t.is_floating_point() else t
    return int(v.to(torch.int64).sum().item()), int((v.to(torch.int64) * 31 % 1000003).sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--train', action='store_true', help='train-mode forward+backward instead of no-grad evaluation')
    ap.add_argument('names', nargs='*', default=['base_c1'])
    args = ap.parse_args()
    from conftest import load_golden
    from model_utils import build_model, text_batch
    dtype = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[args.dtype]
    bad = 0
    for name in args.names:
        g = load_golden(name)
        cfg, model = build_model(g, dtype, 'cuda')
        batch = text_batch(g, 'cuda')
        record = []

        def hook(mod_name):
            def fn(mod, inp, out):
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for i, o in enumerate(outs):
                    if torch.is_tensor(o) and o.numel():
                        record.append((f'{mod_name}[{i}]', digest(o)))
            return fn
        for n, m in model.named_modules():
            if n:
                m.register_forward_hook(hook(n))
        model.train(args.train)
        first = None
        for it in range(args.iters):
            record.clear()
            if args.train:
                for p in model.parameters():
                    p.grad = None
                loss, _ = model(**batch)
                loss.backward()
                for n, p in model.named_parameters():
                    if p.grad is not None and 'embed_tokens' not in n:  # embedding gradient: fp32 atomics, order-dependent
                        record.append(('grad:' + n, digest(p.grad)))
            else:
                with torch.no_grad():
                    loss, _ = model(**batch)
            record.append(('loss', digest(loss.float().view(1))))
            if first is None:
                first = list(record)
                continue
            diffs = [(a[0], a[1], b[1]) for a, b in zip(first, record) if a != b]
            if diffs or len(first) != len(record):
                bad += 1
                print(f'{name} iter {it}: {len(diffs)} outputs differ; first: {diffs[:3]}', flush=True)
            if it % 50 == 0:
                print(f'{name} iter {it} ok so far (bad={bad})', flush=True)
    print('determinism_check:', 'FAILED' if bad else 'ok', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
