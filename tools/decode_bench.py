#!/usr/bin/env python3
"""Greedy incremental decoding micro-benchmark (the `pasero-decode` inner loop, decoding.py:1119-1221, on synthetic
inputs): one BOS column, then one decoder call per generated token with the incremental `state`.
Reports ms per step and generated tokens/s.  usage: tools/decode_bench.py [--batch 64] [--src-len 64] [--steps 64]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--src-len', type=int, default=64)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--config', default='TransformerConfig')
    ap.add_argument('--no-cross-cache', action='store_true')
    args = ap.parse_args()
    import paramgen
    from pasero_amd import config as C, functional as F
    from pasero_amd.transformer import Transformer
    V = 8032
    cfg = getattr(C, args.config)()
    torch.manual_seed(0)
    model = Transformer(cfg, C.DistributedConfig(), C.SyntheticTask(V))
    model = model.to({'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]).cuda().eval()
    if args.no_cross_cache:
        os.environ['PASERO_NO_CROSS_KV_CACHE'] = '1'
    b = paramgen.make_text_batch(1, args.batch, args.src_len, 4, V, ragged=True)
    enc_in = torch.from_numpy(b['encoder_input']).cuda()
    enc_len = torch.from_numpy(b['encoder_input_length']).cuda()

    def decode():
        with torch.no_grad():
            enc_out, enc_mask, _ = model.encoder(enc_in, enc_len)
            tokens = torch.full((args.batch, args.steps + 1), cfg.padding_idx, dtype=torch.long, device='cuda')
            tokens[:, 0] = cfg.bos_idx
            state = {}
            for step in range(1, args.steps + 1):
                logits, _ = model.decoder(enc_out, enc_mask, tokens[:, step - 1:step], state=state)
                F.argmax_rows(logits[:, -1], out=tokens[:, step])
        return tokens

    decode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        tokens = decode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{args.config} {args.dtype} B={args.batch} S={args.src_len} steps={args.steps}: '
          f'{1e3 * dt / args.steps:.3f} ms/step, {args.batch * args.steps / dt:,.0f} generated tokens/s '
          f'(checksum {int(tokens.sum())})')


if __name__ == '__main__':
    main()
