#!/usr/bin/env python3
"""Uninitialised-read hunt: run a fixture's train step and no-grad evaluation twice — once on a fresh allocator, once
after filling the caching allocator's free blocks with NaN bit patterns — and compare losses / logits.  A kernel that
reads memory it (or a predecessor) never wrote shows up as a difference or a NaN.
Usage: python tools/poison_check.py [fixture ...]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))


def poison():
    """leave NaN-filled free blocks of many sizes in the caching allocator"""
    torch.cuda.synchronize()
    held = []
    for shift in range(9, 28):          # 512 B .. 128 MiB
        for _ in range(6 if shift < 24 else 2):
            t = torch.empty((1 << shift) // 4, dtype=torch.float32, device='cuda')
            t.fill_(float('nan'))
            held.append(t)
    torch.cuda.synchronize()
    del held


def run(name, dtype, poisoned):
    from conftest import load_golden
    from model_utils import build_model, text_batch
    g = load_golden(name)
    cfg, model = build_model(g, dtype, 'cuda')
    batch = text_batch(g, 'cuda')
    out = {}
    model.train()
    if poisoned:
        poison()
    loss, _ = model(**batch)
    loss.backward()
    out['train_loss'] = loss.item()
    out['grad_sq'] = sum(p.grad.double().pow(2).sum().item() for p in model.parameters() if p.grad is not None)
    model.eval()
    with torch.no_grad():
        if poisoned:
            poison()
        enc_out, enc_mask, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        if poisoned:
            poison()
        logits, _ = model.decoder(enc_out, enc_mask, batch['decoder_input'][:, :-1])
        if poisoned:
            poison()
        loss2, _ = model(**batch)
    out['eval_loss'] = loss2.item()
    out['enc_sum'] = enc_out.double().sum().item()
    out['logits_sum'] = logits.double().sum().item()
    return out


def main():
    names = sys.argv[1:] or ['base_c1', 'tiny_encdec_post', 'tiny_encdec_pre', 'tiny_encdec_rotary', 'tiny_encdec_swiglu',
                             'tiny_encdec_rms', 'tiny_hd128', 'tiny_lora', 'tiny_adapter']
    bad = 0
    for name in names:
        for dtype in (torch.float32, torch.bfloat16):
            clean = run(name, dtype, False)
            dirty = run(name, dtype, True)
            for k in clean:
                same = clean[k] == dirty[k]
                if not same:
                    bad += 1
                    print(f'DIFF {name} {dtype} {k}: clean {clean[k]!r} poisoned {dirty[k]!r}', flush=True)
            print(f'{name} {dtype}: checked {list(clean)}', flush=True)
    print('poison_check:', 'FAILED' if bad else 'ok', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
