#!/usr/bin/env python3
"""Build-time audit of the attention kernels that stream tiles by LDS-DMA (csrc/attention_long.hip): the NEXT tile's pieces are
requested at the top of a tile and must stay in flight under its arithmetic.  hipcc's wait-count pass drains `vmcnt` in front of
a `ds_read` it knows may alias an LDS-DMA — which puts the whole flight time in front of the tile's first MFMA (round 5: that is
what `attn_dkv_long_kernel` did for a round).  Rule: in every kernel that issues `buffer_load ... lds` inside a loop, no
compiler-placed `s_waitcnt vmcnt(0)` (one outside an inline-asm block) may stand between the last DMA request of the loop body
and the first MFMA behind it.  usage: check_asm_dma.py <file.s>; prints one line per kernel, the last line is
'<n> kernels audited, <p> problems'; exit status 1 on a problem."""
import re
import subprocess
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:\s', l + ' ')]
    audited = problems = 0
    for si, s in enumerate(starts):
        e = next((j for j in range(s, len(lines)) if lines[j].startswith('.Lfunc_end')), len(lines))
        body = lines[s:e]
        loop0 = next((j for j, l in enumerate(body) if 'Loop Header' in l), None)
        if loop0 is None:
            continue
        dma = [j for j, l in enumerate(body) if j > loop0 and re.search(r'buffer_load_dword.* lds\b', l)]
        if not dma:
            continue
        name = subprocess.run(['c++filt', body[0].split(':')[0]], capture_output=True, text=True).stdout.strip()
        name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        audited += 1
        last = dma[-1]
        in_asm, bad = False, None
        for j in range(last + 1, len(body)):
            l = body[j]
            if '#ASMSTART' in l:
                in_asm = True
            elif '#ASMEND' in l:
                in_asm = False
            elif 'v_mfma' in l:
                break
            elif not in_asm and re.search(r's_waitcnt\s+vmcnt\(0\)', l):
                bad = j
                break
        if bad is None:
            print(f'ok       {name}: {len(dma)} LDS-DMA requests in loops, none waited for before the first MFMA behind them')
        else:
            problems += 1
            print(f'PROBLEM  {name}: line {s + bad + 1}: a compiler-placed `s_waitcnt vmcnt(0)` between the tile prefetch '
                  f'(line {s + last + 1}) and the first MFMA behind it: the prefetch is waited for where it was requested')
    print(f'{audited} kernels audited, {problems} problems')
    sys.exit(1 if problems else 0)


if __name__ == '__main__':
    main()
