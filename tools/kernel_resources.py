#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel in a .hip file (cross-compiles to gfx950 assembly, reads the metadata).
Usage: python tools/kernel_resources.py pasero_amd/csrc/attention.hip [more.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

import yaml


def waves_per_simd(vgprs: int) -> int:
    alloc = max(8, (vgprs + 7) // 8 * 8)  # unified VGPR+AGPR file of 512 per lane per SIMD, granularity 8
    return max(1, min(8, 512 // alloc))


def main():
    for src in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'k.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only',
                            '-S', src, '-o', out], check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
        block = text[text.index('.amdgpu_metadata') + len('.amdgpu_metadata'):text.index('.end_amdgpu_metadata')]
        meta = yaml.safe_load('\n'.join(line for line in block.splitlines() if line.strip() and not line.startswith('\t')))
        print(f'== {src}')
        for k in meta['amdhsa.kernels']:
            name = subprocess.run(['c++filt', k['.name']], capture_output=True,
                                  text=True).stdout.strip()
            name = re.sub(r'\(anonymous namespace\)::', '', name)
            name = re.sub(r'^void ', '', name).split('(')[0]
            v = k['.vgpr_count'] + 0
            w = waves_per_simd(v)
            wg_waves = max(1, k['.max_flat_workgroup_size'] // 64)
            lds = k['.group_segment_fixed_size']
            by_lds = (160 * 1024 // lds) if lds else 99
            print(f'  {name[:78]:78s} vgpr {v:3d} spill {k[".vgpr_spill_count"]:3d} lds {lds:6d}  waves/SIMD {w}'
                  f'  WGs/CU by regs {w * 4 // wg_waves if wg_waves <= 4 * w else 0}, by LDS {min(by_lds, 32)}')


if __name__ == '__main__':
    main()
