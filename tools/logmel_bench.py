#!/usr/bin/env python3
"""pk_logmel micro-benchmark: us per call for B clips of 30 s (default 16: the C4 Whisper batch), and the feature checksum."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasero_amd import functional as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wav = (0.1 * torch.randn(B, 480000, generator=torch.Generator().manual_seed(0))).cuda()
for _ in range(3):
    out = F.log_mel(wav)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    out = F.log_mel(wav)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / 50
print(f'pk_logmel, {B} clips of 30 s: {us:.1f} us per call ({B * (480000 * 4 + 3000 * 80 * 4) / us / 1e6:.2f} TB/s of its algorithmic bytes); '
      f'checksum {out.double().sum().item():.6f}')
