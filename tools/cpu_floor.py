#!/usr/bin/env python3
"""CPU enqueue floor of a training step: the C2 model on a batch so small that the GPU is idle most of the time — what is
left is the host time of one forward + backward (Python, autograd engine, ctypes, allocator).  With --profile prints the
top of a cProfile of 20 steps."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import paramgen  # noqa: E402
from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask  # noqa: E402
from pasero_amd.transformer import Transformer  # noqa: E402


def main():
    V = 8032
    model = Transformer(TransformerConfig(dropout=0.1), DistributedConfig(), SyntheticTask(V)).to(torch.bfloat16).cuda().train()
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(1, 2, 128, 128, V, ragged=False).items()}

    def step():
        model.zero_grad(set_to_none=True)
        loss, _ = model(**batch)
        loss.backward()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step()
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f'host enqueue {t_enq * 1e3:.2f} ms per step (with the final sync {t_all * 1e3:.2f} ms)', flush=True)
    if '--profile' in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            step()
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats('tottime').print_stats(35)


if __name__ == '__main__':
    main()
