"""`utils.benchmark` for the mirror classes (pasero/utils.py:1003-1174) + roctx ranges for rocprofv3.

The reference wraps `MultiheadAttention.forward` ('attention', modules.py:578), `Transformer.compute_loss` ('loss',
transformer.py:323), the encoder / decoder forward ('encoder' / 'decoder', transformer.py:697,830) and the output
projection ('output_projection', transformer.py:892) in `utils.benchmark(name)`, so that `pasero-train --benchmark`
logs `<name>_wall`, `<name>_mem`, `<name>_peak_mem` and `max_mem` (cli/train.py:109-110).  The classes of this package
keep the same names on the same calls:

  * inside the reference tree (`pasero.utils` importable) `benchmark` IS `pasero.utils.benchmark`: the Trainer's own
    phases ('forward', 'backward', 'optimizer', ...) and these land in one object, the log lines look the same;
  * stand-alone, `Benchmark` below restates that class (same `enable / disable / pause / reset / metrics` surface, same
    metric names and units: seconds, MiB).

Both synchronise the device at the edges of a block while enabled — that is the reference's measuring method and the
reason it is off by default.  Independently of it, `PASERO_ROCTX=1` (or `roctx(True)`) brackets the same blocks in
roctx ranges (`torch.cuda.nvtx` is roctx on ROCm): `rocprofv3 --marker-trace --kernel-trace` then groups the kernels
by component without any synchronisation."""
import contextlib
import functools
import os
import time

import torch


class Benchmark:
    def __init__(self, use_cuda: bool = True, enabled: bool = True):
        self.use_cuda = use_cuda and torch.cuda.is_available()
        self.enabled = enabled
        self.timers, self.mem_usage, self.peak_mem_usage, self.ongoing = {}, {}, {}, {}
        self.max_mem = 0

    def reset(self) -> None:
        self.timers.clear()
        self.mem_usage.clear()
        self.peak_mem_usage.clear()
        self.ongoing.clear()
        self.max_mem = 0
        if self.use_cuda:
            torch.cuda.reset_peak_memory_stats()

    @contextlib.contextmanager
    def __call__(self, name: str):
        if not self.enabled or name in self.ongoing:  # (nested blocks of one name count once, like the reference)
            yield
            return
        before = 0
        if self.use_cuda:
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated()
            for k in self.ongoing:
                self.ongoing[k] = max(self.ongoing[k], peak)
            self.max_mem = max(self.max_mem, peak)
            torch.cuda.reset_peak_memory_stats()
            before = torch.cuda.memory_allocated()
            self.ongoing[name] = before
        start = time.perf_counter()
        try:
            yield
        finally:
            if self.use_cuda:
                torch.cuda.synchronize()
                after = max(torch.cuda.max_memory_allocated(), self.ongoing.pop(name))
                self.mem_usage[name] = max(after - before, self.mem_usage.get(name, 0))
                self.peak_mem_usage[name] = max(after, self.peak_mem_usage.get(name, 0))
            self.timers[name] = self.timers.get(name, 0) + time.perf_counter() - start

    @property
    def metrics(self) -> dict:
        if self.use_cuda:
            self.max_mem = max(self.max_mem, torch.cuda.max_memory_allocated())
        out = {f'{k}_wall': v for k, v in self.timers.items()}
        if self.use_cuda:
            out['max_mem'] = self.max_mem / 2 ** 20
            out.update({f'{k}_mem': v / 2 ** 20 for k, v in self.mem_usage.items()})
            out.update({f'{k}_peak_mem': v / 2 ** 20 for k, v in self.peak_mem_usage.items()})
        return out

    @contextlib.contextmanager
    def pause(self):
        enabled, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = enabled

    def enable(self) -> None:
        self.enabled = True

    def disable(self) -> None:
        self.enabled = False

    def cpu(self) -> None:
        self.use_cuda = False


def _reference_benchmark():
    try:
        from pasero import utils  # the drop-in case: one shared object with the Trainer's phases
        return utils.benchmark
    except Exception:
        return None


benchmark = _reference_benchmark() or Benchmark(enabled=False)

_roctx = os.environ.get('PASERO_ROCTX', '0') not in ('', '0')


def roctx(on: bool) -> None:
    global _roctx
    _roctx = bool(on)


def region(name: str):
    """decorator for the model methods the reference decorates with `@utils.benchmark(name)`: the benchmark block and,
    if asked for, a roctx range.  Disabled (the default) it costs two attribute reads per call."""
    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            if not (benchmark.enabled or _roctx):
                return fn(*args, **kwargs)
            with block(name):
                return fn(*args, **kwargs)
        return wrapped
    return deco


@contextlib.contextmanager
def block(name: str):
    """`with utils.benchmark(name):` (+ roctx range)"""
    if _roctx and torch.cuda.is_available():
        torch.cuda.nvtx.range_push(name)
    try:
        if benchmark.enabled:
            with benchmark(name):
                yield
        else:
            yield
    finally:
        if _roctx and torch.cuda.is_available():
            torch.cuda.nvtx.range_pop()
