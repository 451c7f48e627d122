"""`utils.benchmark` for the mirror classes (pasero/utils.py:1003-1174) + roctx ranges for rocprofv3.

The reference wraps `MultiheadAttention.forward` ('attention', modules.py:578), `Transformer.compute_loss` ('loss',
transformer.py:323), the encoder / decoder forward ('encoder' / 'decoder', transformer.py:697,830) and the output
projection ('output_projection', transformer.py:892) in `utils.benchmark(name)`, so that `pasero-train --benchmark`
logs `<name>_wall`, `<name>_mem`, `<name>_peak_mem` and `max_mem` (cli/train.py:109-110).  The classes of this package
keep the same names on the same calls:

  * inside the reference tree (`pasero.utils` importable) `benchmark` IS `pasero.utils.benchmark`: the Trainer's own
    phases ('forward', 'backward', 'optimizer', ...) and these land in one object, the log lines look the same;
  * stand-alone, `Benchmark` below offers the same surface (`with b(name):`, `enable / disable / pause / reset / cpu`,
    `metrics`) and the same metric names and units (seconds, MiB) — written for this package, not taken from there.

Both synchronise the device at the edges of a block while enabled — that is the reference's measuring method and the
reason it is off by default.  Independently of it, `PASERO_ROCTX=1` (or `roctx(True)`) brackets the same blocks in
roctx ranges (`torch.cuda.nvtx` is roctx on ROCm): `rocprofv3 --marker-trace --kernel-trace` then groups the kernels
by component without any synchronisation."""
import contextlib
import functools
import os
import time

import torch


class _Stat:
    """what is kept per block name: accumulated wall seconds, the largest growth and the largest peak of device memory"""
    __slots__ = ('seconds', 'growth', 'peak')

    def __init__(self):
        self.seconds, self.growth, self.peak = 0.0, 0, 0


class _Span:
    """one `with recorder(name):` — a plain context object (entered at most once per name at a time: an inner block of a
    name that is already open is transparent, so recursion through a wrapped method is counted once)"""
    __slots__ = ('rec', 'name', 'live', 't0', 'base', 'high')

    def __init__(self, rec, name):
        self.rec, self.name, self.live = rec, name, False

    def __enter__(self):
        rec = self.rec
        if not rec.enabled or self.name in rec._open:
            return self
        self.live, self.base, self.high = True, 0, 0
        if rec.use_cuda:
            self.base = rec._fence_and_fold()  # device idle; the peak so far is credited to every block still open
        rec._open[self.name] = self
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if not self.live:
            return False
        rec, st = self.rec, self.rec._stats.setdefault(self.name, _Stat())
        if rec.use_cuda:
            rec._fence_and_fold()
            st.growth = max(st.growth, self.high - self.base)
            st.peak = max(st.peak, self.high)
        st.seconds += time.perf_counter() - self.t0
        del rec._open[self.name]
        self.live = False
        return False


class Benchmark:
    """Stand-alone stand-in for the object `pasero.utils.benchmark` (used only OUTSIDE the reference tree; inside it that
    object itself is used, see below).  Same surface — callable as `with b(name):`, `enable / disable / pause / reset /
    cpu`, `metrics` — and the same metric names and units: `<name>_wall` seconds, `<name>_mem` / `<name>_peak_mem` /
    `max_mem` MiB.  Memory is sampled with the allocator's high-water mark: at every block edge the device is fenced, the
    mark is read, credited to all blocks open at that moment and reset."""

    def __init__(self, use_cuda: bool = True, enabled: bool = True):
        self.use_cuda = bool(use_cuda) and torch.cuda.is_available()
        self.enabled = bool(enabled)
        self._stats, self._open, self._top = {}, {}, 0

    def __call__(self, name: str) -> _Span:
        return _Span(self, name)

    def _fence_and_fold(self) -> int:
        torch.cuda.synchronize()
        mark = torch.cuda.max_memory_allocated()
        self._top = max(self._top, mark)
        for span in self._open.values():
            span.high = max(span.high, mark)
        torch.cuda.reset_peak_memory_stats()
        return torch.cuda.memory_allocated()

    @property
    def metrics(self) -> dict:
        mib = 1.0 / (1 << 20)
        out = {f'{n}_wall': st.seconds for n, st in self._stats.items()}
        if self.use_cuda:
            self._top = max(self._top, torch.cuda.max_memory_allocated())
            out['max_mem'] = self._top * mib
            for n, st in self._stats.items():
                out[f'{n}_mem'], out[f'{n}_peak_mem'] = st.growth * mib, st.peak * mib
        return out

    def reset(self) -> None:
        self._stats, self._open, self._top = {}, {}, 0
        if self.use_cuda:
            torch.cuda.reset_peak_memory_stats()

    def pause(self):
        rec = self

        class _Paused:
            def __enter__(self):
                self.was, rec.enabled = rec.enabled, False

            def __exit__(self, *exc):
                rec.enabled = self.was
                return False
        return _Paused()

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
