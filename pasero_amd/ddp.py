"""Data-parallel gradient reduction for one-process-per-GPU training: a drop-in for the
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[..], broadcast_buffers=False,
gradient_as_bucket_view=True)` wrap of the reference Trainer (pasero/training.py:243-250), with the same surface the
Trainer touches (`.module`, `__call__`, `.no_sync()`, `.parameters()`, `.train()/.eval()`, `.state_dict()`), plus
`reduce_logs`, the ONE small all-reduce that replaces the per-step `utils.gather_dict` / `all_gather_object` of the
training logs (pasero/utils.py:93-104, called at pasero/training.py:431,538).

Design for 8 x MI355X over xGMI (RCCL):
  * parameters are grouped, in REVERSE registration order (≈ the order autograd produces their gradients: decoder top
    -> encoder bottom -> shared embedding last), into flat buckets of `bucket_cap_mb` (16 MiB: the last, non-overlapped
    all-reduce of a step stays short);
  * a post-accumulate hook per parameter marks it ready; a bucket is launched when it is complete AND every bucket
    before it has been launched (every rank issues the same sequence of collectives, whatever order its gradients
    arrive in — torch's Reducer does the same): its gradients are packed into the flat buffer by one multi-tensor HIP
    kernel (`pk_mt_copy`) and ONE all-reduce (average) is launched on a dedicated communication stream that waits on
    the compute stream's event — the collective overlaps the rest of backward;
  * at the end of backward (autograd engine callback) the remaining buckets are flushed in order, the compute stream
    waits for the communication stream and every `param.grad` is re-pointed at its slice of the reduced bucket (no
    copy back);
  * `find_unused_parameters` (set by AdapterTransformer: ranks may use different adapters in one step, adapters.py:
    232-301): nothing is launched before the end of backward; a bitmap of the parameters that received a gradient is
    all-reduced first, parameters no rank used keep `.grad = None` (the reference's optimizer then skips them, like
    after torch DDP), locally unused ones contribute zeros;
  * `no_sync()` skips the reduction for gradient accumulation (training.py:392-408): only the last micro-batch reduces.
Semantics follow torch DDP: gradients are AVERAGED over ranks, so `Trainer.train_step`'s `grad *= dp_size/num_tokens`
normalisation (training.py:455-477) stays unchanged.
Works with any torch.distributed backend ('nccl' = RCCL on ROCm; 'gloo' on CPU for the tests).  On the GPU with the
'nccl' backend the bucket collectives do not go through torch.distributed at all: `RcclComm` below drives RCCL's C API
from the library's own native entry points (csrc/comm.hip) on the communication stream, with the schedule — RCCL's
all-reduce, reduce-scatter + all-gather, or the direct grouped send/recv exchange over all xGMI links — that it
measured fastest on the actual set of GPUs at construction, after checking each against torch.distributed's result.
"""
import contextlib
import logging
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Variable

logger = logging.getLogger('pasero_amd.ddp')


# tests only (tests/test_ddp_cpu.py): lay natively run layers' parameters out as gradient arenas on CPU tensors too, so that
# the bucket-view layout can be driven over gloo with world sizes the one-GPU box cannot host
_ARENA_ON_ANY_DEVICE = False

class RcclComm:
    """One RCCL communicator per process, created through the C ABI (pk_comm_*): rank 0 draws the unique id, the default
    torch.distributed group carries it to the other ranks (the only use of torch.distributed on this path), every rank
    joins.  `choose_schedule` then runs every schedule of `pk_comm_all_reduce_mean` on random data, requires each to
    agree with `dist.all_reduce(AVG)`, times it, and all ranks adopt the fastest correct one (None: stay on
    torch.distributed).  PASERO_DDP_RCCL=0 disables the native path, PASERO_DDP_SCHEDULE=0|1|2 pins a schedule."""
    SCHEDULES = {0: 'ncclAllReduce', 1: 'ncclReduceScatter + ncclAllGather', 2: 'direct grouped send/recv'}
    _instance = None

    @classmethod
    def get(cls, group, device):
        dev = torch.device(device)
        if cls._instance is None:
            cls._instance = cls(group, dev)
        inst = cls._instance
        if inst.group is not group or inst.device != dev:
            # the library holds ONE communicator per process: a second reducer over another group / device must not
            # silently ride on the first one's ranks
            logger.warning('pasero_amd.ddp: the native RCCL communicator belongs to another process group or device; '
                           'this reducer uses torch.distributed')
            return None
        return inst if inst.schedule is not None else None

    def _all_agree(self, ok: bool) -> bool:
        """True iff `ok` on EVERY rank (one tiny all-reduce over torch.distributed): a rank must not enter a native
        collective its peers have given up on — they would wait for it forever"""
        t = torch.tensor([1.0 if ok else 0.0], device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item() > 0)

    def __init__(self, group, device):
        import ctypes
        from . import lib
        self.lib, self.check = lib.load(), lib.check
        self.device, self.group = device, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.schedule, self.report = None, {}
        self._scratch = None
        # Every step below that can fail on ONE rank alone (library not found, out of memory, a RCCL error) is followed
        # by `_all_agree`: either all ranks go on to the next native call or all fall back to torch.distributed.
        err = None
        try:
            path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            self.check(self.lib.pk_comm_open(path.encode()), 'pk_comm_open')
        except Exception as e:
            err = e
        if self._all_agree(err is None):
            buf = ctypes.create_string_buffer(128)
            try:
                if self.lib.pk_comm_size() == 0 and self.rank == 0:
                    self.check(self.lib.pk_comm_unique_id(buf, 128), 'pk_comm_unique_id')
            except Exception as e:
                err = e
            fresh = self.lib.pk_comm_size() == 0
            if self._all_agree(err is None):
                if fresh:
                    t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=device)
                    dist.broadcast(t, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                    ident = bytes(t.cpu().tolist())
                    try:  # (ncclCommInitRank is itself collective: every rank is here)
                        with torch.cuda.device(device):
                            self.check(self.lib.pk_comm_init(ident, self.world, self.rank), 'pk_comm_init')
                    except Exception as e:
                        err = e
                if self._all_agree(err is None):
                    try:
                        self.choose_schedule()
                    except Exception as e:
                        err, self.schedule = e, None
        if err is not None or self.schedule is None:
            self.schedule = None
            self.report['error'] = repr(err) if err is not None else 'no schedule passed its check on every rank'
            # said ONCE, loudly: a training job would otherwise never know it is not on the native path
            logger.warning('pasero_amd.ddp: RCCL C-API bring-up failed on rank %d (%s); gradient all-reduce falls back '
                           'to torch.distributed', self.rank, self.report['error'])

    def all_reduce_mean(self, flat: torch.Tensor, schedule: Optional[int] = None) -> None:
        """flat <- mean over ranks, in place, on the current stream"""
        from .lib import dtype_code, ptr, stream_ptr
        sched = self.schedule if schedule is None else schedule
        scratch = None
        if sched == 2 and self.world > 1:
            nbytes = flat.numel() * flat.element_size()
            if self._scratch is None or self._scratch.numel() < nbytes:
                self._scratch = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
            scratch = self._scratch
        self.check(self.lib.pk_comm_all_reduce_mean(ptr(flat), flat.numel(), dtype_code(flat), int(sched), ptr(scratch),
                                                    stream_ptr()), 'pk_comm_all_reduce_mean')

    def choose_schedule(self, numel: int = 1 << 22) -> None:
        pinned = os.environ.get('PASERO_DDP_SCHEDULE')
        numel = (numel // (8 * self.world)) * 8 * self.world
        g = torch.Generator(device=self.device).manual_seed(1234 + self.rank)
        x = torch.randn(numel, generator=g, device=self.device).to(torch.bfloat16)
        ref = x.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.AVG, group=self.group)
        tol = 2.0 ** -6 * ref.float().abs().max().item()
        ok, ms = [], []
        for sched in (0, 1, 2):
            try:
                y = x.clone()
                if sched == 2 and self.world > 1:  # the one allocation a trial makes: before the ranks commit to it
                    self._scratch = torch.empty(numel * 2, dtype=torch.uint8, device=self.device)
                ready = True
            except Exception:
                ready = False
            if not self._all_agree(ready):
                ok.append(0.0)
                ms.append(float('inf'))
                continue
            try:
                self.all_reduce_mean(y, sched)
                good = bool(((y.float() - ref.float()).abs().max() <= tol).item())
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(3):
                    self.all_reduce_mean(y, sched)
                ev1.record()
                ev1.synchronize()
                ok.append(1.0 if good else 0.0)
                ms.append(ev0.elapsed_time(ev1) / 3)
            except Exception:
                ok.append(0.0)
                ms.append(float('inf'))
        stats = torch.tensor(ok + [-m if m != float('inf') else -1e9 for m in ms], dtype=torch.float64, device=self.device)
        dist.all_reduce(stats, op=dist.ReduceOp.MIN, group=self.group)  # valid on EVERY rank; slowest rank's time
        ok = stats[:3].tolist()
        ms = [-v for v in stats[3:].tolist()]
        self.report.update({'ms_per_all_reduce_of_%d_MiB' % (numel * 2 >> 20): dict(zip(self.SCHEDULES.values(), ms)),
                            'correct': dict(zip(self.SCHEDULES.values(), [bool(v) for v in ok]))})
        valid = [i for i in (0, 1, 2) if ok[i] > 0]
        if pinned is not None and int(pinned) in valid:
            self.schedule = int(pinned)
        elif valid:
            self.schedule = min(valid, key=lambda i: ms[i])
        else:
            self.schedule = None
        self.report['schedule'] = None if self.schedule is None else self.SCHEDULES[self.schedule]


class _Bucket:
    __slots__ = ('params', 'offsets', 'flat', 'pending', 'ready', 'launched', 'work', 'dtype', 'numel', 'plan', 'packed')

    def __init__(self, params: List[nn.Parameter], multiple: int = 8):
        self.params = params
        self.dtype = params[0].dtype
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 7) // 8 * 8  # keep every slice 16-byte aligned for the kernels that read the grads
        n = (n + multiple - 1) // multiple * multiple  # whole 16-byte chunks per rank shard (reduce-scatter schedules)
        self.numel = n
        self.flat = torch.zeros(n, dtype=self.dtype, device=params[0].device)
        self.pending = 0
        self.ready = [False] * len(params)
        self.launched = False
        self.work = None
        self.plan = None  # chunk list of the pack kernel (device tensors), built at the first pack
        self.packed = [False] * len(params)  # had a gradient when the bucket was packed (this backward or an earlier,
        # unsynchronised micro-batch): these get their reduced slice back

    def view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.flat[self.offsets[i]: self.offsets[i] + p.numel()].view_as(p)


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, output_device=None, broadcast_buffers: bool = False,
                 gradient_as_bucket_view: bool = True, find_unused_parameters: bool = False,
                 bucket_cap_mb: float = 16.0, process_group=None):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.find_unused_parameters = find_unused_parameters
        # a single rank has nothing to reduce; PASERO_DDP_FORCE_REDUCE=1 keeps the bucket / collective / stream path
        # live anyway (how the RCCL path is exercised on a one-GPU box)
        self._reduce_enabled = self.world_size > 1 or (dist.is_initialized()
                                                        and os.environ.get('PASERO_DDP_FORCE_REDUCE') == '1')
        self.require_backward_grad_sync = True
        self._ignore = set(getattr(module, '_ddp_params_and_buffers_to_ignore', []))
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and n not in self._ignore]
        self._params = [p for _, p in named]
        self._is_cuda = bool(self._params) and self._params[0].is_cuda
        self._comm_stream = torch.cuda.Stream() if self._is_cuda else None
        self._buckets: List[_Bucket] = []
        self._arena_layers = []  # layers whose backward writes its gradients into their bucket (native_layer.py)
        self._where = {}
        self._native: Optional[RcclComm] = None
        if (self._reduce_enabled and self._is_cuda and dist.get_backend(process_group) == 'nccl'
                and os.environ.get('PASERO_DDP_RCCL', '1') != '0'):
            self._native = RcclComm.get(process_group, self._params[0].device)
        self._build_buckets(int(bucket_cap_mb * (1 << 20)))
        self._callback_queued = False
        self.packed_copies = 0  # gradients copied into their bucket so far (diagnostic: 0 for natively written layers)
        self._next = 0  # first bucket not launched yet
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(p)) for p in self._params]
        if self.world_size > 1:
            self._broadcast_state()

    def describe(self) -> dict:
        """what a scaling run needs to explain itself (bench.py prints it into its JSON line): the transport and, on the native
        path, the all-reduce schedule in use with the time and the correctness check of EVERY schedule as `choose_schedule`
        measured them on this node; the buckets in launch order (MiB each: the first collective of a step can start once the
        first bucket is complete, the last one is not overlapped with backward) and how many of them are gradient arenas that
        a natively run layer writes in place"""
        sizes = [round(b.numel * b.flat.element_size() / (1 << 20), 3) for b in self._buckets]
        if not self._reduce_enabled:
            transport = 'none (one rank)'
        elif self._native is not None:
            transport = "RCCL C API (csrc/comm.hip) on a dedicated stream"
        else:
            transport = 'torch.distributed (%s)' % (dist.get_backend(self.process_group) if dist.is_initialized() else '-')
        return {'transport': transport, 'world_size': self.world_size,
                'schedule_trial': dict(self._native.report) if self._native is not None else None,
                'buckets': len(sizes), 'bucket_mib': sizes, 'total_mib': round(sum(sizes), 3),
                'largest_bucket_mib': max(sizes) if sizes else 0.0, 'last_bucket_mib': sizes[-1] if sizes else 0.0,
                'arena_layers': len(self._arena_layers), 'find_unused_parameters': self.find_unused_parameters}

    # ---- setup ----
    def _native_groups(self):
        """{parameter: (layer, [its parameters in gradient-arena order])} for every layer whose backward is one native call
        (pasero_amd/native_layer.py): such a layer's parameters sit together in one bucket, in the order pk_layer_bwd
        writes their gradients, so that backward can write them straight into the bucket"""
        groups = {}
        if not (self._reduce_enabled and (self._is_cuda or _ARENA_ON_ANY_DEVICE)):
            return groups
        try:
            from . import native_layer, transformer
        except Exception:  # (the package without its extension: CPU tests)
            return groups
        mine = set(self._params)
        for m in self.module.modules():
            is_dec = isinstance(m, transformer.TransformerDecoderLayer)
            if not (is_dec or isinstance(m, transformer.TransformerEncoderLayer)) or not native_layer._static_ok(m, is_dec):
                continue
            pieces = native_layer.grad_arena_params(m, is_dec)
            flat = [p for piece in pieces for p in piece]
            if any(p is None or p not in mine or p in groups or p.numel() % 8 or p.dtype != flat[0].dtype
                   or p.dtype not in (torch.bfloat16, torch.float16) for p in flat) or len(set(flat)) != len(flat):
                continue  # (a projection without bias, a frozen or shared parameter: the layer keeps the packed path)
            for p in flat:
                groups[p] = (m, pieces)
        return groups

    def _build_buckets(self, cap_bytes: int) -> None:
        groups = self._native_groups()
        # units in reverse registration order (gradients arrive roughly so); a native layer is one unit, placed where its
        # last-registered parameter falls
        units, seen = [], set()
        for p in reversed(self._params):
            if p in seen:
                continue
            if p in groups:
                layer, pieces = groups[p]
                unit = [q for piece in pieces for q in piece]
                units.append((unit, (layer, pieces)))
            else:
                unit = [p]
                units.append((unit, None))
            seen.update(unit)
        cur, cur_bytes, cur_layers = [], 0, []

        def close():
            nonlocal cur, cur_bytes, cur_layers
            if not cur:
                return
            b = _Bucket(cur, 8 * self.world_size)
            self._buckets.append(b)
            index = {p: i for i, p in enumerate(cur)}
            for layer, pieces in cur_layers:  # (whole 16-byte slices: the pieces are contiguous in the bucket)
                layer.__dict__['_pk_grad_arena'] = (b.flat,) + tuple(b.offsets[index[piece[0]]] for piece in pieces)
                layer.__dict__['_pk_arena_claimed'] = False
                self._arena_layers.append(layer)
            cur, cur_bytes, cur_layers = [], 0, []

        for unit, native in units:
            nbytes = sum(p.numel() * p.element_size() for p in unit)
            if cur and (cur_bytes + nbytes > cap_bytes or unit[0].dtype != cur[0].dtype):
                close()
            cur += unit
            cur_bytes += nbytes
            if native is not None:
                cur_layers.append(native)
        close()
        for bi, b in enumerate(self._buckets):
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)

    @torch.no_grad()
    def _broadcast_state(self, chunk_bytes: int = 64 << 20) -> None:
        """rank 0's parameters — trainable AND frozen — and buffers are the starting point on every rank, like torch
        DDP's constructor (`_sync_module_states`); entries of `_ddp_params_and_buffers_to_ignore` are left alone.  Sent
        in flat chunks of one dtype (≤ 64 MiB), so no second copy of the whole model is ever alive."""
        seen, tensors = set(), []
        for name, t in list(self.module.named_parameters()) + list(self.module.named_buffers()):
            if name in self._ignore or t.data_ptr() in seen or t.numel() == 0:
                continue
            seen.add(t.data_ptr())  # tied tensors (shared embeddings) travel once
            tensors.append(t.detach())
        group: List[torch.Tensor] = []
        size = 0

        def flush():
            if not group:
                return
            flat = torch.cat([t.reshape(-1) for t in group])
            dist.broadcast(flat, 0, group=self.process_group)
            off = 0
            for t in group:
                t.copy_(flat[off: off + t.numel()].view_as(t))
                off += t.numel()
            group.clear()

        for t in tensors:
            nbytes = t.numel() * t.element_size()
            if group and (t.dtype != group[0].dtype or t.device != group[0].device or size + nbytes > chunk_bytes):
                flush()
                size = 0
            group.append(t)
            size += nbytes
        flush()

    # ---- backward-time machinery ----
    def _reset(self) -> None:
        """state of one backward pass; also called from `forward`, so a backward that raised half-way (the Trainer's OOM
        path, training.py:411-420, retries with a dummy batch) leaves nothing behind"""
        self._callback_queued = False
        self._next = 0
        for b in self._buckets:
            b.pending = len(b.params)
            b.ready = [False] * len(b.params)
            b.packed = [False] * len(b.params)
            b.launched = False
            if b.work is not None:  # a collective of the aborted step: every rank issued it, let it finish
                b.work.wait()
                b.work = None

    def _make_hook(self, p: nn.Parameter):
        def hook(param):
            if not self._reduce_enabled or not self.require_backward_grad_sync:
                return
            if not self._callback_queued:
                self._reset()
                self._callback_queued = True
                Variable._execution_engine.queue_callback(self._finalize)
            bucket, i = self._where[p]
            if not bucket.ready[i]:
                bucket.ready[i] = True
                bucket.pending -= 1
            if not self.find_unused_parameters:
                self._launch_ready()
        return hook

    def _launch_ready(self) -> None:
        """launch, in bucket order, every complete bucket whose predecessors have all been launched"""
        while self._next < len(self._buckets) and self._buckets[self._next].pending == 0:
            self._reduce(self._buckets[self._next])
            self._next += 1

    @torch.no_grad()
    def _pack(self, b: _Bucket) -> None:
        # every parameter that HAS a gradient is packed — also one whose hook did not fire in this backward: gradients
        # accumulated under no_sync() by earlier micro-batches (update_freq > 1 with rank- or batch-dependent adapters,
        # training.py:392-408) are reduced with the rest and must come back (`_finalize` keeps what was packed)
        b.packed = [p.grad is not None for p in b.params]
        if not all(b.packed):  # parameters without a gradient on this rank contribute zeros
            for i, had in enumerate(b.packed):
                if not had:
                    b.view(i).zero_()
        # gradients a native layer's backward wrote into their slice (native_layer.py) are already where they belong
        base, es = b.flat.data_ptr(), b.flat.element_size()
        idx = [i for i, had in enumerate(b.packed) if had and b.params[i].grad.data_ptr() != base + b.offsets[i] * es]
        self.packed_copies += len(idx)
        if not idx:
            return
        views = [b.view(i) for i in idx]
        grads = [b.params[i].grad for i in idx]
        if self._is_cuda and all(g.is_contiguous() and g.dtype == b.dtype for g in grads):
            from . import functional as PF
            key = tuple(idx)
            if b.plan is None or b.plan[0] != key:
                b.plan = (key, PF.mt_copy_plan([v.numel() for v in views], b.flat.device))
            PF.mt_copy(grads, views, b.plan[1])
        else:  # CPU (gloo tests) or exotic gradient layouts
            torch._foreach_copy_(views, grads)

    @torch.no_grad()
    def _reduce(self, b: _Bucket) -> None:
        self._pack(b)
        b.launched = True
        if self._is_cuda:
            ready = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                self._all_reduce_avg(b)
        else:
            self._all_reduce_avg(b)

    def _all_reduce_avg(self, b: _Bucket) -> None:
        if self._native is not None:  # RCCL's C API on the communication stream; stream order is the completion handle
            self._native.all_reduce_mean(b.flat)
            b.work = None
            return
        backend = dist.get_backend(self.process_group)
        if backend == 'nccl':  # RCCL averages in the collective: no extra pass over the bucket
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.process_group, async_op=True)
        else:
            b.flat.div_(self.world_size)  # (gloo has no AVG; pre-scaling keeps one pass and the same sum)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)

    @torch.no_grad()
    def _finalize(self) -> None:
        """autograd engine callback at the end of backward: flush the remaining buckets in order (unused parameters),
        wait for the collectives and re-point `.grad` at the reduced bucket slices"""
        self._callback_queued = False
        self._release_arenas()
        used_anywhere = None
        if self.find_unused_parameters:
            # "used" = has a gradient to contribute: produced by this backward, or left by an unsynchronised one
            used = torch.tensor([float(r or p.grad is not None) for b in self._buckets
                                 for r, p in zip(b.ready, b.params)], dtype=torch.float32,
                                device=self._buckets[0].flat.device)
            dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.process_group)
            used_anywhere = used.bool().tolist()
        for b in self._buckets[self._next:]:
            self._reduce(b)
        self._next = len(self._buckets)
        for b in self._buckets:
            if b.work is not None:
                if self._is_cuda:
                    with torch.cuda.stream(self._comm_stream):
                        b.work.wait()
                else:
                    b.work.wait()
                b.work = None
        if self._is_cuda:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        k = 0
        for b in self._buckets:
            for i, p in enumerate(b.params):
                if used_anywhere is not None:
                    keep = used_anywhere[k]  # no rank produced a gradient: leave None, like torch DDP does
                else:
                    keep = b.ready[i] or b.packed[i]  # (every rank uses the same parameters; `packed`: gradients of
                    # earlier no_sync() micro-batches whose parameter the last micro-batch did not touch)
                p.grad = b.view(i) if keep else None
                k += 1

    # ---- the surface the Trainer uses ----
    def forward(self, *args, **kwargs):
        if self._callback_queued and torch.is_grad_enabled():
            self._reset()  # the previous backward never reached its end-of-backward callback
        self._release_arenas()
        return self.module(*args, **kwargs)

    def _release_arenas(self) -> None:
        """A native layer's backward that writes its gradients straight into the layer's bucket slice CLAIMS the slice
        (native_layer.py) so that a second application of the layer reaching its backward in the same pass — or a second
        forward pass back-propagated together with the first — gets tensors of its own.  The claim ends when the reducer
        has taken the bucket (`_finalize`) and, for a backward that never got there, with the next forward.  (Not in
        `_reset`: that runs at the first HOOK of a backward pass, after the first layer has already claimed.)"""
        for layer in self._arena_layers:
            layer.__dict__['_pk_arena_claimed'] = False

    @contextlib.contextmanager
    def no_sync(self):
        old = self.require_backward_grad_sync
        self.require_backward_grad_sync = False
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)


# ---- the training logs: one fused all-reduce instead of all_gather_object -------------------------------------------
LOG_KEYS = ('loss', 'nll_loss', 'num_tokens', 'num_lines')
_STATUS_LEVELS = (2, 3, 4)  # Status.FINISHED / INTERRUPTED / FAILED (pasero/training.py:36-40); RUNNING = 1


def reduce_logs(logs: dict, keys: Sequence[str] = LOG_KEYS, group=None, device=None) -> dict:
    """Drop-in for `utils.gather_dict(cfg, logs)` as `Trainer.train_step` / `valid_step` use it (pasero/utils.py:93-104;
    pasero/training.py:429-446,538): the per-rank log values are SUMMED over the ranks and the job status takes the
    WORST value (`Status.__iadd__`, training.py:46-53).  The reference pickles a dict on every rank and all-gathers the
    pickles through the GPU (two collectives, a host sync and a pickle round trip per step); here the values travel as
    ONE small fp64 tensor in ONE all-reduce(SUM): [the `keys`..., #ranks with status >= 2, >= 3, >= 4] — the maximum of
    the statuses is rebuilt from the counts.  A rank with fewer keys (the dummy-batch path passes `{}`,
    training.py:536-537) contributes zeros; keys outside `keys` are refused rather than dropped silently.
    Integer entries stay integers (`num_tokens`, `num_lines`: exact below 2**53)."""
    extra = [k for k in logs if k not in keys and k != 'status']
    if extra:
        raise KeyError(f'reduce_logs: keys {extra} are not in the fixed layout {tuple(keys)}; pass keys=...')
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return dict(logs)
    status = logs.get('status')
    level = int(getattr(status, 'value', status)) if status is not None else 1
    vals = [float(logs.get(k, 0)) for k in keys] + [float(level >= s) for s in _STATUS_LEVELS]
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else 'cpu'
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    out_vals = t.tolist()  # the one device -> host copy of the step's logs
    out = {}
    for k, v in zip(keys, out_vals):
        if k in logs:
            is_int = isinstance(logs[k], int) and not isinstance(logs[k], bool)
        else:
            is_int = k.startswith('num_')
        out[k] = int(round(v)) if is_int else v
    if status is not None or any(out_vals[len(keys):]):
        worst = 1 + sum(1 for c in out_vals[len(keys):] if c > 0)
        if hasattr(status, 'value'):
            status.value = max(int(status.value), worst)  # in place, like `Status.__iadd__`
            out['status'] = status
        else:
            out['status'] = worst
    return out
