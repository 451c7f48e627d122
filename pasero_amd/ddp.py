"""Data-parallel gradient reduction for one-process-per-GPU training: a drop-in for the
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[..], broadcast_buffers=False,
gradient_as_bucket_view=True)` wrap of the reference Trainer (pasero/training.py:243-250), with the same surface the
Trainer touches (`.module`, `__call__`, `.no_sync()`, `.parameters()`, `.train()/.eval()`, `.state_dict()`).

Design for 8 x MI355X over xGMI (RCCL):
  * parameters are grouped, in REVERSE registration order (≈ the order autograd produces their gradients: decoder top
    -> encoder bottom -> shared embedding last), into flat buckets of `bucket_cap_mb` (16 MiB: the last, non-overlapped all-reduce of a step stays short);
  * a post-accumulate hook per parameter marks it ready; when a bucket is complete its gradients are packed into the
    flat buffer with one multi-tensor copy and ONE all-reduce (average) is launched on a dedicated communication stream
    that waits on the compute stream's event — the collective overlaps the rest of backward;
  * at the end of backward (autograd engine callback) the compute stream waits for the communication stream and every
    `param.grad` is re-pointed at its slice of the reduced bucket (no copy back);
  * `no_sync()` skips the reduction for gradient accumulation (training.py:392-408): only the last micro-batch reduces.
Semantics follow torch DDP: gradients are AVERAGED over ranks, so `Trainer.train_step`'s `grad *= dp_size/num_tokens`
normalisation (training.py:455-477) stays unchanged.
Works with any torch.distributed backend ('nccl' = RCCL on ROCm; 'gloo' on CPU for the tests).
"""
import contextlib
import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Variable


class _Bucket:
    __slots__ = ('params', 'offsets', 'flat', 'pending', 'work', 'dtype', 'numel')

    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.dtype = params[0].dtype
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 7) // 8 * 8  # keep every slice 16-byte aligned for the kernels that read the grads
        self.numel = n
        self.flat = torch.zeros(n, dtype=self.dtype, device=params[0].device)
        self.pending = 0
        self.work = None

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
        ignore = set(getattr(module, '_ddp_params_and_buffers_to_ignore', []))
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and n not in ignore]
        self._params = [p for _, p in named]
        self._is_cuda = bool(self._params) and self._params[0].is_cuda
        self._comm_stream = torch.cuda.Stream() if self._is_cuda else None
        self._buckets: List[_Bucket] = []
        self._where = {}
        self._build_buckets(int(bucket_cap_mb * (1 << 20)))
        self._callback_queued = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(p)) for p in self._params]
        if self.world_size > 1:
            self._broadcast_parameters()

    # ---- setup ----
    def _build_buckets(self, cap_bytes: int) -> None:
        cur, cur_bytes = [], 0
        for p in reversed(self._params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > cap_bytes or p.dtype != cur[0].dtype):
                self._buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._buckets.append(_Bucket(cur))
        for b in self._buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)

    @torch.no_grad()
    def _broadcast_parameters(self) -> None:
        """rank 0's parameters (and buffers) are the starting point on every rank, like torch DDP's constructor"""
        for b in self._buckets:
            flat = torch.cat([p.detach().reshape(-1) for p in b.params])
            dist.broadcast(flat, 0, group=self.process_group)
            off = 0
            for p in b.params:
                p.copy_(flat[off: off + p.numel()].view_as(p))
                off += p.numel()

    # ---- backward-time machinery ----
    def _make_hook(self, p: nn.Parameter):
        def hook(param):
            if not self._reduce_enabled or not self.require_backward_grad_sync:
                return
            if not self._callback_queued:
                self._callback_queued = True
                for b in self._buckets:
                    b.pending = len(b.params)
                Variable._execution_engine.queue_callback(self._finalize)
            bucket, _ = self._where[p]
            bucket.pending -= 1
            if bucket.pending == 0:
                self._reduce(bucket)
        return hook

    @torch.no_grad()
    def _reduce(self, b: _Bucket) -> None:
        views = [b.view(i) for i in range(len(b.params))]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params]
        torch._foreach_copy_(views, grads)  # one multi-tensor pack into the flat bucket
        if self._is_cuda:
            ready = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                self._all_reduce_avg(b)
        else:
            self._all_reduce_avg(b)

    def _all_reduce_avg(self, b: _Bucket) -> None:
        backend = dist.get_backend(self.process_group)
        if backend == 'nccl':  # RCCL averages in the collective: no extra pass over the bucket
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.process_group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)

    @torch.no_grad()
    def _finalize(self) -> None:
        """autograd engine callback at the end of backward: flush incomplete buckets (unused parameters), wait for the
        collectives and re-point `.grad` at the reduced bucket slices"""
        self._callback_queued = False
        for b in self._buckets:
            if b.pending > 0:  # some parameters received no gradient this step
                b.pending = 0
                self._reduce(b)
        backend = dist.get_backend(self.process_group)
        for b in self._buckets:
            if b.work is not None:
                if self._is_cuda:
                    with torch.cuda.stream(self._comm_stream):
                        b.work.wait()
                else:
                    b.work.wait()
                b.work = None
                if backend != 'nccl':
                    b.flat.div_(self.world_size)
        if self._is_cuda:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        for b in self._buckets:
            for i, p in enumerate(b.params):
                p.grad = b.view(i)

    # ---- the surface the Trainer uses ----
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

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
