"""Fused optimizer step for the data-parallel Trainer ("next" row): gradient normalisation, global-norm
clipping and Adam in two multi-tensor HIP launches, with the optimizer state laid out like the reference's so that
`optimizer_N.bin` checkpoints stay interchangeable (state per parameter: 'step', 'exp_avg', 'exp_avg_sq', fp32;
pasero/optimization.py:56-149, pasero/training.py:899-906).
"""
from typing import Iterable, Optional

import torch

from . import lib
from .lib import check, ptr, stream_ptr, dtype_code


class Adam(torch.optim.Optimizer):
    """Drop-in for `pasero.optimization.Adam` (fairseq-style AdamW with fp32 state).

    `step()` alone is the reference's Adam.step.  `fused_step(scale, max_norm)` additionally folds in the Trainer's
    `grad *= dp_size / num_tokens` (pasero/training.py:455-470) and `clip_grad_norm_` (pasero/optimization.py:390-427)
    and returns the gradient norm as a device tensor (no host synchronisation)."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, optimizer_states_as_fp32: bool = True, **kwargs):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plan = {}
        self._partial = None
        self._keepalive = None

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, dtype=torch.float32)
            st['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32)
        return st

    def load_state_dict(self, state_dict) -> None:
        """torch.optim.Optimizer.load_state_dict casts floating-point state to the dtype of its parameter: with bf16 /
        fp16 parameters the fp32 moments would come back in 16 bits (and half as large as the kernels assume).  The
        moments are taken from the checkpoint as they are, in fp32, on the parameter's device — the reference guards
        its optimizer against the same cast (pasero/optimization.py:151-164)."""
        super().load_state_dict(state_dict)
        saved = state_dict['state']
        old_ids = [i for g in state_dict['param_groups'] for i in g['params']]
        params = [p for g in self.param_groups for p in g['params']]
        for old_id, p in zip(old_ids, params):
            src = saved.get(old_id)
            if not src:
                continue
            st = self.state[p]
            for k in ('exp_avg', 'exp_avg_sq'):
                st[k] = src[k].detach().to(device=p.device, dtype=torch.float32).contiguous().clone()
            st['step'] = int(src['step'])

    def _chunks(self, key, params):
        plan = self._plan.get(key)
        if plan is None:
            L = lib.load()
            chunk = L.pk_mt_chunk_size()
            ct, cs = [], []
            for i, p in enumerate(params):
                for start in range(0, p.numel(), chunk):
                    ct.append(i)
                    cs.append(start)
            dev = params[0].device
            plan = (torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(cs, dtype=torch.int64, device=dev),
                    [p.numel() for p in params])
            self._plan[key] = plan
        return plan

    def _check(self, p, st):
        if not p.is_cuda:
            raise RuntimeError('pasero_amd.optim.Adam needs CUDA/HIP parameters (no CPU fallback)')
        if not (p.is_contiguous() and p.grad.is_contiguous()):
            raise RuntimeError('pasero_amd.optim.Adam needs contiguous parameters and gradients')
        for k in ('exp_avg', 'exp_avg_sq'):  # the kernels index the moments as fp32 arrays of p.numel() elements
            m = st[k]
            if not (m.dtype == torch.float32 and m.numel() == p.numel() and m.device == p.device and m.is_contiguous()):
                raise RuntimeError(f'pasero_amd.optim.Adam: state {k} must be a contiguous fp32 tensor of the '
                                   f"parameter's size on its device (got {m.dtype}, {tuple(m.shape)}, {m.device})")

    @torch.no_grad()
    def fused_step(self, scale: float = 1.0, max_norm: float = 0.0) -> Optional[torch.Tensor]:
        """One optimizer step over every parameter that has a gradient.  Like the reference: ONE gradient norm over all
        of them (`clip_grad_norm_`, optimization.py:390-427) whatever their group or dtype, parameters without a
        gradient are skipped and keep their own `state['step']` (optimization.py:72-76,120), and each parameter's bias
        correction uses its own step count."""
        import numpy as np
        L = lib.load()
        tables = []  # one per (param_group, dtype): the kernels are typed
        for gi, group in enumerate(self.param_groups):
            by_dtype = {}
            for p in group['params']:
                if p.grad is not None:
                    by_dtype.setdefault(p.dtype, []).append(p)
            for dt, params in by_dtype.items():
                states = [self._state(p) for p in params]
                for p, st in zip(params, states):
                    self._check(p, st)
                tables.append((group, params, states, self._chunks((gi, dt, tuple(id(p) for p in params)), params)))
        if not tables:
            return None
        dev = tables[0][1][0].device
        n_all = sum(t[3][0].numel() for t in tables)
        if self._partial is None or self._partial.numel() < n_all or self._partial.device != dev:
            self._partial = torch.empty(max(n_all, 1), dtype=torch.float32, device=dev)
        gnorm = torch.empty(1, dtype=torch.float32, device=dev)
        # ONE host->device upload for all tables: per table [p | g | m | v | numel] int64 then [bc1 | bc2_sqrt] fp32
        blobs, offs = [], []
        off = 0
        for group, params, states, (ct, cs, numel) in tables:
            b1, b2 = group['betas']
            for st in states:
                st['step'] += 1
            n = len(params)
            ptrs = np.array([p.data_ptr() for p in params] + [p.grad.data_ptr() for p in params]
                            + [st['exp_avg'].data_ptr() for st in states]
                            + [st['exp_avg_sq'].data_ptr() for st in states] + numel, dtype=np.int64)
            steps = np.array([st['step'] for st in states], dtype=np.float64)
            bc = np.concatenate([1.0 - b1 ** steps, np.sqrt(1.0 - b2 ** steps)]).astype(np.float32)
            blob = np.concatenate([ptrs.view(np.uint8), bc.view(np.uint8)])
            blob = np.concatenate([blob, np.zeros((-blob.size) % 16, np.uint8)])
            offs.append((off, off + 5 * n * 8))
            off += blob.size
            blobs.append(blob)
        table = torch.from_numpy(np.concatenate(blobs)).to(dev, non_blocking=True)
        base = table.data_ptr()
        done = 0
        for i, (group, params, states, (ct, cs, numel)) in enumerate(tables):
            last = i == len(tables) - 1
            check(L.pk_mt_sqnorm(base + offs[i][0], len(params), ptr(ct), ptr(cs), ct.numel(), float(scale),
                                 self._partial.data_ptr() + 4 * done, ptr(self._partial), n_all,
                                 ptr(gnorm) if last else None, dtype_code(params[0]), stream_ptr()), 'pk_mt_sqnorm')
            done += ct.numel()
        for i, (group, params, states, (ct, cs, numel)) in enumerate(tables):
            b1, b2 = group['betas']
            check(L.pk_mt_adam(base + offs[i][0], len(params), ptr(ct), ptr(cs), ct.numel(), ptr(gnorm), float(scale),
                               float(max_norm), float(group['lr']), float(b1), float(b2), float(group['eps']),
                               float(group['weight_decay']), 0, base + offs[i][1], dtype_code(params[0]),
                               stream_ptr()), 'pk_mt_adam')
            # the kernel wrote the parameters through raw pointers: tell autograd, as an in-place op would (anything keyed
            # on `_version` — Embedding.effective_weight's merged table — must see the step)
            torch._C._increment_version(params)
        self._keepalive = table  # the launches read it asynchronously
        return gnorm

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.fused_step(1.0, 0.0)
        return loss
