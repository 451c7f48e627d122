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
                    torch.empty(max(len(ct), 1), dtype=torch.float32, device=dev),
                    torch.tensor([p.numel() for p in params], dtype=torch.int64))
            self._plan[key] = plan
        return plan

    @torch.no_grad()
    def fused_step(self, scale: float = 1.0, max_norm: float = 0.0) -> Optional[torch.Tensor]:
        L = lib.load()
        gnorm_out = None
        for gi, group in enumerate(self.param_groups):
            by_dtype = {}
            for p in group['params']:
                if p.grad is not None:
                    by_dtype.setdefault(p.dtype, []).append(p)
            for dt, params in by_dtype.items():
                for p in params:
                    if not p.is_cuda:
                        raise RuntimeError('pasero_amd.optim.Adam needs CUDA/HIP parameters (no CPU fallback)')
                    if not (p.is_contiguous() and p.grad.is_contiguous()):
                        raise RuntimeError('pasero_amd.optim.Adam needs contiguous parameters and gradients')
                states = [self._state(p) for p in params]
                for p, st in zip(params, states):  # the kernels index the moments as fp32 arrays of p.numel() elements
                    for k in ('exp_avg', 'exp_avg_sq'):
                        m = st[k]
                        if not (m.dtype == torch.float32 and m.numel() == p.numel() and m.device == p.device
                                and m.is_contiguous()):
                            raise RuntimeError(f'pasero_amd.optim.Adam: state {k} must be a contiguous fp32 tensor of the '
                                               f"parameter's size on its device (got {m.dtype}, {tuple(m.shape)}, {m.device})")
                ct, cs, partial, numel = self._chunks((gi, dt, tuple(id(p) for p in params)), params)
                table = torch.tensor([p.data_ptr() for p in params] + [p.grad.data_ptr() for p in params]
                                     + [s['exp_avg'].data_ptr() for s in states]
                                     + [s['exp_avg_sq'].data_ptr() for s in states] + numel.tolist(),
                                     dtype=torch.int64).to(params[0].device, non_blocking=True)
                n = len(params)
                gnorm = torch.empty(1, dtype=torch.float32, device=params[0].device)
                code = dtype_code(params[0])
                check(L.pk_mt_sqnorm(ptr(table), n, ptr(ct), ptr(cs), ct.numel(), float(scale), ptr(partial),
                                     ptr(gnorm), code, stream_ptr()), 'pk_mt_sqnorm')
                for s in states:
                    s['step'] += 1
                b1, b2 = group['betas']
                check(L.pk_mt_adam(ptr(table), n, ptr(ct), ptr(cs), ct.numel(), ptr(gnorm), float(scale),
                                   float(max_norm), float(group['lr']), float(b1), float(b2), float(group['eps']),
                                   float(group['weight_decay']), int(states[0]['step']), code, stream_ptr()),
                      'pk_mt_adam')
                gnorm_out = gnorm if gnorm_out is None else torch.sqrt(gnorm_out ** 2 + gnorm ** 2)
        return gnorm_out

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.fused_step(1.0, 0.0)
        return loss
