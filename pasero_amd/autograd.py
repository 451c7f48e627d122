"""torch.autograd.Function wrappers: each one is a forward + a hand-written backward made of C-ABI kernel calls
(pasero_amd.functional).  PyTorch's autograd engine is only the glue that orders them; no aten compute op runs
on the hot path.
"""
import math
import os
import threading
import weakref
from typing import Optional

import torch
from torch import Tensor

from . import functional as F
from . import rng


class _Outer(threading.local):
    grad = True


_outer = _Outer()


class Function(torch.autograd.Function):
    """torch.autograd.Function that knows whether a backward pass can follow.  `ctx.needs_input_grad` only mirrors the
    inputs' `requires_grad` flags — it stays True for parameters under `torch.no_grad()` — and grad mode is always off
    inside `forward`, so `apply` records the caller's grad mode and `wants_grad(ctx)` combines the two.  Forward passes
    use it to skip everything that only a backward pass needs (saved pre-activations, z = x + residual, dropout masks,
    and the gradient GEMMs the fused vocabulary cross-entropy runs in its forward) at inference."""

    _keep_fp32 = ()  # positions of tensor arguments that stay fp32 under autocast (e.g. rotary tables)

    @classmethod
    def apply(cls, *args, **kwargs):
        prev, _outer.grad = _outer.grad, torch.is_grad_enabled()
        try:
            if torch.is_autocast_enabled('cuda'):
                # `--amp` (pasero/training.py:27-31,379): fp32 parameters, 16-bit compute.  Same contract as
                # torch.amp.custom_fwd(cast_inputs=...): fp32 CUDA tensors are cast on the way in (the casts stay in
                # the autograd graph, so parameter gradients come back in fp32) and the kernels run in that type.
                dt = torch.get_autocast_dtype('cuda')
                args = tuple(a.to(dt) if (torch.is_tensor(a) and a.is_cuda and a.dtype == torch.float32
                                          and i not in cls._keep_fp32) else a for i, a in enumerate(args))
                with torch.autocast('cuda', enabled=False):
                    return super().apply(*args, **kwargs)
            return super().apply(*args, **kwargs)
        finally:
            _outer.grad = prev


def wants_grad(ctx) -> tuple:
    """per input: a gradient may be asked for later (requires_grad AND the caller's grad mode is on)"""
    return tuple(bool(f) and _outer.grad for f in ctx.needs_input_grad)


def _2d(x: Tensor) -> Tensor:
    return x.reshape(-1, x.size(-1))


def _contig(x: Tensor) -> Tensor:
    return x if x.is_contiguous() else x.contiguous()


def weight_grad(dy2d: Tensor, x2d: Tensor, want_bias: bool = False):
    """dW[N,K] = dyᵀ x : both operands in "col" form (contraction over the rows), split-K when the output is small.
    With `want_bias` the bias gradient colsum(dy) comes out of the same GEMM (it already streams dy): -> (dW, db)"""
    M = dy2d.size(0)
    N, K = dy2d.size(1), x2d.size(1)
    db = torch.empty(N, dtype=dy2d.dtype, device=dy2d.device) if want_bias else None
    dw = F.gemm(dy2d, x2d, a_col=True, b_col=True, splitk=F.choose_splitk(N, K, M), asum_out=db)
    return (dw, db) if want_bias else dw


class ResidualLink:
    """Couples the two ends of a residual sub-block  y = Norm(x + f(x))  in backward: the block-end function parks the
    gradient of the residual branch here instead of returning it to autograd, and the FIRST op of f (the GEMM that
    consumed x) adds it in the epilogue of its dX GEMM (mode 1).  That removes the engine's separate `dx_f + dres`
    accumulation kernel (a 3-tensor elementwise pass per sub-block).  Valid because the block-end backward always runs
    before f's first op (data dependency) and x has no other consumer inside the block."""
    __slots__ = ('dres', 'attached')

    def __init__(self):
        self.dres = None
        self.attached = False  # set by the forward of the GEMM function that will add `dres` in its dX epilogue

    def attach(self):
        self.attached = True
        return self

    def take(self):
        d, self.dres = self.dres, None
        return d


def _dx_gemm(dy2d: Tensor, weight: Tensor, link, **kw) -> Tensor:
    """dX = dY · W (+ parked residual gradient)"""
    dres = link.take() if link is not None else None
    kw.setdefault('splitk', F.choose_splitk(dy2d.size(0), weight.size(1), weight.size(0)))
    if dres is not None:
        return F.gemm(dy2d, weight, b_col=True, aux=_2d(dres), mode=1, **kw)
    return F.gemm(dy2d, weight, b_col=True, **kw)


def _wgrad(dy2d: Tensor, x2d: Tensor, want_w: bool, want_b: bool):
    if want_w:
        r = weight_grad(dy2d, x2d, want_b)
        return r if want_b else (r, None)
    return None, (F.colsum(dy2d) if want_b else None)


class WGradGroup:
    """The weight-gradient GEMMs of ONE layer, collected during its backward and launched together.

    Each dW = dYᵀ·X of a layer has a small output (d x d ... f x d) and a contraction over all B·T rows: launched one by
    one, every GEMM needs 16-64 K-slices of its own to fill the chip, that many fp32 partial outputs and a reduction
    launch.  Launched together (`F.wgrad_group` -> pk_gemm_wgrad_group) they fill the 256 CUs with 4-5 slices each.
    Mechanics: `WGradSinkFn` is the first autograd node of the layer and takes the layer's weights as extra inputs; the
    GEMM functions of the layer (LinearFn, FFNFn, PackedLinearFn) hand their (dY, X) pairs to the group instead of
    returning dW, and the sink — whose backward runs last in the layer, when the gradient of the layer input arrives —
    launches the group and returns the gradients to autograd, so `.grad` accumulation and the DDP hooks see nothing new.
    Every deferred op was created after the sink (higher sequence number: the engine runs it first when both are ready)
    and is ready no later than a node on the sink's own dependency chain.  Once the sink has run the group is closed: an
    op that comes later (a second backward over a retained graph) computes and returns its gradient itself (`_defer`);
    a direct `add` to a closed group would lose a gradient and raises."""
    __slots__ = ('slots', 'entries', 'closed', 'params')

    def __init__(self):
        self.slots = {}
        self.entries = {}  # (producing node, weight slots) -> entry: a node that adds twice (a partial backward through
        # its part of a retained graph that never reached the sink, then the full one) REPLACES its entry instead of being
        # summed twice; two nodes that share a weight both count.  (Entries keep dY and X of every Linear of the layer
        # alive until the sink runs: the memory price of the single launch.)
        self.closed = False
        self.params = ()

    def bind(self, params):
        # by object identity (the group keeps the parameters alive): a parameter's storage may move between the layer
        # entry and its use — MultiheadAttention packs q|k|v into one arena at its first call — the object does not;
        # the 16-bit copies autocast makes of fp32 parameters are other objects and stay on the one-by-one path
        self.params = tuple(params)
        self.slots = {id(p): i for i, p in enumerate(self.params)}

    def slot(self, p):
        """position of parameter `p` among the sink's inputs (None: not taken — frozen, cast by autocast, LoRA...)"""
        return None if p is None else self.slots.get(id(p))

    def add(self, dy2: Tensor, x2: Tensor, w_targets, b_targets, owner=None) -> None:
        """dW = dy2ᵀ·x2; `w_targets` / `b_targets`: [(slot, row0, row1)] — which rows of dW / db are whose gradient"""
        if self.closed:
            raise RuntimeError('pasero_amd: a weight gradient was handed to a layer\'s WGradGroup after the group had '
                               'been launched; it would be lost (set PASERO_NO_WGRAD_GROUP=1 and report the model)')
        want_b = bool(b_targets)
        key = (owner if owner is not None else ('anon', len(self.entries)), tuple(t[0] for t in w_targets))
        if F.wgrad_group_eligible(dy2, x2):
            self.entries[key] = (dy2, x2, want_b, w_targets, b_targets, None)
        else:  # small / unaligned / fp32 problems: the ordinary GEMM, now
            r = weight_grad(dy2, x2, want_b)
            self.entries[key] = (None, None, want_b, w_targets, b_targets, r if want_b else (r, None))

    def flush(self):
        self.closed = True
        entries, self.entries = list(self.entries.values()), {}
        todo = [e for e in entries if e[5] is None]
        done = F.wgrad_group([(e[0], e[1], e[2]) for e in todo]) if todo else []
        it = iter(done)
        grads = [None] * len(self.params)
        for dy2, x2, want_b, w_t, b_t, res in entries:
            dw, db = res if res is not None else next(it)
            for tensor, targets in ((dw, w_t), (db, b_t)):
                for slot, r0, r1 in targets or ():
                    g = tensor if (r0 == 0 and r1 == tensor.size(0)) else tensor[r0:r1]
                    grads[slot] = g if grads[slot] is None else grads[slot] + g
        return grads


class WGradSinkFn(Function):
    """identity on x at the entry of a layer; its backward launches the layer's grouped weight-gradient GEMM (WGradGroup)
    and returns the gradients of `params` (the layer's Linear weights and biases)"""

    @staticmethod
    def forward(ctx, x, group, *params):
        group.bind(params)
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dx):
        grads = ctx.group.flush()
        return (dx, None, *[g if need else None for g, need in zip(grads, ctx.needs_input_grad[2:])])


def _defer(group, dy2, x2, weight, bias, want_w: bool, want_b: bool, owner=None) -> bool:
    """hand dW (and db) of one nn.Linear to the layer's group; False: the caller computes them itself"""
    if group is None or not want_w or group.closed:
        # (closed: a second backward over a retained graph, or an op the engine ran after the layer's sink — the op
        # computes its gradient itself and returns it to autograd the ordinary way: nothing is lost, only not grouped)
        return False
    ws = group.slot(weight)
    bs = group.slot(bias) if want_b else None
    if ws is None or (want_b and bs is None):
        return False
    N = weight.size(0)
    group.add(dy2, x2, [(ws, 0, N)], [(bs, 0, N)] if want_b else None, owner)
    return True


class LinearFn(Function):
    """y = act(x Wᵀ + b)   (pasero/models/modules.py:92-96 + the activation that follows fc1)"""

    @staticmethod
    def forward(ctx, x, weight, bias, act: str = 'none', link=None, group=None):
        ctx.link = link.attach() if (link is not None and ctx.needs_input_grad[0]) else None
        ctx.group, ctx.bias = group, (bias if group is not None else None)
        x2 = _2d(_contig(x))
        need_pre = act not in ('none', 'relu') and any(wants_grad(ctx))
        pre = torch.empty(x2.size(0), weight.size(0), dtype=x.dtype, device=x.device) if need_pre else None
        y = F.gemm(x2, weight, bias=bias, act=act, preact=pre,
                   splitk=1 if pre is not None else F.fwd_split(x2.size(0), weight.size(0), weight.size(1), x.dtype))
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x2, weight, pre if need_pre else (y if act == 'relu' else None))
        return y.view(*x.shape[:-1], weight.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, weight, aux = ctx.saved_tensors
        dy2 = _2d(_contig(dy))
        if ctx.act != 'none':
            dy2 = F.act_bwd(dy2, aux, ctx.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dx_gemm(dy2, weight, ctx.link).view(*dy.shape[:-1], weight.size(1))
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if _defer(ctx.group, dy2, x2, weight, ctx.bias, ctx.needs_input_grad[1], want_b, owner=id(ctx)):
            pass  # (the layer's WGradSinkFn returns both gradients)
        elif ctx.needs_input_grad[1]:
            dw = weight_grad(dy2, x2, want_b)
            if want_b:
                dw, db = dw
        elif want_b:
            db = F.colsum(dy2)
        return dx, dw, db, None, None, None


class FFNFn(Function):
    """y = fc2(act(fc1(x)))   (pasero/models/transformer.py:999-1019, 1224-1244; no fc3, no activation dropout).
    Backward fuses act'(.) into the epilogue of the dH = dY·W2 GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: str, link=None, group=None):
        ctx.link = link.attach() if (link is not None and ctx.needs_input_grad[0]) else None
        ctx.group, ctx.biases = group, ((b1, b2) if group is not None else (None, None))
        x2 = _2d(_contig(x))
        grad = any(wants_grad(ctx))
        need_pre = grad and act not in ('none', 'relu')
        pre = torch.empty(x2.size(0), w1.size(0), dtype=x.dtype, device=x.device) if need_pre else None
        bits = None
        if grad and act == 'relu' and F.relu_bits_eligible(x2, w1):  # the mask for backward as one bit per element
            h, bits = F.gemm_relu_bits(x2, w1, b1)
        else:
            h = F.gemm(x2, w1, bias=b1, act=act, preact=pre)
        y = F.gemm(h, w2, bias=b2, splitk=F.fwd_split(h.size(0), w2.size(0), w2.size(1), x.dtype))
        ctx.act = act
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.save_for_backward(x2, w1, w2, h, pre, bits)
        return y.view(*x.shape[:-1], w2.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, pre, bits = ctx.saved_tensors
        dy2 = _2d(_contig(dy))
        aux = h if pre is None else pre
        if bits is not None:
            dh = F.gemm_mask_bits(dy2, w2, bits)
        elif ctx.act == 'none':
            dh = F.gemm(dy2, w2, b_col=True)
        else:
            dh = F.gemm(dy2, w2, b_col=True, act=ctx.act, aux=aux, mode=2)
        ng = ctx.needs_input_grad
        dw1 = db1 = dw2 = db2 = None
        if not _defer(ctx.group, dy2, h, w2, ctx.biases[1], ng[3], ctx.has_b2 and ng[4], owner=id(ctx)):
            dw2, db2 = _wgrad(dy2, h, ng[3], ctx.has_b2 and ng[4])
        dx = _dx_gemm(dh, w1, ctx.link).view(*dy.shape[:-1], w1.size(1)) if ng[0] else None
        if not _defer(ctx.group, dh, x2, w1, ctx.biases[0], ng[1], ctx.has_b1 and ng[2], owner=id(ctx)):
            dw1, db1 = _wgrad(dh, x2, ng[1], ctx.has_b1 and ng[2])
        return dx, dw1, db1, dw2, db2, None, None, None


class GatedFFNFn(Function):
    """y = fc2( act(fc1 x) * fc3 x )   (SwiGLU / GEGLU, pasero/models/transformer.py:1011-1018): the gate product is the
    epilogue of the fc1 GEMM (mode 3), its backward one elementwise kernel"""

    @staticmethod
    def forward(ctx, x, w1, b1, w3, b3, w2, b2, act: str, link=None):
        ctx.link = link.attach() if (link is not None and ctx.needs_input_grad[0]) else None
        x2 = _2d(_contig(x))
        u = F.gemm(x2, w3, bias=b3)
        z = torch.empty(x2.size(0), w1.size(0), dtype=x.dtype, device=x.device)
        h = F.gemm(x2, w1, bias=b1, act=act, aux=u, mode=3, preact=z)
        y = F.gemm(h, w2, bias=b2)
        ctx.act = act
        ctx.has_b = (b1 is not None, b3 is not None, b2 is not None)
        ctx.save_for_backward(x2, w1, w3, w2, z, u, h)
        return y.view(*x.shape[:-1], w2.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w3, w2, z, u, h = ctx.saved_tensors
        dy2 = _2d(_contig(dy))
        dh = F.gemm(dy2, w2, b_col=True)
        dz, du = F.gated_act_bwd(dh, z, u, ctx.act)
        dw2, db2 = _wgrad(dy2, h, ctx.needs_input_grad[5], ctx.has_b[2] and ctx.needs_input_grad[6])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx_gemm(dz, w1, ctx.link)
            F.gemm(du, w3, b_col=True, aux=dx, mode=1, out=dx)
            dx = dx.view(*dy.shape[:-1], w1.size(1))
        dw1, db1 = _wgrad(dz, x2, ctx.needs_input_grad[1], ctx.has_b[0] and ctx.needs_input_grad[2])
        dw3, db3 = _wgrad(du, x2, ctx.needs_input_grad[3], ctx.has_b[1] and ctx.needs_input_grad[4])
        return dx, dw1, db1, dw3, db3, dw2, db2, None, None


class PackedLinearFn(Function):
    """y = x · W_flatᵀ + b_flat where W_flat [n*D, K] is the flat arena holding `n` projection weights back to back
    (q|k|v or k|v, pasero/models/modules.py:610-615).  ONE GEMM reads x once; the gradient of each nn.Parameter is the
    matching slice of one flat gradient GEMM.  `params` = n weights followed by n biases (or None)."""

    @staticmethod
    def forward(ctx, x, w_flat, b_flat, n: int, link, *params):
        group = None
        if params and isinstance(params[-1], WGradGroup):
            group, params = params[-1], params[:-1]
        # `link` may also be the decoder pass's encoder-gradient chain (native_layer.DencChain, round 5: the per-op path's
        # cross-attention k|v projections join it): every decoder layer reads the same encoder output, and their dX GEMMs
        # add into ONE tensor (epilogue mode 1, in place) instead of autograd adding 23 (rows, d) tensors of the IWSLT recipe
        ctx.chain = None
        if link is not None and hasattr(link, 'buf'):
            chain, link = link, None
            if (ctx.needs_input_grad[0] and any(wants_grad(ctx)[:1])
                    and chain.key == (x.data_ptr(), tuple(x.shape), x.dtype) and x.is_contiguous()):
                ctx.chain = chain
                chain.n += 1
                chain.left = chain.n
        ctx.link = link.attach() if (link is not None and ctx.needs_input_grad[0]) else None
        ctx.group, ctx.params = group, (params if group is not None else None)
        x2 = _2d(_contig(x))
        y = F.gemm(x2, w_flat, bias=b_flat)
        ctx.n = n
        ctx.has_bias = [p is not None for p in params[n:]]
        ctx.save_for_backward(x2, w_flat)
        return y.view(*x.shape[:-1], w_flat.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, w_flat = ctx.saved_tensors
        n = ctx.n
        dy2 = _2d(_contig(dy))
        dx = None
        chain = ctx.chain
        if ctx.needs_input_grad[0] and chain is not None and chain.n > 1:
            sk = F.choose_splitk(dy2.size(0), w_flat.size(1), w_flat.size(0))
            if chain.left == chain.n:                     # the first of the chain in this backward pass writes the tally,
                chain.buf = torch.empty(*dy.shape[:-1], w_flat.size(1), dtype=dy2.dtype, device=dy2.device)
                torch.autograd.Variable._execution_engine.queue_callback(chain.check)
                F.gemm(dy2, w_flat, b_col=True, out=_2d(chain.buf), splitk=sk)
            else:                                         # the others add theirs to it, in place,
                F.gemm(dy2, w_flat, b_col=True, aux=_2d(chain.buf), mode=1, out=_2d(chain.buf), splitk=sk)
            chain.left -= 1
            if chain.left == 0:                           # and the last one hands the sum to autograd
                dx, chain.buf, chain.left = chain.buf, None, chain.n
        elif ctx.needs_input_grad[0]:
            dx = _dx_gemm(dy2, w_flat, ctx.link).view(*dy.shape[:-1], w_flat.size(1))
        D = w_flat.size(0) // n
        grads = [None] * (2 * n + (1 if ctx.group is not None else 0))
        want_w = any(ctx.needs_input_grad[5:5 + n])
        want_b = any(ctx.has_bias[i] and ctx.needs_input_grad[5 + n + i] for i in range(n))
        if ctx.group is not None and want_w and not ctx.group.closed:
            # every wanted slice must be one of the sink's parameters, or the whole GEMM stays here
            g, ps = ctx.group, ctx.params
            w_t = [(g.slot(ps[i]), i * D, (i + 1) * D) for i in range(n) if ctx.needs_input_grad[5 + i]]
            b_t = [(g.slot(ps[n + i]), i * D, (i + 1) * D) for i in range(n)
                   if ctx.has_bias[i] and ctx.needs_input_grad[5 + n + i]]
            if all(t[0] is not None for t in w_t + b_t):
                g.add(dy2, x2, w_t, b_t, id(ctx))
                return (dx, None, None, None, None, *grads)
        dw, db = _wgrad(dy2, x2, want_w, want_b)
        for i in range(n):
            if want_w and ctx.needs_input_grad[5 + i]:
                grads[i] = dw[i * D:(i + 1) * D]
            if want_b and ctx.has_bias[i] and ctx.needs_input_grad[5 + n + i]:
                grads[n + i] = db[i * D:(i + 1) * D]
        return (dx, None, None, None, None, *grads)


class AttentionFn(Function):
    """softmax(q kᵀ * scale + masks) v on projection outputs, without reshapes or transposes:
       self-attention:  a = packed (B,T,3D) [q|k|v], b = c = None      -> backward returns one (B,T,3D) tensor
       cross-attention: a = q (B,T,D), b = packed (B,S,2D) [k|v], c = None
       general:         a = q, b = k, c = v
    (pasero/models/modules.py:654-720)"""

    @staticmethod
    def _split(a, b, c):
        if b is None:
            D = a.size(-1) // 3
            return D, a[..., :D], a[..., D:2 * D], a[..., 2 * D:]
        if c is None:
            D = a.size(-1)
            return D, a, b[..., :D], b[..., D:]
        return a.size(-1), a, b, c

    @staticmethod
    def forward(ctx, a, b, c, key_pad, num_heads: int, causal: bool, scale: float, p: float = 0.0, rope=None):
        # rope = (cos_t, sin_t, q_pos0, k_pos0): rotary positions applied INSIDE the kernels — q and k stay the unrotated
        # projection, backward returns the gradient of that (modules.py:617-623 without a pass of its own)
        D, q, k, v = AttentionFn._split(a, b, c)
        mask = None
        if p > 0:  # attention-probability dropout: the keep bits are drawn in the forward kernel and kept for backward
            seed, offset = rng.next_offset()
            o, lse, mask = F.attn_fwd(q, k, v, num_heads, key_pad, causal, scale, p, seed, offset, rope=rope)
        else:
            o, lse = F.attn_fwd(q, k, v, num_heads, key_pad, causal, scale, rope=rope)
        ctx.num_heads, ctx.causal, ctx.scale, ctx.p, ctx.rope = num_heads, causal, scale, p, rope
        ctx.save_for_backward(a, b, c, key_pad, o, lse, mask)
        return o

    @staticmethod
    def backward(ctx, d_o):
        a, b, c, key_pad, o, lse, mask = ctx.saved_tensors
        D, q, k, v = AttentionFn._split(a, b, c)
        da = torch.empty_like(a)
        db = torch.empty_like(b) if b is not None else None
        dc = torch.empty_like(c) if c is not None else None
        _, dq, dk, dv = AttentionFn._split(da, db, dc)
        F.attn_bwd(q, k, v, o, _contig(d_o), lse, ctx.num_heads, key_pad, ctx.causal, ctx.scale, dq=dq, dk=dk, dv=dv,
                   drop_p=ctx.p, drop_mask=mask, rope=ctx.rope)
        return da, db, dc, None, None, None, None, None, None


class ResidualLayerNormFn(Function):
    """y = LayerNorm(residual + dropout(x)) in one pass (post-norm blocks, pasero/models/transformer.py:1043-1048);
    with residual=None and p=0 it is a plain LayerNorm; `rms` makes the normalisation an RMSNorm (modules.py:192-202)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps: float, p: float, link=None, rms: bool = False):
        # (a link no GEMM picked up — a sub-block path that does not forward it — must not swallow the gradient)
        ctx.link = link if (link is not None and link.attached and residual is not None
                            and residual.requires_grad) else None
        x = _contig(x)
        residual = _contig(residual) if residual is not None else None
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        fused = residual is not None or p > 0
        y, z, mean, rstd = F.residual_ln_fwd(x, residual, gamma, beta, eps, p, seed, offset,
                                             want_z=fused and any(wants_grad(ctx)), rms=rms)
        ctx.p, ctx.seed, ctx.offset = p, seed, offset
        ctx.has_res, ctx.has_beta = residual is not None, beta is not None
        ctx.save_for_backward(z if fused else x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, gamma, mean, rstd = ctx.saved_tensors
        dy = _contig(dy)
        want_pg = ctx.needs_input_grad[2] or (ctx.has_beta and ctx.needs_input_grad[3])
        need_dx = ctx.needs_input_grad[0]
        dres, dx, dgamma, dbeta = F.residual_ln_bwd(
            dy, None, z, gamma, mean, rstd, want_dres=True, want_dx=need_dx and ctx.p > 0,
            want_param_grads=want_pg, has_beta=ctx.has_beta, drop_p=ctx.p, seed=ctx.seed, offset=ctx.offset)
        if ctx.p == 0:
            dx = dres
        if ctx.link is not None:  # the residual-branch gradient rides on the sub-block's first dX GEMM instead
            ctx.link.dres = dres
            dres = None
        return (dx if need_dx else None, dres if ctx.has_res else None, dgamma,
                dbeta if ctx.has_beta else None, None, None, None, None)


class DropLink:
    """Hand-over between `z = residual + dropout(x)` (ResidualDropoutFn) and the pre-norm fork that consumes z
    (LayerNormForkFn): the fork's backward already computes dz — the LayerNorm gradient plus the residual branch's — in one
    kernel, and that kernel can write the MASKED copy the dropout's backward needs in the same pass (pk_residual_ln_bwd's
    dx output; what the native layer call does).  The producer leaves (p, seed, offset) here in forward; the consumer's
    backward `offer`s the masked gradient together with the dz it belongs to; the producer's backward `take`s it only if
    the dz it receives is that very tensor, untouched.  Another consumer of z makes autograd hand over a SUM — and its input
    buffer adds in place when it holds the last reference, so the sum can sit at dz's address: the link keeps
    a reference to dz until the hand-over (the engine then adds out of place, into a new tensor) and also compares the
    tensor's version counter; in either case the producer draws the mask itself."""
    __slots__ = ('p', 'seed', 'offset', 'masked', 'dz', 'dz_version')

    def __init__(self):
        self.p, self.seed, self.offset, self.masked, self.dz, self.dz_version = 0.0, 0, 0, None, None, 0

    def offer(self, masked, dz) -> None:
        self.masked, self.dz, self.dz_version = masked, dz, dz._version

    def take(self, dz):
        """the masked gradient offered for exactly this `dz`, or None; the offer is consumed either way"""
        masked, mine, version = self.masked, self.dz, self.dz_version
        self.masked = self.dz = None
        if (masked is None or mine is None or dz.data_ptr() != mine.data_ptr() or dz._version != version
                or dz.shape != masked.shape or dz.dtype != mine.dtype):
            return None
        return masked


class LayerNormForkFn(Function):
    """(LayerNorm(x), x) — a pre-norm sub-block reads its input twice: through the LayerNorm and, as the residual, around
    the sub-block (pasero/models/transformer.py:1070-1075: `residual = x; x = self.self_attn_prenorm(x)`).  As two
    consumers of one tensor, autograd adds their gradients in an elementwise pass of its own (141 launches per step of the
    IWSLT recipe, 2.3 % of it); as ONE node the residual branch's gradient enters the LayerNorm backward kernel as
    `dz_extra` — what the native layer call does (csrc/layer.cpp `ln_in_bwd`), with one rounding instead of two.
    `drop`: the DropLink of the `residual + dropout(.)` that produced x, if any (round 5)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, drop=None):
        xc = _contig(x)
        y, _, mean, rstd = F.residual_ln_fwd(xc, None, gamma, beta, eps, want_z=False)
        ctx.has_beta = beta is not None
        ctx.drop = drop if (drop is not None and drop.p > 0 and xc.dtype != torch.float32) else None
        ctx.save_for_backward(xc, gamma, mean, rstd)
        # an output nobody consumed arrives as None in backward, not as a tensor of zeros allocated for the occasion
        ctx.set_materialize_grads(False)
        return y, x  # (an input returned as it is: autograd hands out a view of it whose history is this node)

    @staticmethod
    def backward(ctx, dy, dres):
        x, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:  # only the residual branch was used: the identity
            return (dres if ctx.needs_input_grad[0] else None), None, None, None, None
        want_pg = ctx.needs_input_grad[1] or (ctx.has_beta and ctx.needs_input_grad[2])
        extra = _contig(dres) if dres is not None else None  # None: only the LayerNorm branch was used
        d = ctx.drop if ctx.needs_input_grad[0] else None
        dx, masked, dgamma, dbeta = F.residual_ln_bwd(
            _contig(dy), extra, x, gamma, mean, rstd, want_dres=True, want_dx=d is not None, want_param_grads=want_pg,
            has_beta=ctx.has_beta, drop_p=d.p if d else 0.0, seed=d.seed if d else 0, offset=d.offset if d else 0)
        if d is not None:
            d.offer(masked, dx)
        return (dx if ctx.needs_input_grad[0] else None, dgamma if ctx.needs_input_grad[1] else None,
                dbeta if (ctx.has_beta and ctx.needs_input_grad[2]) else None, None, None)


class BlockTail:
    """What the LAST GEMM of a post-norm sub-block needs to finish the block itself (LinearResidualLnFn): handed by the
    layer to the module that runs that GEMM (MultiheadAttention.out_proj) for one call; `done` tells the layer that the
    output it got back is already  LayerNorm(residual + dropout(.))."""
    __slots__ = ('residual', 'gamma', 'beta', 'eps', 'p', 'done')

    def __init__(self, residual, gamma, beta, eps: float, p: float):
        self.residual, self.gamma, self.beta, self.eps, self.p = residual, gamma, beta, eps, p
        self.done = False


def block_tail_eligible(rows: int, weight: Tensor, residual: Tensor, gamma: Tensor) -> bool:
    """can `LayerNorm(residual + dropout(x Wᵀ + b))` over `rows` rows run as one kernel (pk_gemm_ln_fwd)?  d = 512, whole
    K-tiles, 16-bit tensors of one type, and no autocast (its 16-bit copies of fp32 parameters take the general path)"""
    if torch.is_autocast_enabled('cuda') or not residual.is_cuda or rows == 0:
        return False
    if not (weight.dtype == residual.dtype == gamma.dtype) or residual.size(-1) != weight.size(0):
        return False
    return F.gemm_ln_eligible_shape(rows, weight)


def _ln_tail_backward(ctx, dy, z, gamma, mean, rstd):
    """LayerNorm + dropout backward of a fused block end -> (dres, dsub, dgamma, dbeta): the gradient of the residual
    branch, the (masked, scaled) gradient of the GEMM output, the parameter gradients"""
    want_pg = ctx.need_gamma or ctx.need_beta
    dres, dsub, dgamma, dbeta = F.residual_ln_bwd(
        _contig(dy), None, z, gamma, mean, rstd, want_dres=True, want_dx=ctx.p > 0, want_param_grads=want_pg,
        has_beta=ctx.has_beta, drop_p=ctx.p, seed=ctx.seed, offset=ctx.offset)
    if ctx.p == 0:
        dsub = dres
    return dres, dsub, (dgamma if ctx.need_gamma else None), (dbeta if ctx.need_beta else None)


class LinearResidualLnFn(Function):
    """y = LayerNorm(residual + dropout(x Wᵀ + b)) in ONE kernel (pk_gemm_ln_fwd): the output projection of an attention
    block together with the post-norm block end (pasero/models/modules.py:739 + transformer.py:1043-1048,1076-1086).
    Backward: the stand-alone LayerNorm backward kernel, then the dX GEMM; dW rides in the layer's grouped launch."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, gamma, beta, eps: float, p: float, link=None, group=None):
        x2 = _2d(_contig(x))
        res2 = _2d(_contig(residual))
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        grad = any(wants_grad(ctx))
        y, z, mean, rstd = F.gemm_ln_fwd(x2, weight, bias, res2, gamma, beta, eps, p, seed, offset, want_z=grad)
        # (a link no GEMM picked up — a sub-block path that does not forward it — must not swallow the gradient)
        ctx.link = link if (link is not None and link.attached and residual.requires_grad) else None
        ctx.group, ctx.bias = group, (bias if group is not None else None)
        ctx.p, ctx.seed, ctx.offset = p, seed, offset
        ctx.has_bias, ctx.has_beta = bias is not None, beta is not None
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, weight, z, gamma, mean, rstd)
        return y.view(residual.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, z, gamma, mean, rstd = ctx.saved_tensors
        ng = ctx.needs_input_grad  # x, weight, bias, residual, gamma, beta
        ctx.need_gamma, ctx.need_beta = ng[4], ctx.has_beta and ng[5]
        dres, dsub, dgamma, dbeta = _ln_tail_backward(ctx, dy, z, gamma, mean, rstd)
        dsub2 = _2d(dsub)
        dx = _dx_gemm(dsub2, weight, None).view(ctx.x_shape) if ng[0] else None
        dw = db = None
        want_b = ctx.has_bias and ng[2]
        if not _defer(ctx.group, dsub2, x2, weight, ctx.bias, ng[1], want_b, owner=id(ctx)):
            dw, db = _wgrad(dsub2, x2, ng[1], want_b)
        if ctx.link is not None:  # the residual-branch gradient rides on the sub-block's first dX GEMM
            ctx.link.dres = dres
            dres = None
        return dx, dw, db, (dres.view(dy.shape) if (dres is not None and ng[3]) else None), dgamma, dbeta, None, None, None, None


class FFNResidualLnFn(Function):
    """y = LayerNorm(x + dropout(fc2(act(fc1 x)))): a whole post-norm feed-forward sub-block (pasero/models/transformer.py:
    999-1019 + 1043-1054) as two launches — fc1 with the activation in its epilogue, fc2 with bias, dropout, the residual
    and LayerNorm in its epilogue.  Backward: LayerNorm backward, dH = (dZ·W2) ⊙ act′, dX = dH·W1 + (residual-branch
    gradient, in the epilogue); both weight gradients ride in the layer's grouped launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: str, gamma, beta, eps: float, p: float, group=None):
        x2 = _2d(_contig(x))
        grad = any(wants_grad(ctx))
        need_pre = grad and act not in ('none', 'relu')
        pre = torch.empty(x2.size(0), w1.size(0), dtype=x.dtype, device=x.device) if need_pre else None
        bits = None
        if grad and act == 'relu' and F.relu_bits_eligible(x2, w1):
            h, bits = F.gemm_relu_bits(x2, w1, b1)
        else:
            h = F.gemm(x2, w1, bias=b1, act=act, preact=pre)
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        y, z, mean, rstd = F.gemm_ln_fwd(h, w2, b2, x2, gamma, beta, eps, p, seed, offset, want_z=grad)
        ctx.group, ctx.biases = group, ((b1, b2) if group is not None else (None, None))
        ctx.act, ctx.p, ctx.seed, ctx.offset = act, p, seed, offset
        ctx.has_b1, ctx.has_b2, ctx.has_beta = b1 is not None, b2 is not None, beta is not None
        ctx.save_for_backward(x2, w1, w2, h, pre, z, gamma, mean, rstd, bits)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, pre, z, gamma, mean, rstd, bits = ctx.saved_tensors
        ng = ctx.needs_input_grad  # x, w1, b1, w2, b2, act, gamma, beta
        ctx.need_gamma, ctx.need_beta = ng[6], ctx.has_beta and ng[7]
        dres, dsub, dgamma, dbeta = _ln_tail_backward(ctx, dy, z, gamma, mean, rstd)
        dy2 = _2d(dsub)
        if bits is not None:
            dh = F.gemm_mask_bits(dy2, w2, bits)
        elif ctx.act == 'none':
            dh = F.gemm(dy2, w2, b_col=True)
        else:
            dh = F.gemm(dy2, w2, b_col=True, act=ctx.act, aux=h if pre is None else pre, mode=2)
        dw1 = db1 = dw2 = db2 = None
        if not _defer(ctx.group, dy2, h, w2, ctx.biases[1], ng[3], ctx.has_b2 and ng[4], owner=id(ctx)):
            dw2, db2 = _wgrad(dy2, h, ng[3], ctx.has_b2 and ng[4])
        dx = None
        if ng[0]:  # the residual is x itself: its gradient is the aux operand of the dX GEMM
            dx = F.gemm(dh, w1, b_col=True, aux=_2d(dres), mode=1).view(dy.shape)
        if not _defer(ctx.group, dh, x2, w1, ctx.biases[0], ng[1], ctx.has_b1 and ng[2], owner=id(ctx)):
            dw1, db1 = _wgrad(dh, x2, ng[1], ctx.has_b1 and ng[2])
        return dx, dw1, db1, dw2, db2, None, dgamma, dbeta, None, None, None


_ADAPTER_R4 = os.environ.get('PASERO_ADAPTER_R4', '0') not in ('', '0')  # (A/B: the adapter backward as before round 5)


class AdapterFn(Function):
    """y = res + s · up(act(down(LN(x))))   — the bottleneck adapter of Bapna et al. and, without LayerNorm / biases /
    activation and with `res` = the frozen layer's output, LoRA (pasero/models/modules.py:248-370 AdapterLayer.forward,
    :67-100 Linear.forward).  Three launches forward (LayerNorm, down-projection with the activation in its epilogue,
    up-projection with the scale and the residual in its epilogue); backward fuses act′ into the dA GEMM, the bias
    gradients into the weight-gradient GEMMs and the residual-branch gradient into the LayerNorm backward."""

    @staticmethod
    def forward(ctx, x, res, ln_w, ln_b, eps: float, down_w, down_b, up_w, up_b, act: str, scaling: float, drop=None):
        # `drop`: the DropLink of the `residual + dropout(.)` that produced x (a pre-norm layer's last block end), when x is also the
        # adapter's residual: the LayerNorm backward below then computes the WHOLE gradient of x and can write it through that mask
        ctx.drop = drop if (drop is not None and drop.p > 0 and res is x and ln_w is not None and x.dtype != torch.float32) else None
        x2 = _2d(_contig(x))
        grad = any(wants_grad(ctx))
        if ln_w is not None:
            h, _, mean, rstd = F.residual_ln_fwd(x2, None, ln_w, ln_b, eps, want_z=False)
        else:
            h, mean, rstd = x2, None, None
        need_pre = grad and act not in ('none', 'relu')
        pre = torch.empty(x2.size(0), down_w.size(0), dtype=x.dtype, device=x.device) if need_pre else None
        a = F.gemm(h, down_w, bias=down_b, act=act, preact=pre)
        ub = up_b if (up_b is None or scaling == 1.0) else F.scale(up_b, None, scaling)  # (v + b)·s = s·v + s·b
        res2 = _2d(_contig(res)) if res is not None else None
        y = F.gemm(a, up_w, bias=ub, aux=res2, mode=1 if res2 is not None else 0, alpha=scaling)
        ctx.act, ctx.scaling, ctx.has_ln, ctx.has_ln_b = act, scaling, ln_w is not None, ln_b is not None
        ctx.res_is_x = res is x
        ctx.has_res = res is not None
        ctx.has_db, ctx.has_ub = down_b is not None, up_b is not None
        ctx.save_for_backward(x2, h if ln_w is not None else None, mean, rstd, ln_w, down_w, up_w, a, pre)
        return y.view(*x.shape[:-1], up_w.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, h, mean, rstd, ln_w, down_w, up_w, a, pre = ctx.saved_tensors
        if h is None:
            h = x2
        ng = ctx.needs_input_grad  # x, res, ln_w, ln_b, eps, down_w, down_b, up_w, up_b
        dy2 = _2d(_contig(dy))
        s = ctx.scaling
        dup_w = dup_b = ddown_w = ddown_b = dln_w = dln_b = dx = None
        want_ub = ctx.has_ub and ng[8]
        want_db = ctx.has_db and ng[6]
        need_da = ng[0] or ng[2] or ng[3] or ng[5] or ng[6]
        da = None
        if need_da:
            # d(pre-activation) = s · (dY · W_up) ⊙ act′: a contraction over d into <= 64 columns.  As a (row, col) product it
            # ran on the 128-tile kernel (16 000 x 64, K = 1024: 37 us); over a transposed copy of W_up (64 x d: 128 KB) it is a
            # (row, row) product the few-rows kernel takes (22 us), act′ in its epilogue
            b, b_col = up_w, True
            if up_w.size(1) <= 64 and up_w.size(0) % 64 == 0 and dy2.dtype != torch.float32 and not _ADAPTER_R4:
                b, b_col = up_w.t().contiguous(), False
            if ctx.act == 'none':
                da = F.gemm(dy2, b, b_col=b_col, alpha=s)
            else:
                da = F.gemm(dy2, b, b_col=b_col, alpha=s, act=ctx.act, aux=a if pre is None else pre, mode=2)
        # the two weight gradients (d x r and r x d, contraction over all rows) in ONE grouped launch where the kernel takes
        # them (16-bit, r >= 64: pk_gemm_wgrad_group_eligible), bias sums included; one by one otherwise
        probs = []
        if ng[7] or want_ub:
            probs.append(('up', dy2, a, ng[7], want_ub))
        if need_da and (ng[5] or want_db):
            probs.append(('down', da, h, ng[5], want_db))
        grouped = (len(probs) == 2 and not _ADAPTER_R4
                   and all(w and F.wgrad_group_eligible(g_, x_) for _, g_, x_, w, _ in probs))
        if grouped:
            (dup_w, dup_b), (ddown_w, ddown_b) = F.wgrad_group([(g_, x_, wb) for _, g_, x_, _, wb in probs])
        else:
            for name, g_, x_, w, wb in probs:
                if name == 'up':
                    dup_w, dup_b = _wgrad(g_, x_, w, wb)
                else:
                    ddown_w, ddown_b = _wgrad(g_, x_, w, wb)
        if s != 1.0:
            dup_w = F.scale(dup_w, None, s) if dup_w is not None else None
            dup_b = F.scale(dup_b, None, s) if dup_b is not None else None
        if need_da:
            if ng[0] or ng[2] or ng[3]:
                if ctx.has_ln:
                    dh = F.gemm(da, down_w, b_col=True)
                    want_pg = ng[2] or (ctx.has_ln_b and ng[3])
                    d = ctx.drop if (ctx.res_is_x and ng[0]) else None
                    dx, masked, dln_w, dln_b = F.residual_ln_bwd(dh, dy2 if ctx.res_is_x else None, x2, ln_w, mean, rstd,
                                                                 want_dres=True, want_dx=d is not None, want_param_grads=want_pg,
                                                                 has_beta=ctx.has_ln_b, drop_p=d.p if d else 0.0,
                                                                 seed=d.seed if d else 0, offset=d.offset if d else 0)
                    if d is not None:
                        dx = dx.view(*dy.shape[:-1], x2.size(1))
                        d.offer(masked.view(dx.shape), dx)
                elif ctx.res_is_x:
                    dx = F.gemm(da, down_w, b_col=True, aux=dy2, mode=1)  # residual branch folded into the epilogue
                else:
                    dx = F.gemm(da, down_w, b_col=True)
        elif ctx.res_is_x and ng[0]:
            dx = dy2
        dres = None
        if ctx.has_res and not ctx.res_is_x and ng[1]:
            dres = dy
        if dx is not None:
            dx = dx.view(*dy.shape[:-1], x2.size(1)) if dx.dim() == 2 else dx
        return (dx if ng[0] else None, dres, dln_w if ng[2] else None, dln_b if (ctx.has_ln_b and ng[3]) else None, None,
                ddown_w, ddown_b, dup_w, dup_b, None, None, None)


class ResidualDropoutFn(Function):
    """z = residual + dropout(x)   (pre-norm blocks: pasero/models/transformer.py:1043-1044,1049-1050).
    `link`: a DropLink the caller will hand to the fork that consumes z — its backward then supplies the masked gradient."""

    @staticmethod
    def forward(ctx, x, residual, p: float, link=None):
        x, residual = _contig(x), _contig(residual)
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        _, z, _, _ = F.residual_ln_fwd(x, residual, None, None, 0.0, p, seed, offset)
        ctx.p, ctx.seed, ctx.offset = p, seed, offset
        ctx.link = link
        if link is not None:
            link.p, link.seed, link.offset = p, seed, offset
        return z

    @staticmethod
    def backward(ctx, dz):
        dz = _contig(dz)
        dx = dz
        if ctx.p > 0 and ctx.needs_input_grad[0]:
            masked = ctx.link.take(dz) if ctx.link is not None else None
            # (masked: written by the consumer's LayerNorm backward kernel in the pass that made dz)
            dx = masked if masked is not None else F.dropout(dz, ctx.p, ctx.seed, ctx.offset)
        elif ctx.link is not None:
            ctx.link.take(dz)
        return dx, dz, None, None


class ResidualDropoutLnFn(Function):
    """z = residual + dropout(x) at the end of a pre-norm sub-block AND y = LayerNorm(z) at the head of the next one, as ONE node
    -> (y, z)   (pasero/models/transformer.py:1043-1044 followed by :1070-1075 of the next block).  Forward: one launch of
    pk_residual_ln_fwd with both outputs (the statistics are taken on the rounded z, as a separate pass over z takes them: bit for
    bit ResidualDropoutFn + LayerNormForkFn) — what the native layer call does between its blocks (csrc/layer.cpp,
    `end_and_next_norm`).  Backward: the one kernel call those two nodes share through their DropLink — the LayerNorm gradient plus
    the residual branch's (`dz_extra`), and the same sum written once more through the dropout mask for x."""

    @staticmethod
    def forward(ctx, x, residual, p: float, gamma, beta, eps: float):
        x, residual = _contig(x), _contig(residual)
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        y, z, mean, rstd = F.residual_ln_fwd(x, residual, gamma, beta, eps, p, seed, offset)
        ctx.p, ctx.seed, ctx.offset, ctx.has_beta = p, seed, offset, beta is not None
        ctx.save_for_backward(z, gamma, mean, rstd)
        ctx.set_materialize_grads(False)  # (an output nobody consumed arrives as None, not as zeros)
        return y, z

    @staticmethod
    def backward(ctx, dy, dz):
        z, gamma, mean, rstd = ctx.saved_tensors
        ng = ctx.needs_input_grad
        if dy is None:  # only z was used: the plain residual + dropout
            if dz is None:
                return None, None, None, None, None, None
            dz = _contig(dz)
            dx = F.dropout(dz, ctx.p, ctx.seed, ctx.offset) if (ctx.p > 0 and ng[0]) else dz
            return (dx if ng[0] else None), (dz if ng[1] else None), None, None, None, None
        want_pg = ng[3] or (ctx.has_beta and ng[4])
        extra = _contig(dz) if dz is not None else None
        dres, masked, dgamma, dbeta = F.residual_ln_bwd(
            _contig(dy), extra, z, gamma, mean, rstd, want_dres=True, want_dx=ctx.p > 0 and ng[0], want_param_grads=want_pg,
            has_beta=ctx.has_beta, drop_p=ctx.p, seed=ctx.seed, offset=ctx.offset)
        dx = masked if ctx.p > 0 else dres
        return (dx if ng[0] else None, dres if ng[1] else None, None, dgamma if ng[3] else None,
                dbeta if (ctx.has_beta and ng[4]) else None, None)


class DropoutFn(Function):
    """nn.Dropout with a regenerable Philox mask"""

    @staticmethod
    def forward(ctx, x, p: float):
        ctx.p = p
        ctx.seed, ctx.offset = rng.next_offset()
        return F.dropout(_contig(x), p, ctx.seed, ctx.offset)

    @staticmethod
    def backward(ctx, dy):
        return F.dropout(_contig(dy), ctx.p, ctx.seed, ctx.offset), None


# The tied table's gradient in ONE (V, d) tensor.  With `tied_output_projection` (and `shared_embeddings`) the embedding matrix
# receives the projection's dense dW and the sparse rows of one or two lookups (pasero/models/transformer.py:151-153,
# modules.py:935-947); as three autograd contributions that costs, per lookup, a (V, d) zero fill + a dense `add` pass over the
# table (NLLB-1.3B: 525 MB each, 0.8 ms per step).  Instead:
#   * the model marks the table it hands to the vocabulary loss (`tie_table`: a tensor hook on it, registered once);
#   * the vocabulary loss's backward — the first node of a backward pass — opens a SESSION for that table;
#   * a lookup of a table with an open session does not compute a (V, d) gradient: it leaves (ids, dOut, arguments) in the
#     session and returns None;
#   * the hook fires when autograd has summed every contribution to the table's gradient (all producers have reported, the
#     deferring lookups too) and adds the deferred rows INTO that sum (pk_embed_bwd_acc: touched rows only, one rounding).
# Nothing depends on which tensor object the engine keeps in its input buffers: two graphs back-propagated together
# (`(l1 + l2).backward()`) put two dense contributions and four lookups into one session, and the hook sees their sum.  The
# session ends with the pass (an engine callback); a lookup without a session behaves as before.  PASERO_NO_GRAD_SINK=1:
# three separate contributions, as before round 5 (A/B).
_NO_GRAD_SINK = os.environ.get('PASERO_NO_GRAD_SINK', '0') not in ('', '0')
_table_sessions = {}  # (data_ptr, shape, dtype) of a tied table -> lookups deferred in the running backward pass


def _table_key(weight: Tensor):
    return (weight.data_ptr(), tuple(weight.shape), weight.dtype)


def tie_table(weight: Tensor) -> Tensor:
    """mark `weight` as a table that is both looked up and used as the vocabulary projection (see above); idempotent"""
    if _NO_GRAD_SINK or not weight.requires_grad or not weight.is_cuda or getattr(weight, '_pk_tied_hook', None) is not None:
        return weight
    ref = weakref.ref(weight)

    def add_deferred_rows(grad):
        w = ref()
        items = _table_sessions.pop(_table_key(w), None) if w is not None else None
        if not items:
            return None
        if grad is None:  # an undefined sum (a nested / reentrant pass whose dense contribution never came): start from zeros
            grad = torch.zeros_like(w)
        g = grad if grad.is_contiguous() else grad.contiguous()
        for ids, dout, (V, pad, scale, p, seed, offset) in items:
            F.embed_bwd(ids, dout, V, pad, scale, p, seed, offset, into=g)
        return g
    weight._pk_tied_hook = weight.register_hook(add_deferred_rows)
    return weight


def _open_table_session(key):
    if not _table_sessions:
        torch.autograd.Variable._execution_engine.queue_callback(_table_sessions.clear)
    _table_sessions.setdefault(key, [])


class EmbeddingFn(Function):
    """dropout(E[ids] * scale + pos[pos_start : pos_start+T])
    (pasero/models/modules.py:916-933; pasero/models/transformer.py:727-744, 866-878)"""

    @staticmethod
    def forward(ctx, ids, weight, pos_table, scale: float, pos_start: int, p: float, padding_idx: int):
        ids = _contig(ids)
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        out = F.embed_fwd(ids, weight, pos_table, scale, pos_start, p, seed, offset)
        ctx.args = (scale, pos_start, p, seed, offset, padding_idx, weight.size(0))
        ctx.pos_rows = pos_table.size(0) if pos_table is not None else 0
        # rows are deferred only for the tensor `tie_table` marked — the one whose hook will add them; another autograd
        # tensor over the same storage (a view, a detached copy that requires grad) has a gradient of its own
        # (checked in backward: the model ties the table when it reaches the loss, after the lookups of the same pass)
        ctx.table, ctx.table_ref = _table_key(weight), weakref.ref(weight)
        ctx.save_for_backward(ids)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        scale, pos_start, p, seed, offset, padding_idx, V = ctx.args
        dout = _contig(dout)
        dE = dpos = None
        if ctx.needs_input_grad[1]:
            w = ctx.table_ref()
            tied = w is not None and getattr(w, '_pk_tied_hook', None) is not None
            session = _table_sessions.get(ctx.table) if tied else None
            if session is not None and dout.dtype == ctx.table[2]:
                session.append((ids, dout, (V, padding_idx, scale, p, seed, offset)))  # (added into the table's summed gradient)
            else:
                dE = F.embed_bwd(ids, dout, V, padding_idx, scale, p, seed, offset)
        if ctx.needs_input_grad[2]:  # learned positions: sum over the batch of the (masked) gradient
            B, T, d = dout.shape
            dm = F.dropout(dout, p, seed, offset) if p > 0 else dout
            rows = F.colsum(dm.view(B, T * d)).view(T, d)
            dpos = torch.zeros(ctx.pos_rows, d, dtype=dout.dtype, device=dout.device)
            dpos[pos_start:pos_start + T] = rows
        return None, dE, dpos, None, None, None, None


_CE_BUDGET = int(os.environ.get('PASERO_CE_BUDGET_MB', '1200')) << 20  # bytes of logits per chunk (env: the sweep below)


def _ce_chunk_rows(rows: int, V: int, itemsize: int, budget_bytes: int = _CE_BUDGET) -> int:
    """rows per logits chunk: the (rows, V) logits never exist as a whole, only a chunk of <= 1200 MiB at a time.
    Rounds 1-2 used 128 MiB so that a chunk would stay in the 256 MiB Infinity Cache between the GEMM that writes it, the
    CE kernel that rewrites it in place and the two gradient GEMMs that read it; the GEMM epilogues store streaming since
    then and residency buys nothing, while fewer, larger GEMMs do.  Same-box sweep of the budget (round 3, ms per step):
    C2 (V = 8 032) 64 MiB 13.57, 128 13.23, 256 13.21, >= 512 (one chunk) 13.05-13.08; transformer_big (V = 70 376) 64 MiB
    57.4, 128 52.1, 256 50.9, 512 50.7-50.9, 1024 53.8, 2048 50.0, 4096 54.2 — the ups and downs were ONE GEMM: the chunk's
    dX GEMM (rows x d, K = V) ran unsplit on 96-128 of the 256 CUs at 6144 / 8192 rows because `choose_splitk` asked for
    K % 64 == 0 (1.5 ms instead of 0.8); with that fixed, on a slower box: 512 MiB 52.1, 600 51.8, 900 51.8, 1200 (8192 rows)
    51.3, 1536 52.3, 2048 51.3; NLLB shapes (V = 256 206) 512 MiB 86.1, 1200 85.4, 2400 85.4."""
    per = max(1, budget_bytes // (V * itemsize))
    per = max(128, (per // 128) * 128)
    if per >= 2048:  # whole waves of 256-row tiles over the 256 CUs: 8320 rows are 33 tile rows, 8192 are 32 (C2: the
        per = (per // 2048) * 2048  # chunk GEMMs lose a fifth of their rate to the one tile row too many; same-box
        # A/B of the C2 step: 19.22-19.33 vs 19.37-19.57 ms)
    return min(rows, per)


_NO_PAD_VOCAB = os.environ.get('PASERO_NO_PAD_VOCAB', '0') not in ('', '0')  # (A/B: V % 8 != 0 on the 128-tile kernel, as before round 4)


class VocabCrossEntropyFn(Function):
    """Tied vocabulary projection + label-smoothed cross-entropy, chunked over rows so the (rows, V) logits never
    exist as a whole (pasero/models/modules.py:935-947 + pasero/models/transformer.py:354-380).
    Returns sums = [loss, nll_loss, num_tokens] (fp32, device).  Only sums[0] is differentiable.
    The gradient chunks dX, dW are produced in the forward pass (the logits chunk is still cache-resident) and only
    multiplied by the incoming scalar gradient in backward."""

    @staticmethod
    def forward(ctx, x, weight, target, padding_idx: int, eps: float):
        x2 = _2d(_contig(x))
        tgt = _contig(target).view(-1)
        rows, V = x2.size(0), weight.size(0)
        grad = any(wants_grad(ctx)[:2])
        row_loss = torch.empty(rows, dtype=torch.float32, device=x.device)
        row_nll = torch.empty(rows, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x2) if grad else None
        dw = None
        step = _ce_chunk_rows(rows, V, x.element_size())
        ldp = (V + 15) // 16 * 16  # padded leading dimension: rows stay 16-byte addressable for any vocabulary size
        # dW = dlogitsᵀ·X re-reads and re-writes the whole (V, d) gradient every time it is accumulated: with a large
        # vocabulary a chunk is only a few hundred rows (C5: 256) and that traffic, not the contraction, is the cost
        # (a K = 256 GEMM moving 1 GB: 334 TFLOP/s).  So the gradient chunks of a GROUP of consecutive chunks are kept
        # side by side (HBM is not the constraint: <= 2 GiB) and dW is accumulated once per group, over ~4096 rows
        # (same-box A/B: C5 118.0 -> 111.7 ms per step, C3 70.8 -> 69.4; C2 has one 8192-row chunk per group anyway).
        group = 1
        if grad and step < rows:
            group = max(1, min(-(-4096 // step), (2 << 30) // (step * ldp * x.element_size())))
        logits = torch.empty(group * step, ldp, dtype=x.dtype, device=x.device)[:, :V]
        padded = V % 8 != 0 and x.dtype in (torch.bfloat16, torch.float16) and not _NO_PAD_VOCAB

        def weight_grad_of(g0, g1, first):  # rows [g0, g1) of x2 <-> the first g1 - g0 rows of the group buffer
            lgs = logits[: g1 - g0]
            sk = F.choose_splitk(V, x2.size(1), g1 - g0)  # (V x d) output, contraction over the group's rows
            if first:
                return F.gemm(lgs, x2[g0:g1], a_col=True, b_col=True, splitk=sk)
            return F.gemm(lgs, x2[g0:g1], a_col=True, b_col=True, aux=dw, mode=1, out=dw, splitk=sk)

        g0 = 0
        for i, r0 in enumerate(range(0, rows, step)):
            r1 = min(rows, r0 + step)
            slot = (i % group) * step
            lg = logits[slot: slot + r1 - r0]
            # (a vocabulary that is no multiple of 8 — NLLB's 256 206: the buffer's rows are padded, pk_ce_rows leaves zeros
            # in the pad columns of its gradient, and both GEMMs may then work in whole 16-byte chunks: the 256-tile kernel)
            F.gemm(x2[r0:r1], weight, out=lg, pad_n=padded)
            F.ce_rows(lg, tgt[r0:r1], padding_idx, eps, row_loss[r0:r1], row_nll[r0:r1],
                      dlogits=lg if grad else None)
            if grad:
                # dX chunk: few output tiles (chunk rows x d) but a vocabulary-long contraction -> split-K
                F.gemm(lg, weight, b_col=True, out=dx[r0:r1], splitk=F.choose_splitk(r1 - r0, x2.size(1), V), pad_k=padded)
                if (i + 1) % group == 0 or r1 == rows:
                    dw = weight_grad_of(g0, r1, dw is None)
                    g0 = r1
        sums = F.ce_finalize(row_loss, row_nll, tgt, padding_idx)
        ctx.x_shape = x.shape
        tied = getattr(weight, '_pk_tied_hook', None) is not None and not _NO_GRAD_SINK
        ctx.table = _table_key(weight) if tied else None
        if grad:
            ctx.save_for_backward(dx, dw)
        return sums

    @staticmethod
    def backward(ctx, dsums):
        dx, dw = ctx.saved_tensors
        g = _contig(dsums)[:1].float()  # d(total)/d(loss); nll / num_tokens are logging outputs
        gx = F.scale(dx, g).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        gw = F.scale(dw, g) if ctx.needs_input_grad[1] else None
        if gw is not None and ctx.table is not None:
            _open_table_session(ctx.table)  # (lookups of this table hand their rows to the table's hook: see tie_table)
        return gx, gw, None, None, None


class ActivationFn(Function):
    """stand-alone activation (non-fused fallback of pasero/models/modules.py:220-228)"""

    @staticmethod
    def forward(ctx, x, act: str):
        x = _contig(x)
        ctx.act = act
        ctx.save_for_backward(x)
        return F.act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return F.act_bwd(_contig(dy), x, ctx.act), None


class GLUFn(Function):
    """nn.GLU over the channel (last) dim of a channels-last tensor (pasero/models/modules.py:802)"""

    @staticmethod
    def forward(ctx, x):
        x = _contig(x)
        ctx.save_for_backward(x)
        return F.glu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return F.glu_bwd(_contig(dy), x)


class Conv1dChannelsLastFn(Function):
    """nn.Conv1d(C_in, C_out, k, stride, padding) + optional activation on a channels-last (B, L, C_in) input
    (pasero/models/modules.py:793-799,829-833) as an implicit GEMM: with R = ceil((L+2p)/stride) rows per batch, row r
    of the im2col matrix is the contiguous window x_pad[b, r*stride : r*stride+k, :] — a strided VIEW (lda =
    stride*C_in), never materialised.  `weight` keeps the nn.Conv1d layout (C_out, C_in, k)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride: int, padding: int, act: str):
        B, L, C = x.shape
        O, _, k = weight.shape
        Lout = (L + 2 * padding - k) // stride + 1
        R = -(-(L + 2 * padding) // stride)
        Lp = R * stride
        xp = x.new_zeros(B * Lp + k, C)  # + k rows: the windows of the last (discarded) rows stay in bounds
        xp[:B * Lp].view(B, Lp, C)[:, padding:padding + L] = x
        A = torch.as_strided(xp, (B * R, k * C), (stride * C, 1))
        wr = weight.permute(0, 2, 1).reshape(O, k * C).contiguous()  # [o][(j, c)]
        need_pre = act != 'none' and any(wants_grad(ctx))
        pre = torch.empty(B * R, O, dtype=x.dtype, device=x.device) if need_pre else None
        y = F.gemm(A, wr, bias=bias, act=act, preact=pre)
        ctx.geom = (B, L, C, O, k, stride, padding, Lout, R)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(xp, wr, pre)
        return y.view(B, R, O)[:, :Lout].contiguous()

    @staticmethod
    def backward(ctx, dy):
        xp, wr, pre = ctx.saved_tensors
        B, L, C, O, k, stride, padding, Lout, R = ctx.geom
        dyf = dy.new_zeros(B, R, O)
        dyf[:, :Lout] = dy
        dz = dyf.view(B * R, O)
        if ctx.act != 'none':
            dz = F.act_bwd(dz, pre, ctx.act)
        A = torch.as_strided(xp, (B * R, k * C), (stride * C, 1))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dA = F.gemm(dz, wr, b_col=True)
            dx = F.col2im1d(dA, B, L, C, R, Lout, k, stride, padding)
        dwr, db = _wgrad(dz, A, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        if dwr is not None:
            dw = dwr.view(O, k, C).permute(0, 2, 1).contiguous()
        return dx, dw, db, None, None, None


class ConvStackFn(Function):
    """A stack of Conv1d + activation (Whisper's subsampler: k3 s1 GELU, k3 s2 GELU; pasero/models/modules.py:819-834 without the
    GLU) as ONE node: the same implicit GEMMs as `Conv1dChannelsLastFn`, conv by conv, minus the copies between them.  With R_i
    rows per batch out of conv i and Lp_{i+1} = R_{i+1} * stride_{i+1} rows per batch in the zero-padded input of conv i + 1:
    where R_i == Lp_{i+1} (every stride-1 'same' conv in front of anything) GEMM row b R_i + r IS row b Lp_{i+1} + r of that
    buffer, so the GEMM of conv i writes its activations straight into the padded input of conv i + 1, `padding` rows down —
    no compaction of the R_i - Lout_i rows past the end (they land on padding rows and are zeroed: B x 2 rows instead of a pass
    over the tensor), no zero fill, no copy into the padded layout; on the way back `pk_col2im1d` writes the input gradient of
    conv i + 1 over R_i rows per batch, which is the (B, R_i, C) layout the backward of conv i contracts.  Per Whisper step (16 x
    3000 x 512 between the convs): two 49 MB copies, a 49 MB clone and two 49 MB zero fills less."""

    @staticmethod
    def forward(ctx, x, act: str, geoms, *wb):
        n = len(geoms)
        B, L, C = x.shape
        plan = []
        for i, (stride, padding) in enumerate(geoms):
            O, Cw, k = wb[2 * i].shape
            assert Cw == C, (Cw, C)
            Lout = (L + 2 * padding - k) // stride + 1
            R = -(-(L + 2 * padding) // stride)
            plan.append((L, C, O, k, stride, padding, Lout, R))
            L, C = Lout, O
        need_pre = act != 'none' and any(wants_grad(ctx))
        saved, direct = [], []
        # the padded input of conv 0
        L0, C0, _, k0, s0, p0, _, R0 = plan[0]
        xp = x.new_zeros(B * R0 * s0 + k0, C0)  # + k rows: the windows of the last (discarded) rows stay in bounds
        xp[:B * R0 * s0].view(B, R0 * s0, C0)[:, p0:p0 + L0] = x
        y = None
        for i, (Li, Ci, O, k, stride, padding, Lout, R) in enumerate(plan):
            A = torch.as_strided(xp, (B * R, k * Ci), (stride * Ci, 1))
            wr = wb[2 * i].permute(0, 2, 1).reshape(O, k * Ci).contiguous()  # [o][(j, c)]
            pre = torch.empty(B * R, O, dtype=x.dtype, device=x.device) if need_pre else None
            nxt = plan[i + 1] if i + 1 < n else None
            straight = nxt is not None and R == nxt[7] * nxt[4] and nxt[5] <= nxt[3]
            direct.append(straight)
            if straight:
                _, _, _, kn, sn, pn, _, Rn = nxt
                Lpn = Rn * sn
                xn = torch.empty(B * Lpn + kn, O, dtype=x.dtype, device=x.device)
                F.gemm(A, wr, bias=wb[2 * i + 1], act=act, preact=pre, out=xn[pn:pn + B * R])
                xn[:pn].zero_()
                xn[pn + B * R:].zero_()
                # rows Lout .. R - 1 of every batch (windows past the end) sit on the padding rows between the batches
                torch.as_strided(xn, (B, R - Lout, O), (Lpn * O, O, 1), (pn + Lout) * O).zero_()
            else:
                y = F.gemm(A, wr, bias=wb[2 * i + 1], act=act, preact=pre).view(B, R, O)[:, :Lout]
                if nxt is not None:
                    _, _, _, kn, sn, pn, _, Rn = nxt
                    xn = x.new_zeros(B * Rn * sn + kn, O)
                    xn[:B * Rn * sn].view(B, Rn * sn, O)[:, pn:pn + Lout] = y
            saved += [xp, wr, pre]
            if nxt is not None:
                xp = xn
        ctx.plan, ctx.direct, ctx.act, ctx.B = plan, direct, act, B
        ctx.nsaved = [t is not None for t in saved]
        ctx.save_for_backward(*[t for t in saved if t is not None])
        return y.contiguous()

    @staticmethod
    def backward(ctx, dy):
        it = iter(ctx.saved_tensors)
        saved = [next(it) if has else None for has in ctx.nsaved]
        plan, B, n = ctx.plan, ctx.B, len(ctx.plan)
        grads = [None] * (2 * n)
        dx = None
        dyf = None  # (B, R_i, O_i): the output gradient of conv i in the GEMM's row layout, zero past Lout_i
        for i in range(n - 1, -1, -1):
            Li, Ci, O, k, stride, padding, Lout, R = plan[i]
            xp, wr, pre = saved[3 * i:3 * i + 3]
            if dyf is None:
                dyf = dy.new_zeros(B, R, O)
                dyf[:, :Lout] = dy
            dz = dyf.view(B * R, O)
            if ctx.act != 'none':
                dz = F.act_bwd(dz, pre, ctx.act)
            A = torch.as_strided(xp, (B * R, k * Ci), (stride * Ci, 1))
            dyf = None
            if i > 0 or ctx.needs_input_grad[0]:
                dA = F.gemm(dz, wr, b_col=True)
                if i > 0 and ctx.direct[i - 1]:
                    Rp = plan[i - 1][7]
                    dyf = F.col2im1d(dA, B, Rp, Ci, R, Lout, k, stride, padding)
                    if (Lout - 1) * stride + k - 1 - padding >= Li:  # a window reaches past the input's end: not this conv's to keep
                        dyf[:, Li:].zero_()
                else:
                    dxi = F.col2im1d(dA, B, Li, Ci, R, Lout, k, stride, padding)
                    if i > 0:
                        dy = dxi
                    else:
                        dx = dxi
            dwr, db = _wgrad(dz, A, ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[3 + 2 * i + 1])
            if dwr is not None:
                grads[2 * i] = dwr.view(O, k, Ci).permute(0, 2, 1).contiguous()
            grads[2 * i + 1] = db
        return (dx, None, None, *grads)


class AddPositionsFn(Function):
    """dropout(x * scale + pos[pos_start : pos_start+T]) for dense (speech) encoder inputs
    (pasero/models/transformer.py:739-744)"""

    @staticmethod
    def forward(ctx, x, pos_table, scale: float, pos_start: int, p: float):
        x = _contig(x)
        seed, offset = rng.next_offset() if p > 0 else (0, 0)
        ctx.args = (scale, pos_start, p, seed, offset)
        ctx.pos_rows = pos_table.size(0) if pos_table is not None else 0
        return F.add_positions(x, pos_table, scale, pos_start, p, seed, offset)

    @staticmethod
    def backward(ctx, dout):
        scale, pos_start, p, seed, offset = ctx.args
        dout = _contig(dout)
        dm = F.dropout(dout, p, seed, offset) if p > 0 else dout
        dx = dpos = None
        if ctx.needs_input_grad[0]:
            dx = dm if scale == 1.0 else F.scale(dm, None, scale)
        if ctx.needs_input_grad[1]:
            B, T, d = dout.shape
            rows = F.colsum(dm.view(B, T * d)).view(T, d)
            dpos = torch.zeros(ctx.pos_rows, d, dtype=dout.dtype, device=dout.device)
            dpos[pos_start:pos_start + T] = rows
        return dx, dpos, None, None, None


class CrossEntropyFn(Function):
    """Label-smoothed cross-entropy on materialised logits (the reference API `compute_loss(logits, target, ...)`,
    pasero/models/transformer.py:324-380).  Returns sums = [loss, nll_loss, num_tokens]."""

    @staticmethod
    def forward(ctx, logits, target, padding_idx: int, eps: float):
        lg = _2d(_contig(logits))
        tgt = _contig(target).view(-1)
        rows = lg.size(0)
        row_loss = torch.empty(rows, dtype=torch.float32, device=lg.device)
        row_nll = torch.empty(rows, dtype=torch.float32, device=lg.device)
        grad = wants_grad(ctx)[0]
        dl = torch.empty_like(lg) if grad else None
        F.ce_rows(lg, tgt, padding_idx, eps, row_loss, row_nll, dlogits=dl)
        ctx.shape = logits.shape
        if grad:
            ctx.save_for_backward(dl)
        return F.ce_finalize(row_loss, row_nll, tgt, padding_idx)

    @staticmethod
    def backward(ctx, dsums):
        (dl,) = ctx.saved_tensors
        return F.scale(dl, _contig(dsums)[:1].float()).view(ctx.shape), None, None, None


class RotaryFn(Function):
    """RoPE on the q|k part of a packed projection (pasero/models/modules.py:982-1025); backward = inverse rotation"""
    _keep_fp32 = (1, 2)  # the cos / sin tables

    @staticmethod
    def forward(ctx, x, cos_t, sin_t, ncols: int, pos_offset: int):
        ctx.ncols, ctx.pos_offset = ncols, pos_offset
        ctx.save_for_backward(cos_t, sin_t)
        return F.rope(_contig(x), cos_t, sin_t, ncols, pos_offset, inverse=False)

    @staticmethod
    def backward(ctx, dy):
        cos_t, sin_t = ctx.saved_tensors
        return F.rope(_contig(dy), cos_t, sin_t, ctx.ncols, ctx.pos_offset, inverse=True), None, None, None, None
