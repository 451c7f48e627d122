"""One C call per layer and direction for the STOCK post-norm Transformer layer (pk_layer_fwd / pk_layer_bwd,
include/pasero_hip.h, csrc/layer.cpp): `TransformerEncoderLayer.forward` / `TransformerDecoderLayer.forward`
(pasero/models/transformer.py:1056-1099, 1341-1417) as ONE autograd node whose forward and backward each enqueue the
layer's whole launch sequence from C — the same kernels, in the same order, with the same arguments and dropout offsets
as the per-op path (pasero_amd/autograd.py), so both paths agree bit for bit (tests/test_native_layer_gpu.py; across decoder
layers the encoder-output gradient is summed in the kv dX GEMMs with one rounding less, PASERO_NO_DENC_CHAIN=1 restores
autograd's additions) — instead of
~13 / ~27 Python dispatches with their tensor allocations and ctypes marshalling.  Host time of a C2 step: 9.6 -> ~4 ms.

A layer takes this path only when it is exactly the stock layer: post- or pre-norm, LayerNorm with bias, every Linear with
bias (the key projection may have none: Whisper), no LoRA / adapters / gated feed-forward / rotary positions / attention or activation dropout, no subclass hook
overridden, 16-bit parameters that all require a gradient, training-mode autograd on, no `return_layers`, no incremental
state, shapes the grouped weight-gradient launch takes.  Everything else stays on the per-op path (`takes` says which).
PASERO_NO_NATIVE_LAYER=1 switches it off."""
import ctypes
import os

import torch

from . import lib, rng
from .autograd import Function, block_tail_eligible
from .lib import ACT, PkLayer, check, dtype_code

_OFF = os.environ.get('PASERO_NO_NATIVE_LAYER', '0') not in ('', '0')
_NO_FUSED_TAIL = os.environ.get('PASERO_NO_FUSED_TAIL', '0') not in ('', '0')
_NO_DENC_CHAIN = os.environ.get('PASERO_NO_DENC_CHAIN', '0') not in ('', '0')  # (A/B: every decoder layer returns its own encoder gradient)
_NO_DROP_LINK = os.environ.get('PASERO_NO_DROP_LINK', '0') not in ('', '0')  # (A/B: every pre-norm layer draws its feed-forward mask itself in backward)
_sizes = {}  # (is_decoder, fused, prenorm, act, mask bits, B, T, S, d, f, heads, dtype, drop) -> (scratch_bytes, ws_bytes)



class DencChain:
    """The decoder layers of ONE decoder pass read the same encoder output: their gradients for it are summed by the kv dX
    GEMMs themselves, one layer into the next's output, instead of by autograd's elementwise additions.  The tally belongs
    to the pass (TransformerDecoder.forward opens it around its layer loop), not to the encoder tensor: a pass whose graph
    is never back-propagated takes its tally with it.  A backward that visits only SOME of the chained layers (autograd.grad
    towards an inner tensor) cannot hand the sum to autograd — the layer that would return it never runs — and is refused
    at the end of that backward instead of leaving the encoder without a gradient."""
    __slots__ = ('key', 'n', 'left', 'buf')

    def __init__(self, enc):
        self.key = (enc.data_ptr(), tuple(enc.shape), enc.dtype)
        self.n = self.left = 0
        self.buf = None

    def check(self):
        if self.left != self.n:
            self.left, self.buf = self.n, None
            raise RuntimeError('pasero_amd: a backward pass visited only some of the decoder layers whose encoder-output '
                               'gradients are summed in their GEMMs; run it with PASERO_NO_DENC_CHAIN=1 (every layer '
                               'then returns its own gradient)')


_chain = None  # the decoder pass being recorded


def open_chain(enc):
    """TransformerDecoder.forward, in front of its layer loop; returns what `close_chain` restores"""
    global _chain
    prev = _chain
    _chain = DencChain(enc) if (enc is not None and not _NO_DENC_CHAIN and torch.is_grad_enabled()) else None
    return prev


def close_chain(prev):
    global _chain
    _chain = prev


HOOKS_ENC = ('ffn', 'self_attention', 'self_attn_residual', 'self_attn_prenorm', 'self_attn_postnorm', 'ffn_residual',
             'ffn_prenorm', 'ffn_postnorm', 'forward')
HOOKS_DEC = HOOKS_ENC + ('cross_attention', 'cross_attn_residual', 'cross_attn_prenorm', 'cross_attn_postnorm')


def _attn_params(a):
    return [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.q_proj.bias, a.k_proj.bias, a.v_proj.bias,
            a.out_proj.weight, a.out_proj.bias]


def layer_params(layer, is_decoder: bool):
    """the layer's parameters in the order the autograd node takes them (and returns their gradients)"""
    ps = _attn_params(layer.self_attn) + [layer.self_attn_layer_norm.weight, layer.self_attn_layer_norm.bias]
    if is_decoder:
        ps += _attn_params(layer.encoder_attn) + [layer.encoder_attn_layer_norm.weight, layer.encoder_attn_layer_norm.bias]
    ps += [layer.fc1.weight, layer.fc1.bias, layer.fc2.weight, layer.fc2.bias, layer.final_layer_norm.weight,
           layer.final_layer_norm.bias]
    return ps


def grad_arena_params(layer, is_decoder: bool):
    """the layer's parameters in the order pk_layer_bwd lays their gradients out — three contiguous pieces: the [*, d]
    weights (q | k | v | out [| the same of the cross block] | fc1), fc2's weight, and the vectors (q k v out biases,
    LayerNorm weight and bias per attention block; fc1 bias, fc2 bias, final LayerNorm).  A gradient bucket that keeps
    these pieces contiguous (pasero_amd/ddp.py) receives the layer's gradients straight from the backward call."""
    blocks = [layer.self_attn] + ([layer.encoder_attn] if is_decoder else [])
    norms = [layer.self_attn_layer_norm] + ([layer.encoder_attn_layer_norm] if is_decoder else [])
    wd, vec = [], []
    for a, n in zip(blocks, norms):
        wd += [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.out_proj.weight]
        vec += [a.q_proj.bias, a.k_proj.bias, a.v_proj.bias, a.out_proj.bias, n.weight, n.bias]
    wd.append(layer.fc1.weight)
    vec += [layer.fc1.bias, layer.fc2.bias, layer.final_layer_norm.weight, layer.final_layer_norm.bias]
    return wd, [layer.fc2.weight], vec


def _static_ok(layer, is_decoder: bool) -> bool:
    """what does not change from call to call (decided once per layer object and training mode)"""
    from . import modules, transformer
    cls = transformer.TransformerDecoderLayer if is_decoder else transformer.TransformerEncoderLayer
    if any(getattr(type(layer), h) is not getattr(cls, h) for h in (HOOKS_DEC if is_decoder else HOOKS_ENC)):
        return False
    cfg = layer.cfg
    if cfg.check_inf or cfg.checkpoint_activations or layer.fc3 is not None or cfg.shared_norm:
        return False
    if layer.activation_dropout.p > 0 or layer.activation_fn.name not in ('relu', 'gelu', 'gelu_tanh', 'silu', 'none'):
        return False
    attns = [layer.self_attn] + ([layer.encoder_attn] if is_decoder else [])
    norms = [layer.self_attn_layer_norm, layer.final_layer_norm] + ([layer.encoder_attn_layer_norm] if is_decoder else [])
    for n in norms:
        if not isinstance(n, modules.LayerNorm) or getattr(n, 'weight', None) is None or getattr(n, 'bias', None) is None:
            return False
    for a in attns:
        if a.dropout > 0 or a.rotary_embed is not None or a.head_dim not in (64, 128):
            return False
        if any(m.lora is not None for m in (a.q_proj, a.k_proj, a.v_proj, a.out_proj)):
            return False
        if any(m.bias is None for m in (a.q_proj, a.v_proj, a.out_proj)):  # (k_proj may have no bias: attention_key_bias)
            return False
    if any(m.lora is not None or m.bias is None for m in (layer.fc1, layer.fc2)):
        return False
    return True


def _pad_ok(mask, B: int, S: int) -> bool:
    """the C side reads a key-padding mask as (B, S) bytes: anything else stays on the per-op path, which checks and refuses"""
    return mask is None or (mask.dtype == torch.bool and mask.is_cuda and tuple(mask.shape) == (B, S))


def takes(layer, x, enc, state, return_layers, is_decoder: bool, pad=None) -> bool:
    """`pad`: the key-padding mask the call would hand to C (encoder: of x; decoder: of the encoder output)"""
    if _OFF or state is not None or return_layers or not torch.is_grad_enabled() or not x.is_cuda or not x.requires_grad:
        return False
    if x.dtype not in (torch.bfloat16, torch.float16) or torch.is_autocast_enabled('cuda') or x.dim() != 3:
        return False
    key = (layer.training, x.dtype)
    ok = layer.__dict__.get('_native_static')
    if ok is None or ok[0] != key:  # (once per layer, mode and dtype: the configuration and the parameters' state)
        good = _static_ok(layer, is_decoder) and all(p is None or (p.dtype == x.dtype and p.requires_grad and p.is_cuda)
                                                     for p in layer_params(layer, is_decoder))
        ok = (key, good)
        layer.__dict__['_native_static'] = ok
    if not ok[1]:
        return False
    B, T, d = x.shape
    f = layer.fc1.weight.size(0)
    rows = B * T
    # the grouped weight-gradient launch takes outputs of >= 256 x 256 and whole 16-byte rows (pk_gemm_wgrad_group_eligible)
    # (and a contraction — the rows of the batch — of whole 16-byte columns: 1500 rows, one 30 s clip, are not)
    if rows < 256 or d < 256 or f < 256 or d % 8 or f % 8 or rows % 8:
        return False
    if is_decoder and (enc is None or enc.dtype != x.dtype or enc.dim() != 3 or enc.size(0) != B or enc.size(0) * enc.size(1) < 64
                       or (enc.size(0) * enc.size(1)) % 8):
        return False
    if not _pad_ok(pad, B, enc.size(1) if is_decoder else T):
        return False
    return True


def grad_buffers(layer, params, is_decoder: bool, d: int, f: int, dt, dev):
    """Where pk_layer_bwd writes a layer's parameter gradients: the [*, d] weights as row blocks of one 2-D tensor (q | k | v |
    out [| cross ...] | fc1), fc2's weight on its own, biases and LayerNorm parameters as pieces of one vector — three
    allocations, two splits.  Under the data-parallel reducer the three pieces are slices of the layer's gradient bucket
    (ddp.py lays the bucket out in this order): autograd adopts the returned views as `.grad` and the reducer has nothing to
    pack.  Only for fresh gradients (`.grad is None`: an accumulating micro-batch adds to what is there), and only for the
    FIRST node of the layer that reaches its backward before the reducer takes the bucket: a layer applied twice in one
    graph, or two forward passes back-propagated together, would otherwise both write the same slice before either is
    accumulated (both see `.grad is None`: AccumulateGrad runs after both producers) — later nodes get tensors of their own.
    -> (wd, w2, vec, row counts of wd's blocks, sizes of vec's pieces)"""
    nblk = 2 if is_decoder else 1
    wrows = [d, d, d, d] * nblk + [f]
    vsz = [d, d, d, d, d, d] * nblk + [f, d, d, d]  # q k v biases, out bias, ln weight, ln bias; ... fc1 b, fc2 b, ln w, ln b
    arena = layer.__dict__.get('_pk_grad_arena')
    if (arena is not None and not layer.__dict__.get('_pk_arena_claimed', False) and arena[0].dtype == dt
            and all(prm is not None and prm.grad is None for prm in params)):
        layer.__dict__['_pk_arena_claimed'] = True  # (until the reducer has taken the bucket: ddp._finalize / forward)
        flat, o_wd, o_w2, o_vec = arena
        wd = flat[o_wd: o_wd + sum(wrows) * d].view(sum(wrows), d)
        w2 = flat[o_w2: o_w2 + d * f].view(d, f)
        vec = flat[o_vec: o_vec + sum(vsz)]
    else:
        wd = torch.empty(sum(wrows), d, dtype=dt, device=dev)
        w2 = torch.empty(d, f, dtype=dt, device=dev)
        vec = torch.empty(sum(vsz), dtype=dt, device=dev)
    return wd, w2, vec, wrows, vsz


def grads_in_param_order(wd, w2, vec, wrows, vsz, nblk: int, params):
    """the pieces of `grad_buffers` as the gradients of `layer_params(...)`, in that order (None for an absent parameter)"""
    ws_, vs_ = wd.split(wrows, 0), vec.split(vsz, 0)
    grads = []
    for k in range(nblk):  # q.w k.w v.w q.b k.b v.b out.w out.b ln.w ln.b
        grads += [ws_[4 * k], ws_[4 * k + 1], ws_[4 * k + 2], vs_[6 * k], vs_[6 * k + 1], vs_[6 * k + 2], ws_[4 * k + 3],
                  vs_[6 * k + 3], vs_[6 * k + 4], vs_[6 * k + 5]]
    v0 = 6 * nblk
    grads += [ws_[4 * nblk], vs_[v0], w2, vs_[v0 + 1], vs_[v0 + 2], vs_[v0 + 3]]
    return [g if prm is not None else None for g, prm in zip(grads, params)]  # (a projection without bias)


class NativeLayerFn(Function):
    """y = layer(x [, encoder_out]) — forward: pk_layer_fwd; backward: pk_layer_bwd (one C call each)"""

    @staticmethod
    def forward(ctx, x, enc, self_pad, cross_pad, layer, is_decoder, links, *params):
        L = lib.load()
        a_self = layer.self_attn
        B, T, d = x.shape
        S = enc.size(1) if is_decoder else 0
        rows, rows_kv = B * T, B * S
        H = a_self.num_heads
        f = layer.fc1.weight.size(0)
        x = x if x.is_contiguous() else x.contiguous()
        if is_decoder:
            enc = enc if enc.is_contiguous() else enc.contiguous()
        dt, dev = x.dtype, x.device
        act = layer.activation_fn.name
        need_pre = act not in ('none', 'relu')
        p = float(layer.dropout.p) if layer.training else 0.0
        norm = layer.self_attn_layer_norm
        prenorm = bool(layer.prenorm)
        fused = (not prenorm and not _NO_FUSED_TAIL and block_tail_eligible(rows, a_self.out_proj.weight, x, norm.weight)
                 and block_tail_eligible(rows, layer.fc2.weight, x, layer.final_layer_norm.weight))
        # activations kept for backward: one 16-bit arena + one fp32 arena per layer call
        n16 = rows * (3 * d + 3 * d) + rows * (f * (2 if need_pre else 1) + 2 * d)
        n32 = B * H * T + 2 * rows + 2 * rows
        if is_decoder:
            n16 += rows * 4 * d + rows_kv * 2 * d
            n32 += B * H * T + 2 * rows
        if prenorm:  # + LayerNorm(block input) of every sub-block
            n16 += rows * d * (3 if is_decoder else 2)
        # ReLU at base width: the mask for backward also as one bit per element (pk_gemm_relu_bits; same rule as the per-op path)
        from . import functional as PF
        key = ('bits', rows, d, f, dt, x.data_ptr() % 16, layer.fc1.weight.data_ptr() % 16)
        use_bits = layer.__dict__.get('_native_bits')
        if use_bits is None or use_bits[0] != key:
            use_bits = (key, act == 'relu' and PF.relu_bits_eligible(x.view(rows, d), layer.fc1.weight))
            layer.__dict__['_native_bits'] = use_bits
        use_bits = use_bits[1]
        if use_bits:
            n16 += rows * f // 16
        a16 = torch.empty(n16, dtype=dt, device=dev)
        a32 = torch.empty(n32, dtype=torch.float32, device=dev)
        es = 2
        cur16, cur32 = [a16.data_ptr()], [a32.data_ptr()]

        def t16(n):
            ptr = cur16[0]
            cur16[0] += n * es
            return ptr

        def t32(n):
            ptr = cur32[0]
            cur32[0] += n * 4
            return ptr

        lay = PkLayer()
        lay.dtype, lay.is_decoder, lay.fused_tail, lay.act = dtype_code(x), int(is_decoder), int(fused), ACT[act]
        lay.B, lay.T, lay.S, lay.d, lay.f, lay.heads, lay.prenorm = B, T, S, d, f, H, int(prenorm)
        lay.eps, lay.drop_p = float(norm.eps), p
        lay.attn_scale = 1.0 / (a_self.head_dim ** 0.5) if a_self.scaled else 1.0
        lay.x, lay.enc = x.data_ptr(), (enc.data_ptr() if is_decoder else None)
        lay.self_pad = self_pad.data_ptr() if (self_pad is not None and not is_decoder) else None
        lay.cross_pad = cross_pad.data_ptr() if (cross_pad is not None and is_decoder) else None
        seed = 0

        def attn_block(blk, a, ln, cross):
            w, b = a._flat()
            blk.w_in, blk.b_in = w.data_ptr(), (b.data_ptr() if b is not None else None)
            blk.w_o, blk.b_o = a.out_proj.weight.data_ptr(), a.out_proj.bias.data_ptr()
            blk.ln_g, blk.ln_b = ln.weight.data_ptr(), ln.bias.data_ptr()
            blk.proj = t16(rows * (d if cross else 3 * d))
            blk.kv = t16(rows_kv * 2 * d) if cross else None
            blk.attn, blk.z, blk.y = t16(rows * d), t16(rows * d), t16(rows * d)
            blk.ln_out = t16(rows * d) if prenorm else None
            blk.lse, blk.mean, blk.rstd = t32(B * H * T), t32(rows), t32(rows)

        # dropout offsets in the per-op path's order: self block end, cross block end, feed-forward block end
        attn_block(lay.self_, a_self, layer.self_attn_layer_norm, False)
        if p > 0:
            seed, lay.self_.drop_offset = rng.next_offset()
        if is_decoder:
            attn_block(lay.cross, layer.encoder_attn, layer.encoder_attn_layer_norm, True)
            if p > 0:
                seed, lay.cross.drop_offset = rng.next_offset()
        fb = lay.ffn
        fb.w1, fb.b1, fb.w2, fb.b2 = (layer.fc1.weight.data_ptr(), layer.fc1.bias.data_ptr(), layer.fc2.weight.data_ptr(),
                                      layer.fc2.bias.data_ptr())
        fb.ln_g, fb.ln_b = layer.final_layer_norm.weight.data_ptr(), layer.final_layer_norm.bias.data_ptr()
        fb.h = t16(rows * f)
        fb.pre = t16(rows * f) if need_pre else None
        fb.bits = t16(rows * f // 16) if use_bits else None
        fb.z = t16(rows * d)
        fb.y = t16(rows * d)
        fb.ln_out = t16(rows * d) if prenorm else None
        y_ptr = fb.z if prenorm else fb.y  # the layer's output
        fb.mean, fb.rstd = t32(rows), t32(rows)
        if p > 0:
            seed, fb.drop_offset = rng.next_offset()
        lay.seed = seed
        # masked-gradient hand-over between stacked pre-norm layers (autograd.DropLink): `link_out` tells the consumer of this
        # layer's output which mask the feed-forward block end drew; `link_in` is the same note from the producer of x
        link_in, link_out = links
        if link_out is not None and prenorm and p > 0:
            link_out.p, link_out.seed, link_out.offset = p, seed, fb.drop_offset
        else:
            link_out = None
        if not (link_in is not None and prenorm and p > 0 and link_in.p == p and link_in.seed == seed):
            link_in = None
        ctx.links = (link_in, link_out)
        lay.stream = lib.stream_ptr()
        # a few-rows fc2 with a long contraction (NLLB's 8192 -> 1024 at a 2048-row decoder batch: 32 of 256 CUs as it stands)
        # runs split-K in the forward pass too when the layer brings a workspace (pk_layer_fwd_ws: 0 when no GEMM asks)
        fws = L.pk_layer_fwd_ws(ctypes.byref(lay))
        if fws:
            wsf = lib.workspace(fws, x.device, 'layer_ws')
            lay.ws, lay.ws_bytes = wsf.data_ptr(), wsf.numel()
        check(L.pk_layer_fwd(ctypes.byref(lay)), 'pk_layer_fwd')
        ctx.lay, ctx.layer, ctx.is_decoder = lay, layer, is_decoder
        chain = None
        if is_decoder and _chain is not None and _chain.key == (enc.data_ptr(), tuple(enc.shape), enc.dtype):
            chain = _chain
            chain.n += 1
            chain.left = chain.n
        ctx.chain = chain
        ctx.keep = (x, enc, self_pad, cross_pad, a16, a32, params)  # (parameters: kept alive, the optimizer runs after backward)
        ctx.dims = (B, T, S, d, f, H)
        off = (y_ptr - a16.data_ptr()) // es
        return a16[off: off + rows * d].view(B, T, d)

    @staticmethod
    def backward(ctx, dy):
        L = lib.load()
        lay, layer, is_decoder = ctx.lay, ctx.layer, ctx.is_decoder
        x, enc, _, _, a16, a32, params = ctx.keep
        B, T, S, d, f, H = ctx.dims
        dt, dev = x.dtype, x.device
        link_in, link_out = ctx.links
        handed = link_out.take(dy) if link_out is not None else None  # (dy through this layer's feed-forward mask, or None)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(x)
        masked = torch.empty_like(x) if link_in is not None else None
        lay.dy_masked = handed.data_ptr() if handed is not None else None
        lay.dx_masked, lay.dx_mask_offset = (masked.data_ptr(), link_in.offset) if masked is not None else (None, 0)
        denc = denc_ret = None
        lay.denc_prev = None
        if is_decoder:
            chain = ctx.chain
            if chain is not None and chain.n > 1:
                if chain.left == chain.n:                         # the first of the chain in this backward pass
                    chain.buf = torch.empty_like(enc)             # ... writes the buffer,
                    torch.autograd.Variable._execution_engine.queue_callback(chain.check)
                else:
                    lay.denc_prev = chain.buf.data_ptr()          # the others add theirs to it, in place,
                denc = chain.buf
                chain.left -= 1
                if chain.left == 0:                               # and the last one hands the sum to autograd
                    denc_ret, chain.buf, chain.left = denc, None, chain.n
            else:
                denc = denc_ret = torch.empty_like(enc)
        lay.dy, lay.dx, lay.denc = dy.data_ptr(), dx.data_ptr(), (denc.data_ptr() if is_decoder else None)
        nblk = 2 if is_decoder else 1
        wd, w2, vec, wrows, vsz = grad_buffers(layer, params, is_decoder, d, f, dt, dev)
        es = 2
        wp, vp = wd.data_ptr(), vec.data_ptr()

        def attn_grads(blk, k):
            blk.dw_in = wp + (4 * k) * d * d * es
            blk.dw_o = wp + (4 * k + 3) * d * d * es
            base = vp + 6 * k * d * es
            blk.db_in, blk.db_o, blk.dln_g, blk.dln_b = base, base + 3 * d * es, base + 4 * d * es, base + 5 * d * es

        attn_grads(lay.self_, 0)
        if is_decoder:
            attn_grads(lay.cross, 1)
        fb = lay.ffn
        fb.dw1 = wp + 4 * nblk * d * d * es
        fb.dw2 = w2.data_ptr()
        base = vp + 6 * nblk * d * es
        fb.db1, fb.db2, fb.dln_g, fb.dln_b = base, base + f * es, base + (f + d) * es, base + (f + 2 * d) * es
        key = (is_decoder, lay.fused_tail, lay.prenorm, lay.act, bool(fb.bits), B, T, S, d, f, H, lay.dtype, lay.drop_p > 0)
        sizes = _sizes.get(key)
        if sizes is None:
            sb, wb = ctypes.c_size_t(), ctypes.c_size_t()
            check(L.pk_layer_bwd_sizes(ctypes.byref(lay), ctypes.byref(sb), ctypes.byref(wb)), 'pk_layer_bwd_sizes')
            sizes = _sizes[key] = (sb.value, wb.value)
        # (asked with the gradient pointers in place: which bias sums exist is part of the plan)
        # gradient temporaries and workspaces: grow-only buffers shared by all layers (used in stream order)
        scratch = lib.workspace(sizes[0], dev, 'layer_scratch')
        ws = lib.workspace(sizes[1], dev, 'layer_ws')
        lay.scratch, lay.scratch_bytes, lay.ws, lay.ws_bytes = scratch.data_ptr(), scratch.numel(), ws.data_ptr(), ws.numel()
        lay.stream = lib.stream_ptr()
        check(L.pk_layer_bwd(ctypes.byref(lay)), 'pk_layer_bwd')
        if masked is not None:
            link_in.offer(masked, dx)
        grads = grads_in_param_order(wd, w2, vec, wrows, vsz, nblk, params)
        return (dx, denc_ret, None, None, None, None, None, *grads)


def run(layer, x, enc, self_pad, cross_pad, is_decoder: bool):
    if self_pad is not None and not self_pad.is_contiguous():
        self_pad = self_pad.contiguous()
    if cross_pad is not None and not cross_pad.is_contiguous():
        cross_pad = cross_pad.contiguous()
    link_out = None
    if layer.prenorm and layer.training and layer.dropout.p > 0 and torch.is_grad_enabled() and not _NO_DROP_LINK:
        from .autograd import DropLink
        link_out = DropLink()
    y = NativeLayerFn.apply(x, enc if is_decoder else None, self_pad, cross_pad, layer, is_decoder,
                            (getattr(x, '_pk_drop_link', None) if link_out is not None else None, link_out),
                            *layer_params(layer, is_decoder))
    if link_out is not None and link_out.p > 0:
        y._pk_drop_link = link_out
    return y
