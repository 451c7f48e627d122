"""Incremental decoding fast path: after the first decoder call of a sentence, every further call with
ONE new token per sentence (`decoder(encoder_out, encoder_mask, tokens[:, t:t+1], state=state)`, pasero/decoding.py:
1119-1221) runs as one native call, `pk_decoder_step` (include/pasero_hip.h), instead of ~90 Python-level op dispatches.

What changes relative to the reference's incremental mode (pasero/models/modules.py:621-641) — the results do not:
  * self-attention K/V live in preallocated (B, cap, D) buffers per layer, the new row is appended in place (the
    reference concatenates a new tensor every step); `state['dec_i_self_attn_key'/'_value']` stay what the reference
    documents — (B, t, H, hd) tensors — as views of those buffers;
  * `k_proj/v_proj(encoder_out)` are computed once per sentence (the reference recomputes them at every step).
The caches ride in the `state` dict under '_pk_decode' (`DecodeCache`); `Decoder.reorder_state` reorders them for beam
search.  Layers the native step does not cover (subclass hooks, adapters, gated feed-forward, rotary positions) keep
using the per-op path: `build_plan` returns None for them.
"""
import ctypes
import os
import weakref
from typing import Optional

import torch
from torch import Tensor

from . import functional as F
from . import lib
from .lib import ACT, check, dtype_code, ptr, stream_ptr

_P = ctypes.c_void_p
_LAYER_FIELDS = ['qkv_w', 'qkv_b', 'out_w', 'out_b', 'ln1_g', 'ln1_b', 'cq_w', 'cq_b', 'cout_w', 'cout_b', 'ln2_g',
                 'ln2_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'ln3_g', 'ln3_b']


class PkDecoderLayerWeights(ctypes.Structure):
    _fields_ = [(n, _P) for n in _LAYER_FIELDS]


class PkDecoderPlan(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ('n_layers', 'd', 'heads', 'ffn', 'act', 'prenorm', 'dtype', 'scaled_attn')]
                + [('vocab', ctypes.c_longlong), ('eps', ctypes.c_float), ('embed_scale', ctypes.c_float)]
                + [(n, _P) for n in ('embed', 'pos', 'embed_ln_g', 'embed_ln_b', 'final_ln_g', 'final_ln_b', 'out_w')]
                + [('layers', ctypes.POINTER(PkDecoderLayerWeights))])


class Plan:
    """the PkDecoderPlan of one TransformerDecoder plus what keeps its pointers alive / detects that they went stale"""

    def __init__(self, c_plan, c_layers, keep, fingerprint, cross_w):
        self.c_plan, self.c_layers, self.keep, self.fingerprint, self.cross_w = c_plan, c_layers, keep, fingerprint, cross_w
        self.full = None


def _fingerprint(decoder):
    w = decoder.embed_tokens.weight
    return (w.data_ptr(), w.dtype, str(w.device), decoder.layers[0].fc1.weight.data_ptr())


def _full_fingerprint(decoder):
    """every parameter's address and type: the plan holds raw pointers, so a parameter whose storage was replaced
    (`p.data = ...`, a partial `.to()`) must invalidate it.  ~30 us: checked once per sentence batch, the cheap
    fingerprint above at every step."""
    return tuple((p.data_ptr(), p.dtype) for p in decoder.parameters())


def build_plan(decoder) -> Optional[Plan]:
    """PkDecoderPlan for `decoder` (a pasero_amd TransformerDecoder), or None when a layer is not the stock one"""
    from . import modules
    from .transformer import TransformerDecoderLayer
    cfg = decoder.cfg
    if (decoder.training or not decoder.embed_tokens.weight.is_cuda
            or cfg.embed_dim != 64 * cfg.decoder_attention_heads):  # the native step is built for head_dim 64
        return None
    if getattr(decoder.embed_tokens, 'frozen_embedding', None) is not None:
        # a partially frozen table (modules.py:929-946: lookup and tied projection blend `weight` with
        # `frozen_embedding.weight`; with shared embeddings the decoder carries the encoder's): the plan holds ONE raw
        # table pointer, so these decoders stay on the per-op path, which reads `effective_weight()`
        return None
    pos = decoder.embed_positions
    if not isinstance(pos, (modules.SinusoidalPositionalEmbedding, modules.LearnedPositionalEmbedding)):
        return None  # rotary: positions are applied inside attention — per-op path
    hooks = ('ffn', 'self_attention', 'cross_attention', 'self_attn_residual', 'self_attn_prenorm',
             'self_attn_postnorm', 'cross_attn_residual', 'cross_attn_prenorm', 'cross_attn_postnorm', 'ffn_residual',
             'ffn_prenorm', 'ffn_postnorm', 'forward')

    def ln(m):
        return m if type(m) in (modules.LayerNorm, modules.WrappableLayerNorm, modules.LayerNormWithoutBias) else None

    keep, layers = [], []
    for layer in decoder.layers:
        if not isinstance(layer, TransformerDecoderLayer) or not layer._hooks_are_base(*hooks):
            return None
        if (getattr(layer, '_no_ckpt_forward', None) is not None or layer.fc3 is not None or layer.fc1.lora is not None
                or layer.self_attn.q_proj.lora is not None):
            return None
        norms = [ln(layer.self_attn_layer_norm), ln(layer.encoder_attn_layer_norm), ln(layer._norm_module(layer.final_layer_norm))]
        sa, ca = layer.self_attn, layer.encoder_attn
        if (None in norms or sa.rotary_embed is not None or ca.rotary_embed is not None or layer.prenorm != cfg.decoder_prenorm
                or layer.activation_fn.name not in ('relu', 'gelu', 'gelu_tanh', 'silu')):
            return None
        w_sa, b_sa = sa._flat()
        w_ca, b_ca = ca._flat()
        D = cfg.embed_dim
        keep += [w_sa, b_sa, w_ca, b_ca]
        vals = [w_sa, b_sa, sa.out_proj.weight, sa.out_proj.bias, norms[0].weight, norms[0].bias,
                w_ca[:D], None if b_ca is None else b_ca[:D], ca.out_proj.weight, ca.out_proj.bias, norms[1].weight,
                norms[1].bias, layer.fc1.weight, layer.fc1.bias, layer.fc2.weight, layer.fc2.bias, norms[2].weight,
                norms[2].bias]
        layers.append((vals, (w_ca[D:], None if b_ca is None else b_ca[D:])))
    embed_ln = decoder.layernorm_embedding
    final_ln = decoder.layer_norm
    for m in (embed_ln, final_ln):
        if not isinstance(m, modules.Identity) and ln(m) is None:
            return None
    c_layers = (PkDecoderLayerWeights * len(layers))()
    for cl, (vals, _) in zip(c_layers, layers):
        for name, v in zip(_LAYER_FIELDS, vals):
            setattr(cl, name, ptr(v))
    E = decoder.embed_tokens.weight
    out_w = E if decoder.output_projection is None else decoder.output_projection.weight
    table = pos.table()
    keep += [table]
    p = PkDecoderPlan()
    p.n_layers, p.d, p.heads, p.ffn = len(layers), cfg.embed_dim, cfg.decoder_attention_heads, cfg.decoder_ffn_dim
    p.act = ACT[decoder.layers[0].activation_fn.name]
    p.prenorm, p.dtype, p.scaled_attn = int(bool(cfg.decoder_prenorm)), dtype_code(E), int(bool(cfg.scale_attn))
    p.vocab, p.eps, p.embed_scale = E.size(0), float(cfg.norm_eps), float(decoder.embed_scale)
    p.embed, p.pos = ptr(E), ptr(table)
    p.embed_ln_g = None if isinstance(embed_ln, modules.Identity) else ptr(embed_ln.weight)
    p.embed_ln_b = None if isinstance(embed_ln, modules.Identity) else ptr(embed_ln.bias)
    p.final_ln_g = None if isinstance(final_ln, modules.Identity) else ptr(final_ln.weight)
    p.final_ln_b = None if isinstance(final_ln, modules.Identity) else ptr(final_ln.bias)
    p.out_w = ptr(out_w)
    p.layers = ctypes.cast(c_layers, ctypes.POINTER(PkDecoderLayerWeights))
    keep += [t for vals, _ in layers for t in vals if t is not None] + [E, out_w]  # the pointers stay valid
    keep += [t for m in (embed_ln, final_ln) if not isinstance(m, modules.Identity) for t in (m.weight, m.bias)]
    dtypes = {t.dtype for t in keep if t is not None and t.is_floating_point()}
    if len(dtypes) != 1:
        return None  # mixed parameter dtypes (e.g. fp32 LayerNorm weights in a bf16 model): per-op path, which refuses loudly
    plan = Plan(p, c_layers, keep, _fingerprint(decoder), [cw for _, cw in layers])
    plan.full = _full_fingerprint(decoder)
    return plan


_plans = weakref.WeakKeyDictionary()  # decoder -> (fingerprint, training), Plan | None


def get_plan(decoder) -> Optional[Plan]:
    """the decoder's plan, rebuilt when its parameters moved (`.to()`), changed dtype or it switched train/eval"""
    key = (_fingerprint(decoder), decoder.training)
    cached = _plans.get(decoder)
    if cached is None or cached[0] != key:
        cached = (key, build_plan(decoder))
        _plans[decoder] = cached
    return cached[1]


class DecodeCache:
    """per-sentence-batch decoding state of the native step: self-attention caches, cross-attention K/V, scratch"""

    KEY = '_pk_decode'
    OFF = '_pk_decode_off'  # set once a sentence batch left the native path for good (cache truncation)

    def __init__(self, decoder, plan: Plan, encoder_out: Tensor, encoder_mask: Optional[Tensor], state: dict):
        cfg = decoder.cfg
        self.plan = plan
        self.names = [(f'{layer.self_attn_key}_key', f'{layer.self_attn_key}_value') for layer in decoder.layers]
        first = state[self.names[0][0]]
        B, t = first.size(0), first.size(1)
        self.B, self.t, self.D, self.H = B, t, cfg.embed_dim, cfg.decoder_attention_heads
        self.S = encoder_out.size(1)
        dev, dt = encoder_out.device, encoder_out.dtype
        self.cap = max(64, 1 << (2 * t).bit_length())
        self.k = [torch.empty(B, self.cap, self.D, device=dev, dtype=dt) for _ in self.names]
        self.v = [torch.empty(B, self.cap, self.D, device=dev, dtype=dt) for _ in self.names]
        for (kn, vn), ck, cv in zip(self.names, self.k, self.v):
            ck[:, :t].copy_(state[kn].reshape(B, t, self.D))
            cv[:, :t].copy_(state[vn].reshape(B, t, self.D))
        enc2d = encoder_out.reshape(B * self.S, self.D)
        self.cross = [F.gemm(enc2d, w, bias=b) for w, b in plan.cross_w]  # [k | v] projections, once per sentence
        self.mask = None if encoder_mask is None else encoder_mask.contiguous()
        self.V = decoder.embed_tokens.weight.size(0)
        self.ldl = (self.V + 7) // 8 * 8
        nbytes = lib.load().pk_decoder_step_scratch(ctypes.byref(plan.c_plan), B)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._tables()
        self._publish(state)

    def _tables(self):
        n = len(self.names)
        self.k_tab = (_P * n)(*[t.data_ptr() for t in self.k])
        self.v_tab = (_P * n)(*[t.data_ptr() for t in self.v])
        self.c_tab = (_P * n)(*[t.data_ptr() for t in self.cross])

    def _publish(self, state: dict):
        """the reference's view of the caches: (B, t, H, hd) tensors per layer"""
        B, t, H = self.B, self.t, self.H
        for (kn, vn), ck, cv in zip(self.names, self.k, self.v):
            state[kn] = ck[:, :t].view(B, t, H, 64)
            state[vn] = cv[:, :t].view(B, t, H, 64)

    def matches(self, state: dict, encoder_out: Tensor) -> bool:
        """still the caches of THIS state (nobody replaced the tensors behind our back)"""
        first = state.get(self.names[0][0])
        return (first is not None and first.data_ptr() == self.k[0].data_ptr() and first.size(1) == self.t
                and first.size(0) == self.B and encoder_out.size(0) == self.B and encoder_out.size(1) == self.S)

    def _grow(self):
        self.cap *= 2
        for lst in (self.k, self.v):
            for i, old in enumerate(lst):
                new = torch.empty(self.B, self.cap, self.D, device=old.device, dtype=old.dtype)
                new[:, :self.t].copy_(old[:, :self.t])
                lst[i] = new
        self._tables()

    def step(self, ids: Tensor, pos_start: int, state: dict) -> Tensor:
        if self.t + 1 > self.cap:
            self._grow()
        ids = ids.reshape(-1).contiguous()
        logits = torch.empty(self.B, self.ldl, device=self.scratch.device, dtype=self.k[0].dtype)  # the caller keeps it
        L = lib.load()
        check(L.pk_decoder_step(ctypes.byref(self.plan.c_plan), ptr(ids), self.B, self.t, int(pos_start), self.k_tab,
                                self.v_tab, self.cap, self.c_tab, ptr(self.mask), self.S, ptr(self.scratch),
                                self.scratch.numel(), ptr(logits), self.ldl, stream_ptr()), 'pk_decoder_step')
        self.t += 1
        self._publish(state)
        return logits[:, :self.V].unsqueeze(1)

    def reorder(self, indices: Tensor, state: dict):
        """beam search (transformer.py:68-75): select sentences / hypotheses along the batch dimension"""
        idx = indices.to(self.k[0].device)
        self.B = idx.numel()
        self.k = [c.index_select(0, idx) for c in self.k]
        self.v = [c.index_select(0, idx) for c in self.v]
        self.cross = [c.view(-1, self.S, 2 * self.D).index_select(0, idx).view(-1, 2 * self.D) for c in self.cross]
        self.mask = None if self.mask is None else self.mask.index_select(0, idx)
        nbytes = lib.load().pk_decoder_step_scratch(ctypes.byref(self.plan.c_plan), self.B)
        if nbytes > self.scratch.numel():
            self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.scratch.device)
        self._tables()
        self._publish(state)


def try_step(decoder, encoder_out: Tensor, encoder_mask: Optional[Tensor], decoder_input: Tensor, state: Optional[dict],
             return_layers, project: bool) -> Optional[Tensor]:
    """logits (B, 1, V) of one native decoding step, or None when the call is not one (first call of a sentence,
    several new tokens, training, return_layers, non-stock layers)"""
    if (state is None or decoder.training or torch.is_grad_enabled() or return_layers or not project
            or decoder_input.dim() != 2 or decoder_input.size(1) != 1 or not state.get('offset')
            or state.get(DecodeCache.OFF) or os.environ.get('PASERO_NO_NATIVE_DECODE')):
        return None
    plan = get_plan(decoder)
    if plan is None:
        return None
    cache = state.get(DecodeCache.KEY)
    if cache is not None and cache.plan is not plan:
        return None  # a state dict shared by several decoders (decoding.py EnsembleDecoder): per-op path for the others
    if cache is not None and not cache.matches(state, encoder_out):
        cache = None
    first = decoder.layers[0].self_attn_key + '_key'
    if cache is None:
        if first not in state or encoder_out is None or state[first].size(1) != state['offset']:
            return None
        if plan.full != _full_fingerprint(decoder):  # some parameter moved since the plan was built
            _plans.pop(decoder, None)
            plan = get_plan(decoder)
            if plan is None:
                return None
        cache = DecodeCache(decoder, plan, encoder_out, encoder_mask, state)
        state[DecodeCache.KEY] = cache
    max_len = decoder.layers[0].self_attn.max_len
    if max_len is not None and cache.t + 1 > max_len:
        # the reference starts dropping the oldest keys here (modules.py:629-634): per-op path from now on
        state.pop(DecodeCache.KEY, None)
        state[DecodeCache.OFF] = True
        return None
    pos = decoder.embed_positions
    offset = state['offset']
    pos.check_length(1, offset)
    logits = cache.step(decoder_input, pos.shift + offset, state)
    state['offset'] = offset + 1
    return logits
