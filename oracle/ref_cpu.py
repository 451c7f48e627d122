"""CPU oracle: a from-scratch restatement of Pasero's Transformer encoder-decoder training path.

TEST INFRASTRUCTURE ONLY.  Nothing under `pasero_amd/` may import this file; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only as the checker / CPU
baseline, never as the thing measured or shipped.

Parity status: PINNED.  Every function below is checked against golden vectors produced by running the
real reference (naver/pasero @ /root/reference, PyTorch-CPU fp32) in `oracle/make_golden.py`; see
`tests/test_oracle_golden.py`.  The reference itself ships no tests or fixtures.

The restatement is functional: parameters come as a flat `{name: tensor}` dict using the reference's
checkpoint names (`encoder.layers.0.self_attn.q_proj.weight`, ...), hyper-parameters as any object with
the attributes of `pasero.config.TransformerConfig` (config.py:1054-1299).  The arithmetic is spelled out
with matmul / exp / sum (no `F.scaled_dot_product_attention`, `F.cross_entropy`, `F.layer_norm`,
`nn.Conv1d`), gradients come from autograd over that explicit arithmetic.

All file:line citations are into /root/reference/pasero/.
"""
import math
from typing import Optional

import numpy as np
import torch
from torch import Tensor

LN2 = math.log(2)

# ------------------------------------------------------------------------------------------------------------
# 16-bit activation model (tests only: brackets the error of the 16-bit HIP kernels, tests/test_fullsize_gpu.py)
# ------------------------------------------------------------------------------------------------------------
# The reference trains on the GPU with the model cast wholesale to fp16 / bf16 (training.py:150-151): every module's
# output is a 16-bit tensor computed with fp32 accumulation inside the op, and so is every gradient that flows between
# ops.  With ACT_DTYPE set, the functions below round their result to that dtype at exactly those module boundaries —
# Embedding, Linear (after the bias), activation, the attention probabilities before P.V and the attention output
# (what a flash kernel does), residual additions, LayerNorm / RMSNorm outputs, the logits before `.float()`
# (transformer.py:355) — and the gradient arriving at each of them on the way back.  The arithmetic in between stays fp32.
# This variant is a MODEL of the reference's 16-bit arithmetic, not pinned by a fixture (the reference's GPU path cannot
# run without a GPU); it is only ever compared with the pinned fp32 oracle, to say how far 16-bit storage alone moves a
# result.  ACT_DTYPE = None (default): no rounding anywhere, the pinned fp32 restatement.
ACT_DTYPE = None


class _Round16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def r16(x):
    if ACT_DTYPE is None or not torch.is_tensor(x) or not x.is_floating_point():
        return x
    return _Round16.apply(x, ACT_DTYPE)


# ------------------------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------------------------
def len_to_mask(lengths: Tensor, size: int) -> Tensor:
    """utils.py:258-268 — True at padding positions"""
    return torch.arange(size)[None, :] >= lengths[:, None]


def sinusoidal_table(num_embeddings: int, dim: int, shift: int = 2) -> Tensor:
    """models/modules.py:415-430 — fairseq-style table [sin | cos] halves, `num_embeddings + shift` rows"""
    n = num_embeddings + shift
    half = dim // 2
    w = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -w)
    ang = torch.arange(n, dtype=torch.float)[:, None] * freq[None, :]
    table = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        table = torch.cat([table, torch.zeros(n, 1)], dim=1)
    return table


def positional_embedding(P: dict, prefix: str, kind: str, length: int, dim: int, max_len: int, shift: int,
                         offset: int = 0):
    """models/modules.py:394-404,435-457,467-484 — rows `shift + offset .. shift + offset + length - 1`"""
    if kind == 'sinusoidal':
        table = sinusoidal_table(max_len, dim, shift)
    elif kind == 'learned':
        table = P[prefix + '.weight']
    else:  # rotary / alibi / t5: DummyPositionalEmbedding returns 0.0 (modules.py:407-412)
        return 0.0
    assert length + shift - 1 + offset < table.size(0), 'input sequence is too long'
    return table[shift + offset: shift + offset + length][None]  # 1 x T x D


def layer_norm(x: Tensor, weight: Tensor, bias: Optional[Tensor], eps: float) -> Tensor:
    """nn.LayerNorm as used at models/transformer.py:941-947 — biased variance, eps inside the sqrt"""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    y = xc * torch.rsqrt(var + eps) * weight
    return r16(y if bias is None else y + bias)


def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """models/modules.py:192-202 RMSNorm.forward — no centring, fp32 arithmetic, result cast back to x's dtype"""
    xf = x.float()
    y = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps) * weight
    return r16(y.to(x.dtype))


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """models/modules.py:92-96 (no LoRA): y = x Wᵀ + b"""
    y = x @ weight.t()
    return r16(y if bias is None else y + bias)


def activation(name: str, x: Tensor) -> Tensor:
    """models/modules.py:220-228"""
    if name in ('gelu_tanh', 'geglu'):
        return r16(0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3))))
    if name == 'swiglu':
        return r16(x * torch.sigmoid(x))
    if name == 'gelu':
        return r16(0.5 * x * (1 + torch.erf(x / math.sqrt(2))))
    return torch.clamp(x, min=0)


def attention_core(q: Tensor, k: Tensor, v: Tensor, key_pad: Optional[Tensor], causal: bool, scale: float):
    """models/modules.py:654-677,707-720,742-771.  q (B,T,H,hd), k/v (B,S,H,hd), key_pad (B,S) bool.
    Returns (out (B,T,H,hd), weights (B,T,H,S)).  Masked scores get -inf; a fully masked row gives NaN
    softmax which the reference's custom path turns into zeros (nan_to_num, modules.py:765)."""
    B, T, H, hd = q.shape
    S = k.size(1)
    scores = torch.einsum('bthd,bshd->bhts', q, k) * scale
    if key_pad is not None:
        scores = scores.masked_fill(key_pad[:, None, None, :], -float('inf'))
    if causal:
        # rows are the LAST T positions of an S-long sequence (modules.py:671-673)
        cm = torch.ones(S, S, dtype=torch.bool).triu(1)[S - T:]
        scores = scores.masked_fill(cm[None, None], -float('inf'))
    m = scores.max(-1, keepdim=True).values
    e = torch.exp(scores - m)
    w = e / e.sum(-1, keepdim=True)
    w = torch.nan_to_num(w)
    out = r16(torch.einsum('bhts,bshd->bthd', r16(w), v))
    return out, w.permute(0, 2, 1, 3)


def rotary(q: Tensor, k: Tensor, offset: int, base: float = 10000.0):
    """models/modules.py:950-1025: q (B,T,H,hd), k (B,S,H,hd) with S == T; halves convention rotate(x) = cat(-x2, x1)"""
    hd = q.size(-1)
    T = q.size(1)
    inv_freq = 1.0 / (base ** (torch.arange(0, hd, 2).float() / hd))
    ang = (torch.arange(offset, offset + T, dtype=torch.float)[:, None] * inv_freq[None, :])  # T x hd/2
    cos = torch.cat([ang.cos(), ang.cos()], dim=-1)[None, :, None, :]
    sin = torch.cat([ang.sin(), ang.sin()], dim=-1)[None, :, None, :]

    def rot(x):
        return torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], dim=-1)
    return q * cos + rot(q) * sin, k * cos + rot(k) * sin


def multihead_attention(P: dict, prefix: str, query: Tensor, key: Tensor, value: Tensor, num_heads: int,
                        attn_mask: Optional[Tensor] = None, causal: bool = False,
                        state: Optional[dict] = None, scaled: bool = True, rope_base: Optional[float] = None):
    """models/modules.py:579-739 (MHA, kv_heads == num_heads, optional rotary, no alibi/t5, no LoRA)"""
    if attn_mask is not None and attn_mask.dim() == 2 and causal:
        attn_mask = None  # modules.py:602-605: padding mask is dropped for causal attention
    B, T, D = query.shape
    hd = D // num_heads

    def proj(name, x):
        return linear(x, P[f'{prefix}.{name}.weight'], P.get(f'{prefix}.{name}.bias'))

    q = proj('q_proj', query).view(B, T, num_heads, hd)
    k = proj('k_proj', key).view(B, -1, num_heads, hd)
    v = proj('v_proj', value).view(B, -1, num_heads, hd)
    if rope_base is not None:  # modules.py:621-623
        q, k = rotary(q, k, state['key'].size(1) if state and 'key' in state else 0, rope_base)
    if state is not None and 'key' in state:  # modules.py:625-637
        k = torch.cat([state['key'], k], dim=1)
        v = torch.cat([state['value'], v], dim=1)
    if state is not None:  # modules.py:639-641
        state['key'] = k
        state['value'] = v
    scale = 1.0 / math.sqrt(hd) if scaled else 1.0
    # modules.py:688: `is_causal = causal and tgt_len > 1`; with T == 1 the single query row is the last
    # position and sees every key, so the causal mask is a no-op there.
    out, w = attention_core(q, k, v, attn_mask, causal and T > 1, scale)
    out = out.reshape(B, T, D)
    return proj('out_proj', out), w


def label_smoothed_ce(logits: Tensor, target: Tensor, pad: int, eps: float):
    """models/transformer.py:354-380.  logits (N,V) fp32, target (N,).
    loss = sum_{i: y_i != pad} [(1-eps) * (-log p_i[y_i]) + eps/V * sum_c (-log p_i[c])]  (sum reduction)
    Returns (loss, nll_loss, num_tokens) as tensors."""
    logits = logits.float()
    N, V = logits.shape
    m = logits.max(-1, keepdim=True).values
    lse = (m + torch.log(torch.exp(logits - m).sum(-1, keepdim=True))).squeeze(-1)
    keep = target != pad
    tgt = target.clamp(min=0)
    nll_i = lse - logits.gather(1, tgt[:, None]).squeeze(1)
    smooth_i = lse - logits.sum(-1) / V
    zero = torch.zeros((), dtype=logits.dtype)
    nll = torch.where(keep, nll_i, zero).sum()
    smooth = torch.where(keep, smooth_i, zero).sum()
    loss = (1 - eps) * nll + eps * smooth if eps else nll
    return loss, nll, keep.sum()


def conv1d(x: Tensor, weight: Tensor, bias: Tensor, stride: int, padding: int) -> Tensor:
    """nn.Conv1d (models/modules.py:793-799) on channels-first x (B, C, L): explicit unfold + matmul"""
    B, C, L = x.shape
    O, _, K = weight.shape
    xp = torch.zeros(B, C, L + 2 * padding, dtype=x.dtype)
    xp[:, :, padding:padding + L] = x
    Lout = (L + 2 * padding - K) // stride + 1
    idx = torch.arange(Lout)[:, None] * stride + torch.arange(K)[None, :]  # Lout x K
    cols = xp[:, :, idx]  # B x C x Lout x K
    y = torch.einsum('bclk,ock->bol', cols, weight)
    return r16(y + bias[None, :, None])


def conv_new_length(length: Tensor, kernel_sizes, strides) -> Tensor:
    """models/modules.py:804-817"""
    for k, s in zip(kernel_sizes, strides):
        length = 1 + torch.div(length - k + 2 * (k // 2), s, rounding_mode='floor')
    return length


def conv_subsampler(P: dict, prefix: str, x: Tensor, length: Tensor, kernel_sizes, strides, act: str):
    """models/modules.py:819-834: (B,S,D) -> (B,S',D'); GLU over channels or GELU(erf) after every conv"""
    strides = strides or [2] * len(kernel_sizes)
    x = x.transpose(1, 2)
    for i, (k, s) in enumerate(zip(kernel_sizes, strides)):
        x = conv1d(x, P[f'{prefix}.conv_layers.{i}.weight'], P[f'{prefix}.conv_layers.{i}.bias'], s, k // 2)
        if act == 'glu':
            a, b = x.chunk(2, dim=1)
            x = r16(a * torch.sigmoid(b))
        else:
            x = activation('gelu', x)
    return x.transpose(1, 2), conv_new_length(length, kernel_sizes, strides)


# ------------------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------------------
def _rope(cfg, kind):
    return float(getattr(cfg, 'rope_base', 10000)) if kind == 'rotary' else None


def _scaled(cfg) -> bool:
    return bool(getattr(cfg, 'scale_attn', True))  # modules.py:654: scores are divided by sqrt(head_dim) unless disabled


def _final_norm(cfg, prefix):
    # transformer.py:977-980,1202-1205: with shared_norm the layer's last norm IS its first one
    return prefix + ('.self_attn_layer_norm' if getattr(cfg, 'shared_norm', False) else '.final_layer_norm')


def _ln(P, prefix, x, cfg):
    if getattr(cfg, 'rms_norm', False):  # models/transformer.py:941-947: RMSNorm instead of nn.LayerNorm
        return rms_norm(x, P[prefix + '.weight'], cfg.norm_eps)
    return layer_norm(x, P[prefix + '.weight'], P.get(prefix + '.bias'), cfg.norm_eps)


def _ffn(P, prefix, x, cfg):
    """models/transformer.py:999-1019 / 1224-1244 (no fc3 unless swiglu)"""
    y = activation(cfg.activation_fn, linear(x, P[prefix + '.fc1.weight'], P.get(prefix + '.fc1.bias')))
    if cfg.activation_fn in ('swiglu', 'geglu'):
        y = r16(y * linear(x, P[prefix + '.fc3.weight'], P.get(prefix + '.fc3.bias')))
    return linear(y, P[prefix + '.fc2.weight'], P.get(prefix + '.fc2.bias'))


def adapter_layer(P: dict, prefix: str, x: Tensor, scaling: float = 1.0) -> Tensor:
    """models/modules.py:248-370, AdapterLayer.forward (:338-351) as adapter_transformer configures it (adapters.py:241-249,
    276-284: LayerNorm, biases, ReLU, residual, scaling 1):  x + up(relu(down(LayerNorm(x)))) * scaling.  nn.LayerNorm with
    its default eps (1e-5), whatever cfg.norm_eps says (modules.py:311)."""
    h = layer_norm(x, P[prefix + '.layer_norm.weight'], P[prefix + '.layer_norm.bias'], 1e-5)
    h = activation('relu', linear(h, P[prefix + '.down.weight'], P.get(prefix + '.down.bias')))
    h = linear(h, P[prefix + '.up.weight'], P.get(prefix + '.up.bias'))
    return r16(x + (h * scaling if scaling != 1.0 else h))


def _adapters(P: dict, prefix: str, x: Tensor) -> Tensor:
    """adapters.py:251-263, 286-298: every adapter of `adapters_in_use`, in order, behind the layer (the oracle runs the ones
    the state holds, in the order their parameters are registered)"""
    head = prefix + '.adapters.'
    uids = []
    for k in P:
        if k.startswith(head) and k.endswith('.down.weight'):
            uids.append(k[len(head):-len('.down.weight')])
    for uid in uids:
        x = adapter_layer(P, head + uid, x)
    return x


def encoder_layer(P: dict, prefix: str, x: Tensor, pad_mask: Tensor, cfg) -> Tensor:
    """models/transformer.py:1056-1099 (dropout = identity)"""
    pre = cfg.encoder_prenorm
    res = x
    if pre:
        x = _ln(P, prefix + '.self_attn_layer_norm', x, cfg)
    x, _ = multihead_attention(P, prefix + '.self_attn', x, x, x, cfg.encoder_attention_heads, pad_mask,
                               scaled=_scaled(cfg), rope_base=_rope(cfg, cfg.encoder_positional_encoding))
    x = r16(res + x)
    if not pre:
        x = _ln(P, prefix + '.self_attn_layer_norm', x, cfg)
    res = x
    if pre:
        x = _ln(P, _final_norm(cfg, prefix), x, cfg)
    x = r16(res + _ffn(P, prefix, x, cfg))
    if not pre:
        x = _ln(P, _final_norm(cfg, prefix), x, cfg)
    return _adapters(P, prefix, x)


def decoder_layer(P: dict, prefix: str, x: Tensor, enc_out: Tensor, enc_mask: Tensor, cfg,
                  state: Optional[dict] = None, layer_id: int = 0) -> Tensor:
    """models/transformer.py:1341-1417: causal self-attn (padding mask never passed, :1375), cross-attn with
    the encoder key-padding mask, FFN"""
    pre = cfg.decoder_prenorm
    H = cfg.decoder_attention_heads
    res = x
    if pre:
        x = _ln(P, prefix + '.self_attn_layer_norm', x, cfg)
    sa_state = None
    if state is not None:  # transformer.py:1263-1289
        key = f'dec_{layer_id}_self_attn_'
        sa_state = {k[len(key):]: v for k, v in state.items() if k.startswith(key)}
    x, _ = multihead_attention(P, prefix + '.self_attn', x, x, x, H, None, causal=True, state=sa_state,
                               scaled=_scaled(cfg), rope_base=_rope(cfg, cfg.decoder_positional_encoding))
    if sa_state:
        state.update({f'dec_{layer_id}_self_attn_{k}': v for k, v in sa_state.items()})
    x = r16(res + x)
    if not pre:
        x = _ln(P, prefix + '.self_attn_layer_norm', x, cfg)
    res = x
    if pre:
        x = _ln(P, prefix + '.encoder_attn_layer_norm', x, cfg)
    x, _ = multihead_attention(P, prefix + '.encoder_attn', x, enc_out, enc_out, H, enc_mask, scaled=_scaled(cfg))
    x = r16(res + x)
    if not pre:
        x = _ln(P, prefix + '.encoder_attn_layer_norm', x, cfg)
    res = x
    if pre:
        x = _ln(P, _final_norm(cfg, prefix), x, cfg)
    x = r16(res + _ffn(P, prefix, x, cfg))
    if not pre:
        x = _ln(P, _final_norm(cfg, prefix), x, cfg)
    return _adapters(P, prefix, x)


# ------------------------------------------------------------------------------------------------------------
# encoder / decoder / model
# ------------------------------------------------------------------------------------------------------------
def _embed_scale(cfg):
    return math.sqrt(cfg.embed_dim) if cfg.scale_embed else 1.0


def encoder_max_len(cfg) -> int:
    """models/transformer.py:643-661: positions are added after the convolutions"""
    n = cfg.encoder_max_len
    if cfg.conv_kernel_sizes:
        strides = cfg.conv_strides or [2] * len(cfg.conv_kernel_sizes)
        n = int(conv_new_length(torch.tensor(n), cfg.conv_kernel_sizes, strides))
    return n


def encoder(P: dict, cfg, encoder_input: Tensor, encoder_input_length: Tensor):
    """models/transformer.py:698-752 -> (x (B,S,D), padding_mask (B,S))"""
    length = encoder_input_length
    if encoder_input.dim() == 2:
        ids = encoder_input.clamp(min=0)
        x = P['encoder.embed_tokens.weight'][ids]  # modules.py:916-933
        if 'encoder.embed_tokens.frozen_embedding.weight' in P:  # modules.py:929-933: masked tokens read the frozen table
            mask = P['encoder.embed_tokens.freeze_mask'][ids][..., None]  # (not a parameter: the task's (V,) bool mask)
            x = (~mask) * x + mask * P['encoder.embed_tokens.frozen_embedding.weight'][ids]
    else:  # speech features, transformer.py:731-737
        x = encoder_input
        if cfg.conv_kernel_sizes:
            conv_in = cfg.conv_input_dim or cfg.input_dim or cfg.embed_dim
            if conv_in != (cfg.input_dim or cfg.embed_dim):
                x = torch.clamp(linear(x, P['encoder.in_linear.0.weight'], P['encoder.in_linear.0.bias']), min=0)
            x, length = conv_subsampler(P, 'encoder.subsample', x, length, cfg.conv_kernel_sizes,
                                        cfg.conv_strides, cfg.conv_activation)
        elif (cfg.input_dim or cfg.embed_dim) != cfg.embed_dim:
            x = linear(x, P['encoder.in_linear.weight'], P['encoder.in_linear.bias'])
    x = r16(x * _embed_scale(cfg))
    S = x.size(1)
    pad_mask = len_to_mask(length, S)
    x = r16(x + positional_embedding(P, 'encoder.embed_positions', cfg.encoder_positional_encoding, S,
                                     cfg.embed_dim, encoder_max_len(cfg), cfg.positional_encoding_shift))
    if cfg.encoder_embed_norm:
        x = _ln(P, 'encoder.layernorm_embedding', x, cfg)
    for i in range(cfg.encoder_layers):
        x = encoder_layer(P, f'encoder.layers.{i}', x, pad_mask, cfg)
    if cfg.encoder_prenorm:
        x = _ln(P, 'encoder.layer_norm', x, cfg)
    return x, pad_mask


def _dec_embed_weight(P, cfg):
    # shared_embeddings: the decoder reuses the encoder's Embedding object (transformer.py:151-153) — including its frozen
    # table: lookup and tied projection blend `weight` and `frozen_embedding.weight` per token / per column
    # (modules.py:929-933, 942-946), which is the table where(mask, frozen, weight) applied to either
    pre = 'decoder' if 'decoder.embed_tokens.weight' in P else 'encoder'
    E = P[pre + '.embed_tokens.weight']
    if pre + '.embed_tokens.frozen_embedding.weight' in P:
        mask = P['encoder.embed_tokens.freeze_mask'][:, None]
        E = (~mask) * E + mask * P[pre + '.embed_tokens.frozen_embedding.weight']
    return E


def decoder(P: dict, cfg, enc_out: Tensor, enc_mask: Tensor, decoder_input: Tensor,
            state: Optional[dict] = None) -> Tensor:
    """models/transformer.py:831-898 -> logits (B,T,V)"""
    T = decoder_input.size(1)
    offset = state.get('offset', 0) if state else 0
    pos = positional_embedding(P, 'decoder.embed_positions', cfg.decoder_positional_encoding, T, cfg.embed_dim,
                               cfg.decoder_max_len, cfg.positional_encoding_shift, offset=offset)
    if state is not None:
        state['offset'] = offset + T
    E = _dec_embed_weight(P, cfg)
    x = r16(r16(E[decoder_input.clamp(min=0)] * _embed_scale(cfg)) + pos)
    if cfg.decoder_embed_norm:
        x = _ln(P, 'decoder.layernorm_embedding', x, cfg)
    for i in range(cfg.decoder_layers):
        x = decoder_layer(P, f'decoder.layers.{i}', x, enc_out, enc_mask, cfg, state=state, layer_id=i)
    if cfg.decoder_prenorm:
        x = _ln(P, 'decoder.layer_norm', x, cfg)
    if cfg.tied_output_projection:  # modules.py:935-947
        return r16(x @ E.t())
    return r16(x @ P['decoder.output_projection.weight'].t())


def transformer_forward(P: dict, cfg, encoder_input: Tensor, encoder_input_length: Tensor,
                        decoder_input: Tensor, prompt_mask: Optional[Tensor] = None, **unused):
    """models/transformer.py:227-321 + compute_loss :324-380.
    Returns (loss tensor, logs dict with loss/nll_loss in bits, num_tokens, num_lines)."""
    target = decoder_input[:, 1:]
    dec_in = decoder_input[:, :-1]
    enc_out, enc_mask = encoder(P, cfg, encoder_input, encoder_input_length)
    logits = decoder(P, cfg, enc_out, enc_mask, dec_in)

    def ce(tgt):
        loss, nll, ntok = label_smoothed_ce(logits.reshape(-1, logits.size(-1)), tgt.reshape(-1),
                                            cfg.padding_idx, cfg.label_smoothing or 0.0)
        return loss, {'loss': loss.item() / LN2, 'nll_loss': nll.item() / LN2, 'num_tokens': int(ntok),
                      'num_lines': target.size(0)}
    scale = getattr(cfg, 'prompt_loss', 1.0)
    if scale == 1.0:
        return ce(target)
    # transformer.py:283-321: generated tokens at weight 1, prompt tokens at weight `prompt_loss`
    pmask = prompt_mask[:, 1:]
    loss, logs = ce(target.masked_fill(pmask, cfg.padding_idx))
    if scale > 0:
        p_loss, p_logs = ce(target.masked_fill(~pmask, cfg.padding_idx))
        logs['prompt_nll_loss'] = p_logs['nll_loss']
        logs['loss'] = logs['loss'] + scale * p_logs['loss']
        logs['num_tokens'] += p_logs['num_tokens']
        logs['num_prompt_tokens'] = p_logs['num_tokens']
        loss = loss + scale * p_loss
    return loss, logs


def greedy_decode(P: dict, cfg, enc_out: Tensor, enc_mask: Tensor, max_output_len: int) -> Tensor:
    """decoding.py:1005-1221 restricted to greedy search from a single BOS column: incremental decoding with
    the self-attention K/V `state`, argmax (lowest index wins ties, like torch.argmax), finished rows emit pad."""
    B = enc_out.size(0)
    max_len = min(cfg.decoder_max_len, 1 + max_output_len)
    tokens = torch.full((B, max_len), cfg.padding_idx, dtype=torch.long)
    tokens[:, 0] = cfg.bos_idx
    has_eos = torch.zeros(B, dtype=torch.bool)
    state = {}
    prev = 0
    prompt_len = 1
    last = 0
    for step in range(1, max_len):
        has_eos = has_eos | (step >= prompt_len + max_output_len)
        logits = decoder(P, cfg, enc_out, enc_mask, tokens[:, prev:step], state=state)[:, -1].clone()
        pad_logit = logits[:, cfg.padding_idx].clone()
        logits[has_eos] = -float('inf')
        logits[:, cfg.padding_idx] = pad_logit
        tokens[:, step] = logits.argmax(-1)
        last = step
        has_eos = (has_eos | (tokens[:, step] == cfg.eos_idx)) & (step >= prompt_len)
        prev = step
        if bool(has_eos.all()):
            break
    return tokens[:, 1:last + 1]


# ------------------------------------------------------------------------------------------------------------
# "next" rows: optimizer
# ------------------------------------------------------------------------------------------------------------
def clip_grad_norm(grads: list, max_norm: float):
    """optimization.py:390-427 (unsharded): global L2 norm in fp32, scale by max_norm / (norm + 1e-6) if > max"""
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    if max_norm > 0:
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        grads = [g * coef for g in grads]
    return total, grads


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float, beta2: float,
              eps: float, weight_decay: float):
    """optimization.py:56-149 (fairseq Adam, fp32 state, decoupled weight decay applied to the data)"""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = torch.sqrt(v) / math.sqrt(bc2) + eps
    if weight_decay:
        p = p - weight_decay * lr * p
    p = p - (lr / bc1) * m / denom
    return p, m, v


# ------------------------------------------------------------------------------------------------------------
# K8: Whisper log-mel (third-party arithmetic: transformers.WhisperFeatureExtractor, called from
# examples/Whisper/extract-features.py:107-117).  numpy float64 restatement of its published algorithm.
# ------------------------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    mel = 3.0 * f / 200.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) * logstep, mel)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    f = 200.0 * m / 3.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filter_bank(n_freq: int = 201, n_mel: int = 80, fmin: float = 0.0, fmax: float = 8000.0,
                    sr: int = 16000) -> np.ndarray:
    """(n_freq, n_mel) triangular filters, slaney scale + slaney (area) normalisation"""
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mel + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (hz_pts[2:n_mel + 2] - hz_pts[:n_mel])
    return fb * enorm[None, :]


def log_mel(wav: np.ndarray, n_samples: int = 480000, n_fft: int = 400, hop: int = 160) -> np.ndarray:
    """wav (n,) fp32 -> (3000, 80) fp32 log-mel features.
    pad/truncate to 30 s; reflect-pad n_fft/2; periodic hann; |STFT|^2; mel; log10(clamp 1e-10); drop the last
    frame; max(x, max-8); (x+4)/4; transposed to (frames, mel) as extract-features.py:116 does."""
    x = np.zeros(n_samples, dtype=np.float64)
    n = min(len(wav), n_samples)
    x[:n] = wav[:n]
    window = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)  # periodic hann
    xp = np.pad(x, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(xp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = xp[idx] * window[None, :]
    spec = np.fft.rfft(frames, n=n_fft, axis=1)
    power = spec.real ** 2 + spec.imag ** 2  # (frames, 201)
    mel = power @ mel_filter_bank(n_fft // 2 + 1)  # (frames, 80)
    logm = np.log10(np.maximum(mel, 1e-10))[:-1]
    logm = np.maximum(logm, logm.max() - 8.0)
    return ((logm + 4.0) / 4.0).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------
def to_torch_state(np_state: dict) -> dict:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_state.items()}


def count_flops(cfg, B: int, S: int, T: int, V: int) -> float:
    """SURVEY §8d algorithmic FLOPs of one fwd+bwd step (2·MACs, bwd = 2x fwd, causal self-attn at half)"""
    d, fe, fd = cfg.embed_dim, cfg.encoder_ffn_dim, cfg.decoder_ffn_dim
    Le, Ld = cfg.encoder_layers, cfg.decoder_layers
    fwd = (B * S * Le * (8 * d * d + 4 * d * fe) + B * S * Le * 4 * S * d
           + B * T * Ld * (12 * d * d + 4 * d * fd) + B * S * Ld * 4 * d * d
           + B * T * Ld * 2 * T * d + B * T * Ld * 4 * S * d + B * T * 2 * d * V)
    return 3.0 * fwd
