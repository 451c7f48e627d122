"""Host-side mirror of `pasero/models/modules.py` for the Transformer hot path: same class names, constructor
arguments, parameter names/shapes (= checkpoint keys) and forward signatures, with every forward/backward routed to
the HIP kernels of libpasero_hip.so.  Variants that are outside the hot-path scope (tensor parallelism, ALiBi / T5
biases, grouped-query attention, sliding windows) raise NotImplementedError loudly instead of silently falling back to
eager PyTorch.
"""
import contextlib
import functools
import math
import os
import weakref
from typing import Optional, Union

import torch
import torch.nn as nn
from torch import Tensor, LongTensor, BoolTensor

from . import functional as F
from .profiling import region as _bench_region
from .autograd import (LinearFn, PackedLinearFn, AttentionFn, ResidualLayerNormFn, DropoutFn, EmbeddingFn,
                       LinearResidualLnFn, block_tail_eligible,
                       ActivationFn, GLUFn, RotaryFn, ResidualLink)

_ROPE_PASS = os.environ.get('PASERO_ROPE_PASS', '0') not in ('', '0')  # rotary positions as a pass of their own (pk_rope), for A/B


class Identity(nn.Identity):
    """nn.Identity whose forward accepts extra dummy arguments (modules.py:30-40)"""

    def __init__(self, return_tuple: bool = False):
        super().__init__()
        self.return_tuple = return_tuple

    def forward(self, input, *args, **kwargs):
        return (input, *args) if self.return_tuple else input


_fast_init = False


@contextlib.contextmanager
def fast_init(device=None, dtype=None):
    """Skip random initialisation and create parameters directly with the target dtype/device (modules.py:45-64)"""
    global _fast_init
    old_dtype, old_flag = torch.get_default_dtype(), _fast_init
    _fast_init = True
    if dtype:
        torch.set_default_dtype(dtype)
    if device is not None:
        torch.set_default_device(device)
    try:
        yield
    finally:
        _fast_init = old_flag
        torch.set_default_dtype(old_dtype)
        torch.set_default_device(None)


def set_tp_group(tp_group=None):
    """modules.py:174-176.  Tensor parallelism is out of the hot-path scope: only `None` is accepted."""
    if tp_group is not None:
        raise NotImplementedError('pasero_amd: tensor parallelism (--tp-size) is not implemented; use data parallelism')


def set_sequence_parallel(enable: bool = True):
    """modules.py:169-171 (no-op without tensor parallelism)"""
    if enable:
        raise NotImplementedError('pasero_amd: Megatron sequence parallelism is not implemented')


class Linear(nn.Linear):
    """nn.Linear running on the MFMA GEMM kernel, with the reference's optional LoRA branch (modules.py:67-100):
    output = x Wᵀ + b + (alpha / rank) · up(down(x))"""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 lora_rank: int = 0, lora_alpha: int = 1):
        super().__init__(in_features, out_features, bias, device, dtype)
        self.lora_rank, self.lora_alpha = lora_rank, lora_alpha
        self.lora = AdapterLayer.LoRA(in_features, lora_rank, out_features, lora_alpha) if lora_rank else None

    def forward(self, input: Tensor, link=None, group=None) -> Tensor:
        output = LinearFn.apply(input, self.weight, self.bias, 'none', link, group)
        if self.lora is not None:
            output = self.lora(input, residual=output)
        return output

    def reset_parameters(self) -> None:
        if not _fast_init:
            super().reset_parameters()


class WrappableLinear(Linear):
    pass


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (transformer.py:941-947) on the wave-per-row HIP kernel"""

    def forward(self, x: Tensor) -> Tensor:
        return ResidualLayerNormFn.apply(x, None, self.weight, self.bias, self.eps, 0.0)


class LayerNormWithoutBias(LayerNorm):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.bias = None


class WrappableLayerNorm(LayerNorm):
    pass


class RMSNorm(nn.Module):
    """modules.py:192-202 — x * rsqrt(mean(x^2) + eps) * weight, computed in fp32 whatever the storage type: the
    wave-per-row LayerNorm kernel with the mean fixed at 0"""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x: Tensor) -> Tensor:
        return ResidualLayerNormFn.apply(x, None, self.weight, None, self.eps, 0.0, None, True)


class WrappableRMSNorm(RMSNorm):
    pass


class Dropout(nn.Dropout):
    """nn.Dropout with an in-kernel Philox mask that the backward pass regenerates"""

    def forward(self, x: Tensor) -> Tensor:
        if not self.training or self.p == 0:
            return x
        return DropoutFn.apply(x, self.p)


class Activation(nn.Module):
    def __init__(self, name: str):
        super().__init__()
        self.name = name

    def forward(self, x: Tensor) -> Tensor:
        return ActivationFn.apply(x, self.name)


def get_activation_fn(activation_fn: str = 'relu'):
    """modules.py:220-228 — returns a module with `.name` (the fused FFN path reads it)"""
    if activation_fn in ('gelu_tanh', 'geglu'):
        return Activation('gelu_tanh')
    if activation_fn == 'swiglu':
        return Activation('silu')
    if activation_fn == 'gelu':
        return Activation('gelu')
    return Activation('relu')


# adapter activation names (the reference builds nn.ReLU / nn.GELU(approximate='tanh') / Identity, modules.py:276-283)
_ADAPTER_ACTS = {None: 'none', 'none': 'none', 'relu': 'relu', 'gelu': 'gelu_tanh'}


class AdapterLayer(nn.Module):
    """Bottleneck adapter (Bapna et al., 2019) and, without LayerNorm / biases / activation / residual, a LoRA branch —
    the checkpoint and fine-tuning contract of pasero/models/modules.py:248-370: sub-modules `layer_norm`, `down`,
    `up` (parameter names), near-zero or LoRA-style initialisation, `enable()` / `disable()`, and at inference an
    adapter missing from the checkpoint switches itself off for good while a different bottleneck size is adopted.
    The forward pass is ONE fused autograd function (AdapterFn): LayerNorm -> down + activation -> up·scaling + residual."""

    def __init__(self, input_dim: int, projection_dim: int, output_dim: Optional[int] = None, zero_init: bool = False,
                 layer_norm: bool = True, bias: bool = True, residual: bool = True, activation_fn: str = 'relu',
                 scaling: float = 1.0):
        super().__init__()
        if activation_fn not in _ADAPTER_ACTS:
            raise NotImplementedError(f'adapter activation {activation_fn!r}')
        out_dim = output_dim or input_dim
        if residual and out_dim != input_dim:
            raise AssertionError('a residual adapter keeps the width of its input')
        self.input_dim, self.projection_dim, self.output_dim = input_dim, projection_dim, out_dim
        self.zero_init, self.has_layer_norm, self.bias, self.residual = zero_init, layer_norm, bias, residual
        self.act_name, self.scaling = _ADAPTER_ACTS[activation_fn], scaling
        self.disabled = False
        self.layer_norm = self.down = self.up = None
        self._build(device=None, dtype=None)

    @classmethod
    def LoRA(cls, input_dim: int, projection_dim: int, output_dim: int, lora_alpha: int) -> 'AdapterLayer':
        """the low-rank branch of `Linear` (modules.py:288-301): x -> up(down(x)) * alpha / rank, starts at zero"""
        return cls(input_dim, projection_dim, output_dim, zero_init=True, layer_norm=False, bias=False,
                   residual=False, activation_fn=None, scaling=lora_alpha / projection_dim)

    def _build(self, device, dtype) -> None:
        """(re)create the three sub-modules for the current bottleneck size and draw their initial values"""
        kw = dict(device=device, dtype=dtype)
        self.down = nn.Linear(self.input_dim, self.projection_dim, bias=self.bias, **kw)
        self.up = nn.Linear(self.projection_dim, self.output_dim, bias=self.bias, **kw)
        self.layer_norm = nn.LayerNorm(self.input_dim, **kw) if self.has_layer_norm else Identity()
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """zero_init: LoRA's start (down ~ kaiming-uniform, up = 0: the branch is exactly zero); otherwise both
        projections within 1e-6 of zero, so a fresh adapter is (almost) the identity; biases zero (modules.py:310-325)"""
        if self.zero_init:
            nn.init.kaiming_uniform_(self.down.weight, a=math.sqrt(5))
            self.up.weight.zero_()
        else:
            for lin in (self.down, self.up):
                lin.weight.uniform_(-1e-6, 1e-6)
        if self.bias:
            self.down.bias.zero_()
            self.up.bias.zero_()

    @property
    def permanently_disabled(self) -> bool:
        return self.down is None

    def enable(self) -> None:
        self.disabled = self.permanently_disabled

    def disable(self) -> None:
        self.disabled = True

    def forward(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        """x (B, S, D) -> (B, S, D_out).  `residual` (LoRA inside Linear): the tensor the branch is added to."""
        if self.disabled:
            return x if residual is None else residual
        from .autograd import AdapterFn
        norm = self.layer_norm if self.has_layer_norm else None
        if residual is None and self.residual:
            residual = x
        return AdapterFn.apply(x, residual, getattr(norm, 'weight', None), getattr(norm, 'bias', None),
                               getattr(norm, 'eps', 1e-5), self.down.weight, self.down.bias, self.up.weight,
                               self.up.bias, self.act_name, float(self.scaling), getattr(x, '_pk_drop_link', None))

    def _load_from_state_dict(self, state_dict, prefix: str, *args, **kwargs) -> None:
        """inference-time conveniences of the reference loader (modules.py:349-370); training loads strictly"""
        if not self.training:
            saved = state_dict.get(prefix + 'down.weight')
            if saved is None:  # the checkpoint has no such adapter: it stays off
                self.disabled = True
                self.layer_norm = self.down = self.up = None
            elif saved.size(0) != self.projection_dim:  # another bottleneck size: follow the checkpoint
                like = self.down.weight
                self.projection_dim = saved.size(0)
                self._build(device=like.device, dtype=like.dtype)
                self.disabled = False
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


def remove_unused_parameters(model: nn.Module, state_dict: dict, param_regex: Optional[str] = None) -> dict:
    """modules.py:837-864: pops (and returns, on the CPU) the entries of `state_dict` the model has no use for — all
    of them, or those whose name matches `param_regex` (groups: model part, parameter-set name)"""
    import re
    known = set(model.state_dict())
    unused = {}
    for name in list(state_dict):
        if name not in known and (param_regex is None or re.match(param_regex, name)):
            unused[name] = state_dict.pop(name).cpu()
    return unused


def add_missing_parameters(model: nn.Module, state_dict: dict, param_regex: Optional[str] = None) -> None:
    """modules.py:867-887: parameters of the model that the checkpoint lacks keep their current (random) value"""
    import re
    for name, param in model.state_dict().items():
        if name not in state_dict and (param_regex is None or re.match(param_regex, name)):
            state_dict[name] = param


def checkpoint_wrapper(module: nn.Module, activate: bool = True) -> nn.Module:
    """modules.py:386-391"""
    if activate:
        from torch.utils.checkpoint import checkpoint
        from . import rng
        module._no_ckpt_forward = inner = module.forward

        def forward(*args, **kwargs):
            # Dropout masks here are functions of (seed, offset) handed out by pasero_amd.rng, not of torch's generator:
            # the recomputation in the backward pass must replay the offsets of the first run (what torch's
            # preserve_rng_state does for the reference), then put the stream back where the step had got to.
            entry = rng.get_state()

            def run(*a, **k):
                now = rng.get_state()
                rng.set_state(entry)
                try:
                    return inner(*a, **k)
                finally:  # (the recomputation may be cut short once every saved tensor has been rebuilt)
                    if now != entry:  # this was the recomputation
                        rng.set_state(now)
            return checkpoint(run, *args, use_reentrant=False, preserve_rng_state=False, **kwargs)
        module.forward = forward
    return module


# ------------------------------------------------------------------------------------------------------------
# positional embeddings
# ------------------------------------------------------------------------------------------------------------
class DummyPositionalEmbedding(nn.Module):
    """modules.py:407-412 — rotary models add no absolute positions to the embeddings"""
    shift = 0

    def __init__(self, num_embeddings: int = 0, embedding_dim: int = 0):
        super().__init__()

    def table(self):
        return None

    def check_length(self, length: int, offset: int = 0):
        pass

    def forward(self, input, offset=0):
        return 0.0


def PositionalEmbedding(type: str, num_embeddings: int, embedding_dim: int, shift: int = 2):
    """modules.py:394-404"""
    if type == 'rotary':
        return DummyPositionalEmbedding(num_embeddings, embedding_dim)
    if type in ('alibi', 't5'):
        raise NotImplementedError(f"pasero_amd: '{type}' positional encoding is outside the hot-path scope")
    if type == 'learned':
        return LearnedPositionalEmbedding(num_embeddings, embedding_dim, shift=shift)
    if type == 'sinusoidal':
        return SinusoidalPositionalEmbedding(num_embeddings, embedding_dim, shift=shift)
    raise NotImplementedError(type)


class SinusoidalPositionalEmbedding(nn.Module):
    """fairseq-style table [sin | cos], `shift` extra leading rows (modules.py:415-457).  The table is built once in
    fp32 on the host; `table()` returns it on the model's device/dtype for the embedding kernel to add in place."""

    def __init__(self, num_embeddings: int, embedding_dim: int, shift: int = 0):
        super().__init__()
        self.shift = shift
        self.embedding_dim = embedding_dim
        n = num_embeddings + shift
        half = embedding_dim // 2
        step = math.log(10000) / (half - 1)
        freq = torch.exp(torch.arange(half, dtype=torch.float) * -step)
        ang = torch.arange(n, dtype=torch.float)[:, None] * freq[None, :]
        weight = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
        if embedding_dim % 2 == 1:
            weight = torch.cat([weight, torch.zeros(n, 1)], dim=1)
        self.weight = weight
        self.register_buffer('_float_tensor', torch.FloatTensor(1))  # follows module.to(): tells device and dtype
        self._cache = None

    def table(self) -> Tensor:
        ref = self._float_tensor
        if self._cache is None or self._cache.device != ref.device or self._cache.dtype != ref.dtype:
            self._cache = self.weight.to(device=ref.device, dtype=ref.dtype).contiguous()
        return self._cache

    def check_length(self, length: int, offset: int = 0):
        assert length + self.shift - 1 + offset < self.weight.size(0), (
            f'input sequence is too long: {length}, positional embedding size: {self.weight.size(0)}')

    def forward(self, length: int, offset: Union[LongTensor, int] = 0) -> Tensor:
        if torch.is_tensor(offset):
            raise NotImplementedError('pasero_amd: per-sequence position offsets are not implemented')
        self.check_length(length, offset)
        start = self.shift + offset
        return self.table()[start:start + length][None]


class LearnedPositionalEmbedding(nn.Embedding):
    """modules.py:460-484"""

    def __init__(self, num_embeddings: int, embedding_dim: int, shift: int = 0):
        self.shift = shift
        super().__init__(num_embeddings + shift, embedding_dim)
        nn.init.normal_(self.weight, mean=0, std=embedding_dim ** -0.5)

    def table(self) -> Tensor:
        return self.weight

    def check_length(self, length: int, offset: int = 0):
        assert length + self.shift - 1 + offset < self.weight.size(0), (
            f'input sequence is too long: {length}, positional embedding size: {self.weight.size(0)}')

    def forward(self, length: int, offset: Union[LongTensor, int] = 0) -> Tensor:
        if torch.is_tensor(offset):
            raise NotImplementedError('pasero_amd: per-sequence position offsets are not implemented')
        self.check_length(length, offset)
        start = self.shift + offset
        return self.weight[start:start + length][None]


class RotaryEmbedding(nn.Module):
    """cos / sin tables of modules.py:950-975 (inv_freq = base^(-2i/dim), fp32), kept as [max_len][dim/2] because the
    two halves of the reference's `cat(freqs, freqs)` are identical; extended to the next power of two on demand."""

    def __init__(self, dim: int, base: int = 10000):
        super().__init__()
        self.dim = dim
        self.inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.build(256)
        self._dev = None

    def build(self, max_len: int):
        t = torch.arange(max_len, dtype=torch.float32)
        freqs = torch.einsum('i,j->ij', t, self.inv_freq)
        self.cos, self.sin = freqs.cos().contiguous(), freqs.sin().contiguous()
        self.max_len = max_len
        self._dev = None

    def tables(self, total_len: int, device):
        if total_len > self.max_len:
            self.build(2 ** math.ceil(math.log2(total_len)))
        if self._dev is None or self._dev[0].device != device:
            self._dev = (self.cos.to(device), self.sin.to(device))
        return self._dev


# ------------------------------------------------------------------------------------------------------------
# embeddings
# ------------------------------------------------------------------------------------------------------------
_merged_tables = weakref.WeakKeyDictionary()  # Embedding -> (key, merged table): Embedding.effective_weight


class MergeTablesFn(torch.autograd.Function):
    """E = where(mask[:, None], frozen, weight) with a backward that keeps nothing the engine frees (the mask rides on the
    context as a plain attribute), so the node may serve several graphs: see `Embedding.effective_weight`."""

    @staticmethod
    def forward(ctx, weight: Tensor, frozen: Tensor, mask: Tensor) -> Tensor:
        ctx.mask = mask
        ctx.frozen_dtype = frozen.dtype
        return torch.where(mask[:, None], frozen.to(weight.dtype), weight)

    @staticmethod
    def backward(ctx, g: Tensor):
        m = ctx.mask[:, None]
        dw = g.masked_fill(m, 0) if ctx.needs_input_grad[0] else None
        df = g.masked_fill(~m, 0).to(ctx.frozen_dtype) if ctx.needs_input_grad[1] else None
        return dw, df, None


class Embedding(nn.Embedding):
    """Token embedding + tied output projection (modules.py:890-947).

    Partially frozen embeddings (`freeze_mask` (V,) bool, from `--freeze-encoder-embed-regex`, tasks/translation.py:141-146;
    modules.py:909-913, 929-933, 942-946): rows of the mask come from a second table `frozen_embedding.weight`, the others from
    `weight`.  The reference gathers from both tables and blends per token (and, for the projection, multiplies by both and
    blends per column); here the two tables are merged ONCE per call into the effective table E = where(mask, frozen, weight)
    — a (V, d) select, bytes of one table — and every kernel of the path (fused embed, tied projection, fused vocabulary loss)
    runs on it unchanged; autograd routes dE back to the rows of either table.  As in the reference, `frozen_embedding.weight`
    is an ordinary parameter (`self.frozen_embedding.requires_grad = False` at modules.py:911 sets an attribute of the MODULE
    and freezes nothing: the table is frozen by the trainer's freeze regex, cli/train.py:235-238, or not at all)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: int,
                 freeze_mask: Optional[BoolTensor] = None):
        super().__init__(num_embeddings, embedding_dim, padding_idx)
        if not _fast_init:
            nn.init.normal_(self.weight, mean=0, std=embedding_dim ** -0.5)
            nn.init.constant_(self.weight[padding_idx], 0)
        if freeze_mask is not None:
            assert freeze_mask.dtype == torch.bool and freeze_mask.numel() == num_embeddings
            self.freeze_mask = freeze_mask
            self.frozen_embedding = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
            self.frozen_embedding.requires_grad = False  # (sic: modules.py:911)
        else:
            self.frozen_embedding = None

    def effective_weight(self) -> Tensor:
        """the table the kernels read: `weight`, or where(freeze_mask, frozen_embedding.weight, weight).

        Merged once per parameter state, not once per caller: with shared embeddings and a tied projection one training step
        reads the table three times (encoder lookup, decoder lookup, vocabulary loss) — they share ONE merge node, whose
        backward splits the summed (V, d) gradient between the two tables once.  The merged table is kept until either
        table changes (`_version`: an optimizer step, `load_state_dict`, `.to()`) or the grad mode differs."""
        if self.frozen_embedding is None:
            return self.weight
        w, f = self.weight, self.frozen_embedding.weight
        if self.freeze_mask.device != w.device:
            self.freeze_mask = self.freeze_mask.to(w.device)
        key = (w._version, f._version, w.data_ptr(), f.data_ptr(), w.dtype, torch.is_grad_enabled(),
               w.requires_grad, f.requires_grad)
        cached = _merged_tables.get(self)  # beside the module, not on it: deepcopy / pickling of the model stay as they were
        if cached is None or cached[0] != key:
            cached = (key, MergeTablesFn.apply(w, f, self.freeze_mask))
            _merged_tables[self] = cached
        return cached[1]

    def forward(self, input: LongTensor) -> Tensor:
        # the reference asserts `input.max() < V` here (a host sync per call, modules.py:924-926); the kernel clamps
        return EmbeddingFn.apply(input, self.effective_weight(), None, 1.0, 0, 0.0, self.padding_idx)

    def embed(self, input: LongTensor, pos_table: Optional[Tensor], scale: float, pos_start: int, p: float) -> Tensor:
        """fused  dropout(E[ids] * scale + positions)  (transformer.py:727-744, 866-878)"""
        return EmbeddingFn.apply(input, self.effective_weight(), pos_table, scale, pos_start, p, self.padding_idx)

    def projection(self, input: Tensor) -> Tensor:
        return LinearFn.apply(input, self.effective_weight(), None, 'none')


# ------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------
def _contiguous(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


# attributes the layers set on a module for ONE call (hand-overs between a layer and the module that runs part of its
# sub-block, per-call bookkeeping): plain Python values — nn.Module.__setattr__ would run its Parameter / Module / buffer
# checks on every one of them, ~2.5 us each and ~280 of them per C2 step
PER_CALL_ATTRS = frozenset({'_residual_link', '_wgroup', '_tail', '_ffn_link', '_ffn_group', 'return_layers',
                            'layer_outputs'})


class _PerCallAttrs:
    def __setattr__(self, name, value):
        if name in PER_CALL_ATTRS:
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)


class MultiheadAttention(_PerCallAttrs, nn.Module):
    """modules.py:487-771.  q/k/v projections live in one flat [3D, D] arena (the three nn.Parameters are views of
    it, names and shapes unchanged) so self-attention runs ONE projection GEMM that reads x once, and the attention
    kernels read q, k, v in place from the packed output."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, shard_id: int = 0, shard_count: int = 1,
                 positional_encoding: str = 'none', lora_rank: int = 0, max_len: Optional[int] = None,
                 causal: bool = False, has_bias: bool = True, kv_heads: Optional[int] = None, key_bias: bool = True,
                 sliding_window: Optional[int] = None, layer_id: int = 0, scaled: bool = True, rope_base: int = 10000,
                 alibi_max_bias: int = 8, max_qkv: Optional[float] = None, lora_alpha: int = 1):
        super().__init__()
        if shard_count != 1:
            raise NotImplementedError('pasero_amd: tensor parallelism is not implemented')
        if (kv_heads or num_heads) != num_heads:
            raise NotImplementedError('pasero_amd: grouped-query attention is not implemented')
        if positional_encoding in ('alibi', 't5'):
            raise NotImplementedError(f"pasero_amd: '{positional_encoding}' attention bias is not implemented")
        if sliding_window or max_qkv:
            raise NotImplementedError('pasero_amd: sliding-window attention / qkv clamping are not implemented')
        self.layer_id = layer_id
        self.embed_dim = embed_dim
        self.num_heads = self.kv_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        if self.head_dim not in (64, 128):
            raise NotImplementedError(f'pasero_amd: attention kernels are built for head_dim 64 and 128, got {self.head_dim}')
        self.kv_dim = self.q_dim = embed_dim
        self.scaled = scaled
        self.has_bias = has_bias
        self.sliding_window = None
        self.max_qkv = None
        lora = dict(lora_rank=lora_rank, lora_alpha=lora_alpha)
        self.k_proj = Linear(embed_dim, embed_dim, bias=has_bias and key_bias, **lora)
        self.v_proj = Linear(embed_dim, embed_dim, bias=has_bias, **lora)
        self.q_proj = Linear(embed_dim, embed_dim, bias=has_bias, **lora)
        self.out_proj = Linear(embed_dim, embed_dim, bias=has_bias, **lora)
        self.shard_count, self.shard_id = 1, 0
        self.max_len = max_len
        self.rotary_embed = self.alibi = self.t5_embed = None
        if positional_encoding == 'rotary':
            self.rotary_embed = RotaryEmbedding(self.head_dim, base=rope_base)
        if not _fast_init:
            self.reset_parameters()
        self.causal = causal
        self._w_flat = self._b_flat = None
        self._residual_link = None  # set by the owning layer for one call (see transformer._LayerBase._linked)
        self._wgroup = None         # likewise: the layer's grouped weight-gradient launch (autograd.WGradGroup)
        self._tail = None           # likewise: the post-norm block end, to be fused into out_proj (autograd.BlockTail)

    def reset_parameters(self) -> None:
        # xavier-uniform with gain 1/sqrt(2) for q, k, v (modules.py:565-576)
        for proj in (self.k_proj, self.v_proj, self.q_proj):
            nn.init.xavier_uniform_(proj.weight, gain=1 / math.sqrt(2))
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    # ---- flat q|k|v arena ----
    def _packed(self) -> bool:
        w = self._w_flat
        q, k, v = self.q_proj.weight, self.k_proj.weight, self.v_proj.weight
        if w is None or w.device != q.device or w.dtype != q.dtype:
            return False
        n = q.numel() * q.element_size()
        base = w.data_ptr()
        return q.data_ptr() == base and k.data_ptr() == base + n and v.data_ptr() == base + 2 * n

    @torch.no_grad()
    def _pack(self) -> None:
        D = self.embed_dim
        q, k, v = self.q_proj, self.k_proj, self.v_proj
        w = torch.empty(3 * D, D, dtype=q.weight.dtype, device=q.weight.device)
        for i, proj in enumerate((q, k, v)):
            w[i * D:(i + 1) * D].copy_(proj.weight)
            proj.weight.data = w[i * D:(i + 1) * D]
        self._w_flat = w
        self._b_flat = None
        if self.has_bias:
            b = torch.zeros(3 * D, dtype=q.weight.dtype, device=q.weight.device)
            for i, proj in enumerate((q, k, v)):
                if proj.bias is not None:
                    b[i * D:(i + 1) * D].copy_(proj.bias)
                    proj.bias.data = b[i * D:(i + 1) * D]
            self._b_flat = b

    def _flat(self):
        if not self._packed():
            self._pack()
        return self._w_flat, self._b_flat

    @_bench_region('attention')
    def forward(self, query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[BoolTensor] = None,
                state: Optional[dict] = None, return_attn: bool = False):
        """
        Shape: query (B,T,D), key/value (B,S,D), attn_mask (B,S) bool with True at masked keys.
        Returns (attn (B,T,D), attn_weights or None)
        """
        if attn_mask is not None and attn_mask.dim() == 2 and self.causal:
            attn_mask = None  # padding is at the end: useless for causal attention (modules.py:602-605)
        if attn_mask is not None and attn_mask.dim() != 2:
            raise NotImplementedError('pasero_amd: (B,T,S) attention masks are not implemented')
        drop = float(self.dropout) if self.training else 0.0  # attention-probability dropout (modules.py:707-720)
        B, T, D = query.shape
        H = self.num_heads
        scale = 1.0 / math.sqrt(self.head_dim) if self.scaled else 1.0
        if attn_mask is not None:
            attn_mask = attn_mask if attn_mask.is_contiguous() else attn_mask.contiguous()
        w, b = self._flat()
        link, self._residual_link = self._residual_link, None  # gradient of the residual branch rides on the dX GEMM
        group, self._wgroup = self._wgroup, None  # weight gradients ride in the layer's grouped launch
        gp = () if group is None else (group,)
        tail, self._tail = self._tail, None  # residual + dropout + LayerNorm of the block end ride on the out_proj GEMM
        q_w, k_w, v_w = self.q_proj.weight, self.k_proj.weight, self.v_proj.weight
        q_b, k_b, v_b = self.q_proj.bias, self.k_proj.bias, self.v_proj.bias

        def rope(packed, offset):  # modules.py:621-623: q and k are rotated, v is not
            if self.rotary_embed is None:
                return packed
            cos_t, sin_t = self.rotary_embed.tables(offset + T, packed.device)
            return RotaryFn.apply(packed, cos_t, sin_t, 2 * D, offset)

        def rope_in_kernel():  # the rotation folded into the attention kernels (q and k stay unrotated in memory)
            if self.rotary_embed is None or _ROPE_PASS:
                return None
            cos_t, sin_t = self.rotary_embed.tables(T, query.device)
            return (cos_t, sin_t, 0, 0)

        if self.rotary_embed is not None and not (key is query and value is query):
            raise NotImplementedError('pasero_amd: rotary embeddings are implemented for self-attention only')
        weights = None
        if self.q_proj.lora is not None or return_attn:
            # LoRA branches on the projections (modules.py:67-100) / attention weights wanted (return_layers): three
            # separate projections; the fused packed projection is for the plain layer
            q = self.q_proj(query, link=link)
            k = self.k_proj(key)
            v = self.v_proj(value)
            fold = None
            if self.rotary_embed is not None and state is None and not return_attn:
                fold = rope_in_kernel()
            if self.rotary_embed is not None and fold is None:  # modules.py:621-623: the new q and k rows, at their absolute positions
                offset = state['key'].size(1) if (state is not None and 'key' in state) else 0
                cos_t, sin_t = self.rotary_embed.tables(offset + T, q.device)
                q = RotaryFn.apply(_contiguous(q), cos_t, sin_t, D, offset)
                k = RotaryFn.apply(_contiguous(k), cos_t, sin_t, D, offset)
            if state is not None:
                k4, v4 = k.reshape(B, -1, H, self.head_dim), v.reshape(B, -1, H, self.head_dim)
                if 'key' in state:
                    prev_k, prev_v = state['key'], state['value']
                    if self.max_len is not None:
                        delta = max(0, prev_k.size(1) + k4.size(1) - self.max_len)
                        prev_k, prev_v = prev_k[:, delta:], prev_v[:, delta:]
                    k4, v4 = torch.cat([prev_k, k4], dim=1), torch.cat([prev_v, v4], dim=1)
                state['key'], state['value'] = k4, v4
                k, v = k4.reshape(B, -1, D), v4.reshape(B, -1, D)
            attn = AttentionFn.apply(q, k, v, attn_mask, H, self.causal and T > 1, scale, drop, fold)
            if return_attn:  # (B,T,H,S) softmax weights before dropout, as the reference's explicit path returns them
                with torch.no_grad():
                    weights = F.attn_probs(q.detach(), k.detach(), H, attn_mask, self.causal and T > 1, scale)
        elif state is not None:  # incremental decoding (inference): K/V cache of shape (B,S,H,hd) (modules.py:621-641)
            qkv = PackedLinearFn.apply(query, w, b, 3, None, q_w, k_w, v_w, q_b, k_b, v_b)
            qkv = rope(qkv, state['key'].size(1) if 'key' in state else 0)
            q = qkv[..., :D]
            k = qkv[..., D:2 * D].reshape(B, T, H, self.head_dim)
            v = qkv[..., 2 * D:].reshape(B, T, H, self.head_dim)
            if 'key' in state:
                prev_k, prev_v = state['key'], state['value']
                if self.max_len is not None:
                    delta = max(0, prev_k.size(1) + T - self.max_len)
                    prev_k, prev_v = prev_k[:, delta:], prev_v[:, delta:]
                k = torch.cat([prev_k, k], dim=1)
                v = torch.cat([prev_v, v], dim=1)
            state['key'], state['value'] = k, v
            S = k.size(1)
            attn = AttentionFn.apply(q, k.view(B, S, D), v.view(B, S, D), attn_mask, H, self.causal and T > 1, scale)
        elif key is query and value is query:
            fold = rope_in_kernel()
            qkv = PackedLinearFn.apply(query, w, b, 3, link, q_w, k_w, v_w, q_b, k_b, v_b, *gp)
            if fold is None:
                qkv = rope(qkv, 0)
            attn = AttentionFn.apply(qkv, None, None, attn_mask, H, self.causal and T > 1, scale, drop, fold)
        elif key is value:
            q = LinearFn.apply(query, q_w, q_b, 'none', link, group)
            # (the decoder pass's encoder-gradient chain, if one is open: native_layer.open_chain / PackedLinearFn)
            from . import native_layer
            kv = PackedLinearFn.apply(key, w[D:], None if b is None else b[D:], 2, native_layer._chain, k_w, v_w, k_b, v_b, *gp)
            attn = AttentionFn.apply(q, kv, None, attn_mask, H, self.causal and T > 1, scale, drop)
        else:
            q = LinearFn.apply(query, q_w, q_b, 'none', link, group)
            k = LinearFn.apply(key, k_w, k_b, 'none', None, group)
            v = LinearFn.apply(value, v_w, v_b, 'none', None, group)
            attn = AttentionFn.apply(q, k, v, attn_mask, H, self.causal and T > 1, scale, drop)
        if (tail is not None and state is None and self.out_proj.lora is None
                and block_tail_eligible(B * T, self.out_proj.weight, tail.residual, tail.gamma)):
            attn = LinearResidualLnFn.apply(attn, self.out_proj.weight, self.out_proj.bias, tail.residual, tail.gamma,
                                            tail.beta, tail.eps, tail.p, link, group)
            tail.done = True
        else:
            attn = self.out_proj(attn, group=group) if group is not None else self.out_proj(attn)
        return attn, weights


# ------------------------------------------------------------------------------------------------------------
# speech frontend
# ------------------------------------------------------------------------------------------------------------
_NO_CONV_STACK = bool(os.environ.get('PASERO_NO_CONV_STACK'))  # diagnostic: every conv a node of its own


class ConvolutionSubsampler(nn.Module):
    """Conv1d (+GLU | GELU) stack (modules.py:774-834) computed channels-last as implicit GEMMs on the MFMA kernel:
    a (k*C_in)-wide window of the zero-padded (B, L, C_in) input IS a contiguous row of the im2col matrix, so the
    activations are never unfolded, and the two transposes of the reference disappear."""

    def __init__(self, in_channels: int, mid_channels: int, out_channels: int, kernel_sizes=(3, 3),
                 strides=None, activation: str = 'glu'):
        super().__init__()
        strides = strides or tuple(2 for _ in kernel_sizes)
        assert len(strides) == len(kernel_sizes)
        r = 2 if activation == 'glu' else 1
        self.conv_layers = nn.ModuleList()
        for i, (k, s) in enumerate(zip(kernel_sizes, strides)):
            first, last = i == 0, i == len(kernel_sizes) - 1
            self.conv_layers.append(nn.Conv1d(in_channels if first else mid_channels // r,
                                              out_channels * r if last else mid_channels, k, stride=s, padding=k // 2))
        self.activation_name = activation
        self.activation = nn.GLU(dim=1) if activation == 'glu' else nn.GELU()  # kept for introspection only

    def get_new_length(self, length: LongTensor) -> LongTensor:
        for conv in self.conv_layers:
            length = 1 + torch.div(length - conv.kernel_size[0] + 2 * conv.padding[0], conv.stride[0],
                                   rounding_mode='floor')
        return length

    def forward(self, x: Tensor, length: LongTensor):
        from .autograd import Conv1dChannelsLastFn, ConvStackFn
        if self.activation_name != 'glu' and len(self.conv_layers) > 1 and not _NO_CONV_STACK:
            # (one node for the whole stack: conv i writes into the padded input of conv i + 1 — autograd.ConvStackFn)
            wb = [t for conv in self.conv_layers for t in (conv.weight, conv.bias)]
            geoms = tuple((conv.stride[0], conv.padding[0]) for conv in self.conv_layers)
            return ConvStackFn.apply(x, 'gelu', geoms, *wb), self.get_new_length(length)
        for conv in self.conv_layers:
            act = 'gelu' if self.activation_name != 'glu' else 'none'
            x = Conv1dChannelsLastFn.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], act)
            if self.activation_name == 'glu':
                x = GLUFn.apply(x)
        return x, self.get_new_length(length)
