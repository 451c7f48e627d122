"""Host-side mirror of `pasero/models/transformer.py`: the encoder-decoder Transformer with the reference's class
names, constructor signature `(cfg, dist_cfg, task)`, parameter names (checkpoint keys), forward signatures / return
values, and subclass hooks (`build_encoder/build_decoder/build_layer`, `ffn`, `self_attention`, `*_prenorm`,
`*_postnorm`, `*_residual`), with all tensor work on the HIP kernels of libpasero_hip.so.

Differences from the reference are confined to HOW the same quantities are computed:
  * embedding * sqrt(d) + positions + dropout is one kernel;
  * q/k/v projections are one GEMM on a flat weight arena, attention reads them in place (no mask tensor);
  * `residual + dropout(x)` and the following LayerNorm are one kernel (when the hooks are not overridden);
  * the training forward never materialises (B, T, V) logits: tied projection + label-smoothed CE are chunked and
    fused, and the three `.item()` syncs of compute_loss become one 3-float device->host copy.
File:line citations are into /root/reference/pasero/models/transformer.py unless stated otherwise.
"""
import logging
import math
import os
from typing import Optional, Union

import torch
import torch.nn as nn
from torch import Tensor, LongTensor, BoolTensor

from . import modules, native_layer
from .profiling import block as _bench_block, region as _bench_region
from .modules import Embedding, Identity
from .autograd import (FFNFn, GatedFFNFn, ResidualLayerNormFn, ResidualDropoutFn, VocabCrossEntropyFn, CrossEntropyFn, tie_table,
                       AddPositionsFn, LinearFn, ResidualLink, WGradGroup, WGradSinkFn, BlockTail, FFNResidualLnFn,
                       block_tail_eligible, LayerNormForkFn, DropLink, ResidualDropoutLnFn)

from .config import register_model  # also enters the reference's registry when `pasero` is importable

logger = logging.getLogger('models')
# diagnostic: every weight-gradient GEMM launched on its own, as in round 1 (A/B of the grouped launch)
_NO_WGRAD_GROUP = bool(int(os.environ.get('PASERO_NO_WGRAD_GROUP', '0') or 0))
# diagnostic: the post-norm block ends as stand-alone LayerNorm launches (A/B of the fused GEMM + LayerNorm kernel)
_NO_FUSED_TAIL = bool(int(os.environ.get('PASERO_NO_FUSED_TAIL', '0') or 0))
# diagnostic: pre-norm sub-blocks call their LayerNorm hook and leave the sum of the two input gradients to autograd
_NO_LN_FORK = bool(int(os.environ.get('PASERO_NO_LN_FORK', '0') or 0))
_NO_DROP_LINK = bool(int(os.environ.get('PASERO_NO_DROP_LINK', '0') or 0))  # (A/B: stand-alone dropout masks in the pre-norm backward)
_NO_END_NORM = bool(int(os.environ.get('PASERO_NO_END_NORM', '0') or 0))  # (A/B: a pre-norm block end and the next block's LayerNorm as two launches)
# diagnostic: read the step's sums at the end of the forward pass, as the reference does (A/B of the deferred read)
_EAGER_LOGS = bool(int(os.environ.get('PASERO_EAGER_LOGS', '0') or 0))
LN2 = math.log(2)


def defined(*args):
    """first argument that is not None (pasero/utils.py `defined`)"""
    return next((x for x in args if x is not None), None)


def len_to_mask(lengths: LongTensor, size: Optional[int] = None) -> BoolTensor:
    """pasero/utils.py:258-268 — True at padding positions"""
    size = size or int(lengths.max())
    return torch.arange(size, device=lengths.device).unsqueeze(0) >= lengths.unsqueeze(1)


def _norm_cls(cfg, wrappable: bool = False):
    if cfg.rms_norm:
        return modules.WrappableRMSNorm if wrappable else modules.RMSNorm
    if cfg.norm_bias:
        return modules.WrappableLayerNorm if wrappable else modules.LayerNorm
    return modules.LayerNormWithoutBias


class StepLogs(dict):
    """`logs` of a training / validation step (pasero/models/transformer.py:375-380: 'loss', 'nll_loss' in bits,
    'num_tokens', 'num_lines') whose numbers are still on their way from the device.  The reference reads them with three
    `.item()` calls at the end of the forward pass, which makes the host wait for the whole forward before it can enqueue
    the backward (the GPU then idles until the first backward kernel arrives).  Here the three sums are copied to pinned
    memory asynchronously and the wait happens when a value is first READ — in `Trainer.train_step` that is after
    `loss.backward()` has been enqueued (training.py:402-446).  It is a dict with the reference's keys from the start
    (`in`, `len`, `keys()` need no wait); every way of reading a value goes through `_wait` first."""
    __slots__ = ('_pending',)

    def __init__(self, sums: Tensor, batch_size: int):
        super().__init__(loss=None, nll_loss=None, num_tokens=None, num_lines=batch_size)
        host = torch.empty(3, dtype=torch.float32, pin_memory=True)
        host.copy_(sums, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._pending = (host, event)

    def _wait(self) -> None:
        if self._pending is not None:
            (host, event), self._pending = self._pending, None
            event.synchronize()
            loss, nll, ntok = host.tolist()
            super().__setitem__('loss', loss / LN2)
            super().__setitem__('nll_loss', nll / LN2)
            super().__setitem__('num_tokens', int(ntok))

    def __getitem__(self, key):
        self._wait()
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        if key in ('loss', 'nll_loss', 'num_tokens'):  # a write before the first read must not be overwritten when the
            self._wait()                               # sums arrive; other keys (:300-321) need no wait
        super().__setitem__(key, value)

    def __delitem__(self, key):
        self._wait()
        super().__delitem__(key)

    def update(self, *args, **kwargs):
        self._wait()
        super().update(*args, **kwargs)

    def clear(self):
        self._pending = None
        super().clear()

    def __ior__(self, other):
        self._wait()
        super().update(other)
        return self

    def get(self, key, default=None):
        self._wait()
        return super().get(key, default)

    def __iter__(self):  # (overridden so that dict(logs) / {**logs} take the keys() + __getitem__ path, not a raw copy)
        return super().__iter__()

    def keys(self):
        return super().keys()

    def items(self):
        self._wait()
        return super().items()

    def values(self):
        self._wait()
        return super().values()

    def pop(self, *args):
        self._wait()
        return super().pop(*args)

    def popitem(self):
        self._wait()
        return super().popitem()

    def setdefault(self, key, default=None):
        self._wait()
        return super().setdefault(key, default)

    def copy(self):
        self._wait()
        return dict(super().items())

    def __eq__(self, other):
        self._wait()
        return super().__eq__(other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __or__(self, other):
        self._wait()
        return dict(super().items()) | other

    def __ror__(self, other):
        self._wait()
        return other | dict(super().items())

    def __repr__(self):
        self._wait()
        return super().__repr__()

    def __reduce__(self):
        self._wait()
        return (dict, (dict(super().items()),))


class BaseModel(nn.Module):
    """:24-42"""
    @property
    def padding_idx(self) -> int: return self.cfg.padding_idx
    @property
    def bos_idx(self) -> int: return self.cfg.bos_idx
    @property
    def eos_idx(self) -> int: return self.cfg.eos_idx
    @property
    def unk_idx(self) -> int: return self.cfg.unk_idx

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def parallelize(self, devices) -> None:
        if len(devices) > 1:
            raise NotImplementedError('pasero_amd: inference pipeline parallelism (--devices) is not implemented')


class Encoder(BaseModel):
    @property
    def max_len(self) -> int:
        return self.cfg.encoder_max_len


class Decoder(BaseModel):
    @property
    def max_len(self) -> int:
        return self.cfg.decoder_max_len

    @staticmethod
    def reorder_state(state: Optional[dict], indices: LongTensor) -> None:
        """:68-75 — beam search reorders the incremental state along the batch dimension"""
        if not state:
            return
        cache = state.get('_pk_decode')  # native decoding step (pasero_amd/decode.py): reorders its own buffers
        own = set()
        if cache is not None:
            cache.reorder(indices, state)
            own = {n for pair in cache.names for n in pair}
        for k, v in state.items():
            if k not in own and torch.is_tensor(v) and v.dim() > 0:
                state[k] = v.index_select(0, indices.to(v.device))


class DummyEncoder(Encoder):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.embed_tokens = None

    @property
    def max_len(self) -> int:
        return 0

    def forward(self, *args, **kwargs):
        return None, None, {}


class EncoderDecoder(BaseModel):
    @property
    def total_param_count(self) -> int:
        return sum(p.numel() for p in self.parameters())


@register_model('transformer')
class Transformer(EncoderDecoder):
    """:106-607"""

    def __init__(self, cfg, dist_cfg, task):
        super().__init__()
        self.cfg, self.dist_cfg, self.task = cfg, dist_cfg, task
        if (getattr(dist_cfg, 'tp_size', None) or 1) > 1:
            raise NotImplementedError('pasero_amd: tensor parallelism is not implemented; use data parallelism')
        if cfg.model_type == 'decoder':
            raise NotImplementedError('pasero_amd: decoder-only models are outside the hot-path scope')
        self.find_unused_parameters = False
        self.batch_by = None
        self.encoder = self.build_encoder()
        embed = self.encoder.embed_tokens if cfg.shared_embeddings else None
        self.decoder = self.build_decoder(embed=embed)

    @property
    def shard_count(self): return 1
    @property
    def shard_id(self): return 0
    @property
    def is_sharded(self): return False

    def build_encoder(self, embed: Optional[Embedding] = None):
        return TransformerEncoder(self.cfg, self.dist_cfg, task=self.task, embed=embed)

    def build_decoder(self, embed: Optional[Embedding] = None):
        return TransformerDecoder(self.cfg, self.dist_cfg, task=self.task, embed=embed)

    def disable_adapters(self) -> None: pass
    def enable_adapters(self) -> None: pass

    def forward(self, encoder_input: Optional[Tensor] = None, encoder_input_length: Optional[LongTensor] = None,
                decoder_input: Optional[LongTensor] = None, prompt_mask: Optional[Tensor] = None, **kwargs):
        """:227-321.  Returns (loss, logs) exactly like the reference: `loss` a 0-d fp32 tensor (SUM over target
        tokens, not normalised), `logs` = {'loss', 'nll_loss' (bits, floats), 'num_tokens', 'num_lines'}."""
        target = decoder_input[:, 1:]
        decoder_input = decoder_input[:, :-1]
        encoder_out, encoder_mask, enc_layer_outputs = self.encoder(encoder_input, encoder_input_length, **kwargs)
        pm = prompt_mask[:, :-1] if prompt_mask is not None else None
        fused = (type(self).compute_loss is Transformer.compute_loss and self.cfg.prompt_loss == 1.0)
        if fused:
            features, dec_layer_outputs = self.decoder(encoder_out, encoder_mask, decoder_input, prompt_mask=pm,
                                                       project=False, **kwargs)
            return self.compute_loss_fused(features, target)
        decoder_out, dec_layer_outputs = self.decoder(encoder_out, encoder_mask, decoder_input, prompt_mask=pm,
                                                      **kwargs)
        layer_outputs = {**enc_layer_outputs, **dec_layer_outputs}
        scale = self.cfg.prompt_loss
        if scale == 1.0:
            return self.compute_loss(decoder_out, target, layer_outputs)
        pmask = prompt_mask[:, 1:]  # :300-321
        loss, logs = self.compute_loss(decoder_out, target.masked_fill(pmask, self.padding_idx), layer_outputs)
        if scale > 0:
            p_loss, p_logs = self.compute_loss(decoder_out, target.masked_fill(~pmask, self.padding_idx),
                                               layer_outputs)
            logs['prompt_nll_loss'] = p_logs['nll_loss']
            logs['loss'] = logs['loss'] + scale * p_logs['loss']
            logs['num_tokens'] += p_logs['num_tokens']
            logs['num_prompt_tokens'] = p_logs['num_tokens']
            loss = loss + scale * p_loss
        return loss, logs

    @staticmethod
    def _logs(sums: Tensor, batch_size: int) -> dict:
        if sums.is_cuda and not _EAGER_LOGS:
            return StepLogs(sums, batch_size)  # no host sync here: the copy is waited for when a value is first read
        loss, nll, ntok = sums.tolist()  # the ONE host sync of the step (reference: 3x .item(), :375-377)
        return {'loss': loss / LN2, 'nll_loss': nll / LN2, 'num_tokens': int(ntok), 'num_lines': batch_size}

    @_bench_region('loss')
    def compute_loss_fused(self, features: Tensor, target: LongTensor):
        """tied projection + label-smoothed CE without materialising the logits (:324-380 + modules.py:935-947); in the
        `--benchmark` log the reference's 'output_projection' and 'loss' entries are both inside this one 'loss'"""
        dec = self.decoder
        weight = dec.embed_tokens.effective_weight() if dec.output_projection is None else dec.output_projection.weight
        if dec.output_projection is None:
            tie_table(weight)  # (the table is looked up AND projected onto: its gradient is assembled in one tensor)
        sums = VocabCrossEntropyFn.apply(features, weight, target, self.padding_idx, self.cfg.label_smoothing or 0.0)
        return sums[0], self._logs(sums.detach(), target.size(0))

    @_bench_region('loss')
    def compute_loss(self, logits: Tensor, target: LongTensor, layer_outputs: dict, *args, **kwargs):
        """:324-380 on materialised logits (API-compatible entry point; subclasses may override it)"""
        sums = CrossEntropyFn.apply(logits, target, self.padding_idx, self.cfg.label_smoothing or 0.0)
        return sums[0], self._logs(sums.detach(), target.size(0))

    # ---- checkpoint plumbing the Trainer touches (training.py:135-148,622-623,797-800,926) ----
    def remap_state_dict(self, state_dict: dict) -> None:
        """:382-417 — one-off conversions when fine-tuning another model: vocabulary re-mapping of the embedding-shaped
        tensors by the task, and `--shift-{encoder,decoder}-layers` (training only)"""
        for name, fn in (('encoder.embed_tokens.weight', 'remap_encoder_embed'),
                         ('decoder.embed_tokens.weight', 'remap_decoder_embed'),
                         ('decoder.output_projection.weight', 'remap_decoder_embed')):
            if name in state_dict:
                state_dict[name] = getattr(self.task, fn)(state_dict[name])
        import re
        for component in ('encoder', 'decoder'):
            shift = getattr(self.cfg, f'shift_{component}_layers', 0)
            if not (self.training and shift):
                continue
            pattern = re.compile(rf'{component}\.layers\.(\d+)\.')
            renamed = {}
            for key, value in state_dict.items():
                m = pattern.match(key)
                if m:
                    key = f'{component}.layers.{int(m.group(1)) + shift}.' + key[m.end():]
                renamed[key] = value
            state_dict.clear()
            state_dict.update(renamed)

    def update_state_dict(self, state_dict: dict) -> None:
        """:419-497: stale fairseq / HuggingFace keys, shared-embedding aliases, fairseq `in_proj` split, HF's name for
        the decoder's last norm, the frozen-embedding copy, LoRA (new branches when training, merged at inference)"""
        for k in list(state_dict):
            if k.endswith('.version'):
                state_dict.pop(k)
        state_dict.pop('lm_head.weight', None)
        enc, dec = 'encoder.embed_tokens.weight', 'decoder.embed_tokens.weight'
        if enc in state_dict and dec not in state_dict:
            state_dict[dec] = state_dict[enc]
        if dec in state_dict and enc not in state_dict:
            state_dict[enc] = state_dict[dec]
        if self.encoder.embed_tokens is None:
            state_dict.pop(enc, None)
        if self.cfg.tied_output_projection:
            for name in list(state_dict):
                if name.endswith('.output_projection.weight'):
                    state_dict.pop(name)
        for name in list(state_dict):
            if name.endswith('.in_proj_weight') or name.endswith('.in_proj_bias'):
                param = state_dict.pop(name)
                dim = param.size(0) // 3
                for i, s in enumerate(['.q_proj.', '.k_proj.', '.v_proj.']):
                    state_dict[name.replace('.in_proj_', s)] = param[dim * i:dim * (i + 1)]
            else:  # (HuggingFace checkpoints call the decoder's last norm `final_layer_norm`)
                state_dict[name.replace('decoder.final_layer_norm.', 'decoder.layer_norm.')] = state_dict.pop(name)
        frozen = 'encoder.embed_tokens.frozen_embedding.weight'
        if self.task.freeze_encoder_embed_mask is not None:
            state_dict[frozen] = state_dict[enc]
        else:
            state_dict.pop(frozen, None)
        if self.cfg.lora_rank and self.training:  # :479-482: new LoRA branches start from their random init
            modules.add_missing_parameters(self, state_dict, r'.*\.lora\..*')
        if not self.training:  # :484-497: at inference the low-rank updates are merged into the linear weights
            for name in list(state_dict):
                if name.endswith('.lora.down.weight'):
                    prefix = name[:-len('lora.down.weight')]
                    down = state_dict.pop(prefix + 'lora.down.weight')
                    up = state_dict.pop(prefix + 'lora.up.weight')
                    patch = torch.matmul(up.float(), down.float() * self.cfg.lora_alpha / down.size(0)).to(down.dtype)
                    state_dict[prefix + 'weight'] = state_dict[prefix + 'weight'] + patch.to(state_dict[prefix + 'weight'].device)

    @classmethod
    def shard_state_dict(cls, state_dict, shard_id=0, shard_count=1, **kwargs):
        if shard_count != 1:
            raise NotImplementedError('pasero_amd: tensor-parallel checkpoints are not supported')
        return state_dict

    @classmethod
    def unshard_state_dict(cls, *state_dicts, **kwargs):
        if len(state_dicts) != 1:
            raise NotImplementedError('pasero_amd: tensor-parallel checkpoints are not supported')
        return state_dicts[0]

    def set_ddp_params_and_buffers_to_ignore(self):
        self._ddp_params_and_buffers_to_ignore = getattr(self, '_ddp_params_and_buffers_to_ignore', [])

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        status = super().load_state_dict(state_dict, strict)
        if not strict:
            if status.missing_keys:
                logger.warning('missing keys: ' + ' '.join(status.missing_keys))
            if status.unexpected_keys:
                logger.warning('unexpected keys: ' + ' '.join(status.unexpected_keys))
        return status

    def clean_state_dict(self, state_dict: dict) -> None:
        key = 'decoder.embed_tokens.weight'
        if key in state_dict and state_dict[key].numel() == 0:
            state_dict.pop(key)
        state_dict.pop('encoder.embed_tokens.frozen_embedding.weight', None)
        state_dict.pop('decoder.embed_tokens.frozen_embedding.weight', None)

    def parallelize(self, devices) -> None:
        assert not self.training
        self.encoder.parallelize(devices)
        self.decoder.parallelize(devices)


class TransformerEncoder(Encoder):
    """:610-764"""

    def __init__(self, cfg, dist_cfg, task, embed: Optional[Embedding] = None):
        super().__init__()
        self.cfg, self.dist_cfg, self.task = cfg, dist_cfg, task
        if self.task.encoder_num_embeddings == 0:
            self.embed_tokens = None
        elif embed is None:
            self.embed_tokens = Embedding(self.task.encoder_num_embeddings, cfg.embed_dim, self.padding_idx,
                                          task.freeze_encoder_embed_mask)
        else:
            self.embed_tokens = embed
        self.subsample = self.in_linear = None
        input_dim = cfg.input_dim or cfg.embed_dim
        encoder_max_len = cfg.encoder_max_len
        if cfg.conv_kernel_sizes:
            conv_input_dim = cfg.conv_input_dim or input_dim
            conv_channels = cfg.conv_channels or conv_input_dim
            if conv_input_dim != input_dim:
                self.in_linear = nn.Sequential(modules.WrappableLinear(input_dim, cfg.conv_input_dim), nn.ReLU())
            self.subsample = modules.ConvolutionSubsampler(conv_input_dim, conv_channels, cfg.embed_dim,
                                                           cfg.conv_kernel_sizes, cfg.conv_strides,
                                                           cfg.conv_activation)
            encoder_max_len = int(self.subsample.get_new_length(torch.tensor(encoder_max_len)))
        elif input_dim != cfg.embed_dim:
            self.in_linear = modules.WrappableLinear(input_dim, cfg.embed_dim)
        self.embed_positions = modules.PositionalEmbedding(cfg.encoder_positional_encoding, encoder_max_len,
                                                           cfg.embed_dim, shift=cfg.positional_encoding_shift)
        self.embed_scale = math.sqrt(cfg.embed_dim) if cfg.scale_embed else 1
        Norm = _norm_cls(cfg, wrappable=True)
        self.layernorm_embedding = Norm(cfg.embed_dim, eps=cfg.norm_eps) if cfg.encoder_embed_norm else Identity()
        self.dropout = modules.Dropout(defined(cfg.embed_dropout, cfg.dropout))
        self.layers = nn.ModuleList([self.build_layer(i) for i in range(cfg.encoder_layers)])
        self.layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps) if cfg.encoder_prenorm else Identity()
        self.device = None

    def build_layer(self, layer_id: int):
        layer = TransformerEncoderLayer(self.cfg, self.dist_cfg, layer_id)
        return modules.checkpoint_wrapper(layer, activate=self.cfg.checkpoint_activations)

    @_bench_region('encoder')
    def forward(self, encoder_input: Tensor, encoder_input_length: LongTensor, return_layers=[], meta: dict = {},
                **kwargs):
        """:698-752 -> (encoder_out (B,S,D), padding_mask (B,S) bool, layer_outputs)"""
        return_layers = return_layers or ()
        length = encoder_input_length
        pos = self.embed_positions
        embed_norm = not isinstance(self.layernorm_embedding, Identity)
        p = self.dropout.p if (self.training and not embed_norm) else 0.0
        if encoder_input.ndim == 2:  # token ids: gather * scale + positions (+ dropout) in one kernel
            S = encoder_input.size(1)
            pos.check_length(S)
            x = self.embed_tokens.embed(encoder_input, pos.table(), self.embed_scale, pos.shift, p)
        else:  # speech features (:731-744)
            x = encoder_input
            if self.in_linear is not None:
                if isinstance(self.in_linear, nn.Sequential):
                    lin = self.in_linear[0]
                    x = LinearFn.apply(x, lin.weight, lin.bias, 'relu')
                else:
                    x = self.in_linear(x)
            if self.subsample is not None:
                x, length = self.subsample(x, length)
            S = x.size(1)
            pos.check_length(S)
            x = AddPositionsFn.apply(x, pos.table(), float(self.embed_scale), pos.shift, p)
        padding_mask = len_to_mask(length, size=S)
        if embed_norm:
            x = self.dropout(self.layernorm_embedding(x))
        layer_outputs = {}
        for layer in self.layers:
            x, layer_output = layer(x, padding_mask, return_layers)
            layer_outputs.update(layer_output)
        x = self.layer_norm(x)
        return x, padding_mask, layer_outputs


class TransformerDecoder(Decoder):
    """:767-910"""

    def __init__(self, cfg, dist_cfg, task, embed: Optional[Embedding] = None):
        super().__init__()
        self.cfg, self.dist_cfg, self.task = cfg, dist_cfg, task
        self.embed_tokens = (Embedding(task.decoder_num_embeddings, cfg.embed_dim, self.padding_idx)
                             if embed is None else embed)
        self.embed_positions = modules.PositionalEmbedding(cfg.decoder_positional_encoding, cfg.decoder_max_len,
                                                           cfg.embed_dim, shift=cfg.positional_encoding_shift)
        self.embed_scale = math.sqrt(cfg.embed_dim) if cfg.scale_embed else 1
        self.dropout = modules.Dropout(defined(cfg.embed_dropout, cfg.decoder_dropout, cfg.dropout))
        self.layers = nn.ModuleList([self.build_layer(i) for i in range(cfg.decoder_layers)])
        Norm = _norm_cls(cfg, wrappable=True)
        self.layernorm_embedding = Norm(cfg.embed_dim, eps=cfg.norm_eps) if cfg.decoder_embed_norm else Identity()
        self.layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps) if cfg.decoder_prenorm else Identity()
        if cfg.tied_output_projection:
            self.output_projection = None
        else:
            self.output_projection = modules.WrappableLinear(cfg.embed_dim, self.task.decoder_num_embeddings,
                                                             bias=False)
            nn.init.xavier_uniform_(self.output_projection.weight)
        self.device = None

    def build_layer(self, layer_id: int):
        layer = TransformerDecoderLayer(self.cfg, self.dist_cfg, layer_id)
        return modules.checkpoint_wrapper(layer, activate=self.cfg.checkpoint_activations)

    @_bench_region('decoder')
    def forward(self, encoder_out: Tensor, encoder_mask: BoolTensor, decoder_input: LongTensor,
                prompt_mask: Optional[Tensor] = None, state: Optional[dict] = None, return_layers=[],
                meta: dict = {}, project: bool = True, **kwargs):
        """:831-898 -> (logits (B,T,V), layer_outputs).  `project=False` (used by the fused training loss) returns the
        features before the output projection instead of the logits."""
        return_layers = return_layers or ()
        if state:  # one new token per sentence on top of a non-empty state: the native decoding step
            from .decode import try_step
            logits = try_step(self, encoder_out, encoder_mask, decoder_input, state, return_layers, project)
            if logits is not None:
                return logits, {}
        padding_mask = decoder_input.eq(self.padding_idx)
        T = decoder_input.size(1)
        pos_offset = state.get('offset', 0) if state else 0
        pos = self.embed_positions
        pos.check_length(T, pos_offset)
        if state is not None:
            state['offset'] = pos_offset + T
        embed_norm = not isinstance(self.layernorm_embedding, Identity)
        p = self.dropout.p if (self.training and not embed_norm) else 0.0
        x = self.embed_tokens.embed(decoder_input, pos.table(), self.embed_scale, pos.shift + pos_offset, p)
        if embed_norm:
            x = self.dropout(self.layernorm_embedding(x))
        layer_outputs = {}
        chain = native_layer.open_chain(encoder_out if state is None else None)  # (this pass's encoder-gradient tally)
        try:
            for layer in self.layers:
                x, layer_output = layer(x, encoder_out, encoder_mask, padding_mask, prompt_mask=prompt_mask, state=state,
                                        return_layers=return_layers)
                layer_outputs.update(layer_output)
        finally:
            native_layer.close_chain(chain)
        x = self.layer_norm(x)
        if not project:
            return x, layer_outputs
        with _bench_block('output_projection'):
            if self.output_projection is None:
                x = self.embed_tokens.projection(x)
            else:
                x = self.output_projection(x)
        return x, layer_outputs


class _LayerBase(modules._PerCallAttrs, nn.Module):
    """pieces shared by the encoder and decoder layers"""

    def _build_ffn(self, cfg, ffn_dim: int):
        lora = dict(lora_rank=cfg.lora_rank, lora_alpha=cfg.lora_alpha)
        self.fc1 = modules.Linear(cfg.embed_dim, ffn_dim, bias=cfg.has_bias, **lora)
        self.fc2 = modules.Linear(ffn_dim, cfg.embed_dim, bias=cfg.has_bias, **lora)
        self.fc3 = (modules.Linear(cfg.embed_dim, ffn_dim, bias=cfg.has_bias, **lora)
                    if cfg.activation_fn in ('swiglu', 'geglu') else None)  # Llama / T5 gate (:966-972)
        self.activation_fn = modules.get_activation_fn(cfg.activation_fn)
        self.activation_dropout = modules.Dropout(cfg.activation_dropout)

    def _ffn(self, x: Tensor) -> Tensor:
        link, self._ffn_link = getattr(self, '_ffn_link', None), None
        if self.training and self.activation_dropout.p > 0 and self.fc3 is not None:
            raise NotImplementedError('pasero_amd: activation dropout inside a gated FFN is not implemented')
        if self.fc1.lora is not None:  # LoRA branches on fc1 / fc2 / fc3: the module-by-module formulation (:999-1019)
            if self.fc3 is not None:
                raise NotImplementedError('pasero_amd: LoRA on a gated feed-forward is not implemented')
            h = self.fc1(x, link=link)
            h = self.activation_fn(h)
            return self.fc2(self.activation_dropout(h))
        if self.fc3 is not None:
            return GatedFFNFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc3.weight, self.fc3.bias,
                                    self.fc2.weight, self.fc2.bias, self.activation_fn.name, link)
        group, self._ffn_group = getattr(self, '_ffn_group', None), None
        if not (self.training and self.activation_dropout.p > 0):
            return FFNFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                               self.activation_fn.name, link, group)
        y = self.activation_dropout(self.activation_fn(self.fc1(x, link=link, group=group)))
        return self.fc2(y, group=group)

    def _wgrad_open(self, x: Tensor, attns) -> Tensor:
        """Entry of a layer: collect this layer's weight-gradient GEMMs into one grouped launch (autograd.WGradGroup).
        The sink node goes on `x`; the attention modules and the feed-forward pick the group up for this one call."""
        self._wgrad_close(attns)
        if (_NO_WGRAD_GROUP or not torch.is_grad_enabled() or not x.is_cuda or not x.requires_grad
                or x.dtype not in (torch.bfloat16, torch.float16) or torch.is_autocast_enabled('cuda')
                or self.cfg.checkpoint_activations or self.fc3 is not None):
            return x
        linears = [self.fc1, self.fc2]
        for a in attns:
            linears += [a.q_proj, a.k_proj, a.v_proj, a.out_proj]
        if any(m.lora is not None for m in linears):
            return x
        params = [p for m in linears for p in (m.weight, m.bias) if p is not None and p.requires_grad
                  and p.dtype == x.dtype]
        if not params:
            return x
        group = WGradGroup()
        for a in attns:
            a._wgroup = group
        self._ffn_group = group
        y = WGradSinkFn.apply(x, group, *params)
        # (the sink is an identity whose backward passes the gradient through untouched: the note of the dropout that produced x
        # — autograd.DropLink, left by the layer below — stays valid on its output)
        link = getattr(x, '_pk_drop_link', None)
        if link is not None:
            y._pk_drop_link = link
        return y

    def _wgrad_close(self, attns) -> None:
        """(a hook that skipped its sub-block must not leave the group behind for a later call)"""
        for a in attns:
            a._wgroup = None
        self._ffn_group = None

    def _residual(self, x: Tensor, residual: Tensor) -> Tensor:
        p = self.dropout.p if self.training else 0.0
        if p > 0 and self.prenorm and not _NO_LN_FORK and not _NO_DROP_LINK and torch.is_grad_enabled() and x.dtype != torch.float32:
            # (the fork that reads z next writes the masked gradient for this dropout in its own backward pass: autograd.DropLink)
            link = DropLink()
            z = ResidualDropoutFn.apply(x, residual, p, link)
            z._pk_drop_link = link
            return z
        return ResidualDropoutFn.apply(x, residual, p)

    def _norm_module(self, norm):
        """`final_layer_norm` is a lambda when --shared-norm (:977-980)"""
        return norm if isinstance(norm, nn.Module) else self.self_attn_layer_norm

    def _prenorm(self, x: Tensor, hook: str, norm):
        """`residual = x; x = *_prenorm(x)` of a sub-block -> (x, residual).  Pre-norm layers with the stock hook and a
        LayerNorm module: one autograd node for both uses of the input (autograd.LayerNormForkFn — the two gradients meet
        inside the LayerNorm backward kernel instead of in an elementwise addition); the reference's hook call otherwise.
        The node reads the module's parameters and does not go through its `__call__`: a norm module that carries forward /
        pre-forward hooks (`register_forward_hook`) keeps the hook call, so those hooks fire as in the reference."""
        m = self._norm_module(norm)
        pre = x.__dict__.pop('_pk_prenormed', None) if isinstance(x, Tensor) else None
        if pre is not None and pre[0] is m and self._hooks_are_base(hook):
            return pre[1], x  # (LayerNorm(x) came with x: the block end in front of this block wrote both — `_block_end`)
        if (self.prenorm and not _NO_LN_FORK and torch.is_grad_enabled() and x.requires_grad and x.is_cuda
                and isinstance(m, modules.LayerNorm) and getattr(m, 'weight', None) is not None
                and not m._forward_hooks and not m._forward_pre_hooks
                and self._hooks_are_base(hook) and not torch.is_autocast_enabled('cuda')):
            y, residual = LayerNormForkFn.apply(x, m.weight, m.bias, m.eps, getattr(x, '_pk_drop_link', None))
            return y, residual
        return getattr(self, hook)(x), x

    def _hooks_are_base(self, *names) -> bool:
        cls = TransformerDecoderLayer if isinstance(self, TransformerDecoderLayer) else TransformerEncoderLayer
        return all(getattr(type(self), n) is getattr(cls, n) for n in names)

    def _stock_postnorm(self, branch_hooks, residual_hook: str, postnorm_hook: str, norm) -> bool:
        """a post-norm sub-block whose hooks are all the stock ones and whose norm is a LayerNorm module"""
        return (not self.prenorm and isinstance(self._norm_module(norm), modules.LayerNorm)
                and self._hooks_are_base(residual_hook, postnorm_hook, *branch_hooks))

    def _linked(self, branch_hooks, residual_hook: str, postnorm_hook: str, norm, target, attr: str):
        """For a post-norm sub-block whose hooks are all the stock ones: hand a ResidualLink to the module that runs the
        sub-block's first GEMM (`target.attr`), to be passed on to the fused block end.  None otherwise."""
        if not torch.is_grad_enabled() or not self._stock_postnorm(branch_hooks, residual_hook, postnorm_hook, norm):
            return None
        link = ResidualLink()
        setattr(target, attr, link)
        return link

    def _tail_for(self, attn, branch_hooks, residual_hook: str, postnorm_hook: str, norm, residual: Tensor):
        """For a stock post-norm attention block: let `attn` (the MultiheadAttention that runs the block's last GEMM)
        finish the block inside its out_proj kernel — residual + dropout + LayerNorm in the GEMM epilogue
        (autograd.LinearResidualLnFn).  The module may decline (shape, dtype, LoRA, incremental state): `tail.done`."""
        attn._tail = None
        if _NO_FUSED_TAIL or not self._stock_postnorm(branch_hooks, residual_hook, postnorm_hook, norm):
            return None
        norm = self._norm_module(norm)
        if getattr(norm, 'weight', None) is None:  # LayerNorm(elementwise_affine=False): the stand-alone kernel
            return None
        tail = BlockTail(residual, norm.weight, norm.bias, norm.eps, self.dropout.p if self.training else 0.0)
        attn._tail = tail
        return tail

    def _ffn_block(self, x: Tensor, residual: Tensor, padding_mask, link) -> Tensor:
        """`x = ffn_prenorm(x); x = ffn(x, ...); x = ffn_residual(x, residual); x = ffn_postnorm(x)` — for a stock
        post-norm block of base width: fc1 (+ activation) and fc2 + bias + dropout + residual + LayerNorm as two launches
        (autograd.FFNResidualLnFn); the reference's hook sequence otherwise"""
        norm = self._norm_module(self.final_layer_norm)
        if (not _NO_FUSED_TAIL and x is residual and self.fc3 is None and self.fc1.lora is None
                and not (self.training and self.activation_dropout.p > 0)
                and self._stock_postnorm(('ffn', 'ffn_prenorm'), 'ffn_residual', 'ffn_postnorm', self.final_layer_norm)
                and getattr(norm, 'weight', None) is not None
                and block_tail_eligible(x.numel() // x.size(-1), self.fc2.weight, x, norm.weight)):
            self._ffn_link = None
            group, self._ffn_group = getattr(self, '_ffn_group', None), None
            return FFNResidualLnFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                                         self.activation_fn.name, norm.weight, norm.bias, norm.eps,
                                         self.dropout.p if self.training else 0.0, group)
        if x is residual:
            x, residual = self._prenorm(x, 'ffn_prenorm', self.final_layer_norm)
        else:
            x = self.ffn_prenorm(x)
        x = self.ffn(x, residual, padding_mask)
        return self._block_end(x, residual, self.final_layer_norm, 'ffn_residual', 'ffn_postnorm', link)

    def _block_end(self, x: Tensor, residual: Tensor, norm, residual_hook: str, postnorm_hook: str,
                   link=None, tail=None, next_norm=None, next_hook: str = None) -> Tensor:
        """`x = *_residual(x, residual); x = *_postnorm(x)` — one fused kernel for post-norm layers whose hooks are not
        overridden by a subclass, the reference's two hook calls otherwise; nothing at all if the sub-block's last GEMM
        has already done it (`tail.done`)"""
        if tail is not None and tail.done:
            return x
        norm = self._norm_module(norm)
        if self._hooks_are_base(residual_hook, postnorm_hook) and not self.prenorm and isinstance(norm, modules.LayerNorm):
            return ResidualLayerNormFn.apply(x, residual, norm.weight, norm.bias, norm.eps,
                                             self.dropout.p if self.training else 0.0, link)
        if next_norm is not None and self._end_and_next_norm_ok(x, residual_hook, postnorm_hook, next_hook, next_norm):
            # pre-norm: z = residual + dropout(x) and the LayerNorm that opens the NEXT block of this layer in one launch (and one
            # autograd node: autograd.ResidualDropoutLnFn); `_prenorm` of that block finds LayerNorm(z) on z
            m = self._norm_module(next_norm)
            y, z = ResidualDropoutLnFn.apply(x, residual, self.dropout.p if self.training else 0.0, m.weight, m.bias, m.eps)
            z._pk_prenormed = (m, y)
            return z
        x = getattr(self, residual_hook)(x, residual)
        return getattr(self, postnorm_hook)(x)

    def _end_and_next_norm_ok(self, x: Tensor, residual_hook: str, postnorm_hook: str, next_hook: str, next_norm) -> bool:
        m = self._norm_module(next_norm)
        return (self.prenorm and not _NO_END_NORM and not _NO_LN_FORK and torch.is_grad_enabled() and x.requires_grad and x.is_cuda
                and x.dtype in (torch.bfloat16, torch.float16) and isinstance(m, modules.LayerNorm)
                and getattr(m, 'weight', None) is not None and m.weight.dtype == x.dtype
                and not m._forward_hooks and not m._forward_pre_hooks
                and self._hooks_are_base(residual_hook, postnorm_hook, next_hook) and not torch.is_autocast_enabled('cuda'))


class TransformerEncoderLayer(_LayerBase):
    """:913-1099"""

    def __init__(self, cfg, dist_cfg, layer_id: int):
        super().__init__()
        self.cfg, self.dist_cfg, self.layer_id = cfg, dist_cfg, layer_id
        self.self_attn = modules.MultiheadAttention(
            cfg.embed_dim, cfg.encoder_attention_heads, dropout=cfg.attention_dropout,
            positional_encoding=cfg.encoder_positional_encoding, lora_rank=cfg.lora_rank, lora_alpha=cfg.lora_alpha,
            has_bias=cfg.has_bias, key_bias=cfg.attention_key_bias, layer_id=layer_id, scaled=cfg.scale_attn,
            rope_base=cfg.rope_base, alibi_max_bias=cfg.alibi_max_bias, max_qkv=cfg.max_qkv)
        Norm = _norm_cls(cfg)
        self.dropout = modules.Dropout(cfg.dropout)
        self.self_attn_layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps)
        self.prenorm = cfg.encoder_prenorm
        self._build_ffn(cfg, cfg.encoder_ffn_dim)
        if cfg.shared_norm:
            self.final_layer_norm = lambda x: self.self_attn_layer_norm(x)
        else:
            self.final_layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps)
        self.set_name(f'enc_{layer_id}')
        self.return_layers = []
        self.layer_outputs = {}
        self.enter = self.exit = nn.Identity()
        self.device = None

    def set_name(self, name: str):
        self.name = name
        self.self_attn_key = f'{name}_self_attn'

    def ffn(self, x: Tensor, residual: Tensor, padding_mask: BoolTensor) -> Tensor:
        return self._ffn(x)

    def self_attention(self, x: Tensor, residual: Tensor, padding_mask: BoolTensor) -> Tensor:
        return_attn = self.self_attn_key in self.return_layers
        x, self_attn = self.self_attn(query=x, key=x, value=x, attn_mask=padding_mask, return_attn=return_attn)
        if return_attn:
            self.layer_outputs[self.self_attn_key] = self_attn
        return x

    def self_attn_residual(self, x, residual): return self._residual(x, residual)
    def self_attn_prenorm(self, x): return self.self_attn_layer_norm(x) if self.prenorm else x
    def self_attn_postnorm(self, x): return x if self.prenorm else self.self_attn_layer_norm(x)
    def ffn_residual(self, x, residual): return self._residual(x, residual)
    def ffn_prenorm(self, x): return self.final_layer_norm(x) if self.prenorm else x
    def ffn_postnorm(self, x): return x if self.prenorm else self.final_layer_norm(x)

    def forward(self, x: Tensor, padding_mask: BoolTensor, /, return_layers=[]):
        """:1056-1099"""
        if self.cfg.check_inf:
            raise NotImplementedError('pasero_amd: --check-inf clamping (fp16 T5) is not implemented')
        if native_layer.takes(self, x, None, None, return_layers, False, padding_mask):  # the stock layer: one C call per direction
            return native_layer.run(self, x, None, padding_mask, None, False), {}
        self.return_layers = return_layers
        x = self._wgrad_open(x, (self.self_attn,))
        residual = x
        link = self._linked(('self_attention', 'self_attn_prenorm'), 'self_attn_residual', 'self_attn_postnorm',
                            self.self_attn_layer_norm, self.self_attn, '_residual_link')
        tail = self._tail_for(self.self_attn, ('self_attention', 'self_attn_prenorm'), 'self_attn_residual',
                              'self_attn_postnorm', self.self_attn_layer_norm, residual)
        x, residual = self._prenorm(x, 'self_attn_prenorm', self.self_attn_layer_norm)
        x = self.self_attention(x, residual, padding_mask)
        self.self_attn._tail = None
        x = self._block_end(x, residual, self.self_attn_layer_norm, 'self_attn_residual', 'self_attn_postnorm', link,
                            tail, self.final_layer_norm, 'ffn_prenorm')
        residual = x
        link = self._linked(('ffn', 'ffn_prenorm'), 'ffn_residual', 'ffn_postnorm', self.final_layer_norm, self,
                            '_ffn_link')
        x = self._ffn_block(x, residual, padding_mask, link)
        self._wgrad_close((self.self_attn,))
        layer_outputs = self.layer_outputs
        self.layer_outputs = {}
        self.return_layers = []
        if self.name in return_layers:
            layer_outputs[self.name] = x
        return x, layer_outputs


class TransformerDecoderLayer(_LayerBase):
    """:1102-1417"""

    def __init__(self, cfg, dist_cfg, layer_id: int):
        super().__init__()
        self.cfg, self.dist_cfg, self.layer_id = cfg, dist_cfg, layer_id
        if cfg.parallel_attention:
            raise NotImplementedError('pasero_amd: parallel attention/FFN blocks are outside the hot-path scope')
        common = dict(dropout=cfg.attention_dropout, lora_rank=cfg.lora_rank, lora_alpha=cfg.lora_alpha,
                      has_bias=cfg.has_bias, key_bias=cfg.attention_key_bias, layer_id=layer_id,
                      scaled=cfg.scale_attn, max_qkv=cfg.max_qkv)
        self.self_attn = modules.MultiheadAttention(
            cfg.embed_dim, cfg.decoder_attention_heads, kv_heads=cfg.attention_heads_kv,
            sliding_window=cfg.sliding_window, positional_encoding=cfg.decoder_positional_encoding,
            max_len=cfg.decoder_max_len, causal=True, rope_base=cfg.rope_base, alibi_max_bias=cfg.alibi_max_bias,
            **common)
        Norm = _norm_cls(cfg)
        self.dropout = modules.Dropout(defined(cfg.decoder_dropout, cfg.dropout))
        self.self_attn_layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps)
        self.prenorm = cfg.decoder_prenorm
        self.encoder_attn = modules.MultiheadAttention(cfg.embed_dim, cfg.decoder_attention_heads, **common)
        self.encoder_attn_layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps)
        self._build_ffn(cfg, cfg.decoder_ffn_dim)
        if cfg.shared_norm:
            self.final_layer_norm = lambda x: self.self_attn_layer_norm(x)
        else:
            self.final_layer_norm = Norm(cfg.embed_dim, eps=cfg.norm_eps)
        self.set_name(f'dec_{layer_id}')
        self.return_layers = []
        self.layer_outputs = {}
        self.enter = self.exit = nn.Identity()
        self.device = None

    def set_name(self, name: str):
        self.name = name
        self.self_attn_key = f'{name}_self_attn'
        self.cross_attn_key = f'{name}_cross_attn'

    def ffn(self, x: Tensor, residual: Tensor, padding_mask: BoolTensor) -> Tensor:
        return self._ffn(x)

    def self_attention(self, x: Tensor, residual: Tensor, padding_mask: BoolTensor,
                       self_attn_mask: Optional[BoolTensor] = None, state: Optional[dict] = None) -> Tensor:
        """:1246-1291 — the flat `state` dict holds '{dec_i}_self_attn_key' / '_value' tensors (B,S,H,hd)"""
        if state is not None:
            prefix = f'{self.self_attn_key}_'
            self_attn_state = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
        else:
            self_attn_state = None
        return_attn = self.self_attn_key in self.return_layers
        x, self_attn = self.self_attn(query=x, key=x, value=x, state=self_attn_state, return_attn=return_attn,
                                      attn_mask=self_attn_mask)
        if return_attn:
            self.layer_outputs[self.self_attn_key] = self_attn
        if self_attn_state:
            state.update({f'{prefix}{k}': v for k, v in self_attn_state.items()})
        return x

    def cross_attention(self, x: Tensor, residual: Tensor, encoder_out: Tensor, encoder_mask: BoolTensor) -> Tensor:
        return_attn = self.cross_attn_key in self.return_layers
        x, cross_attn = self.encoder_attn(query=x, key=encoder_out, value=encoder_out, attn_mask=encoder_mask,
                                          return_attn=return_attn)
        if return_attn:
            self.layer_outputs[self.cross_attn_key] = cross_attn
        return x

    def self_attn_residual(self, x, residual): return self._residual(x, residual)
    def self_attn_prenorm(self, x): return self.self_attn_layer_norm(x) if self.prenorm else x
    def self_attn_postnorm(self, x): return x if self.prenorm else self.self_attn_layer_norm(x)
    def cross_attn_residual(self, x, residual): return self._residual(x, residual)
    def cross_attn_prenorm(self, x): return self.encoder_attn_layer_norm(x) if self.prenorm else x
    def cross_attn_postnorm(self, x): return x if self.prenorm else self.encoder_attn_layer_norm(x)
    def ffn_residual(self, x, residual): return self._residual(x, residual)
    def ffn_prenorm(self, x): return self.final_layer_norm(x) if self.prenorm else x
    def ffn_postnorm(self, x): return x if self.prenorm else self.final_layer_norm(x)

    def forward(self, x: Tensor, encoder_out: Optional[Tensor], encoder_mask: Optional[BoolTensor],
                padding_mask: BoolTensor, /, self_attn_mask: Optional[BoolTensor] = None,
                prompt_mask: Optional[BoolTensor] = None, state: Optional[dict] = None, return_layers=[]):
        """:1341-1417"""
        if self.cfg.check_inf:
            raise NotImplementedError('pasero_amd: --check-inf clamping (fp16 T5) is not implemented')
        if self_attn_mask is None and native_layer.takes(self, x, encoder_out, state, return_layers, True, encoder_mask):
            return native_layer.run(self, x, encoder_out, None, encoder_mask, True), {}
        self.return_layers = return_layers
        if state is None:
            x = self._wgrad_open(x, (self.self_attn, self.encoder_attn))
        residual = x
        link = None
        if state is None:
            link = self._linked(('self_attention', 'self_attn_prenorm'), 'self_attn_residual', 'self_attn_postnorm',
                                self.self_attn_layer_norm, self.self_attn, '_residual_link')
        tail = None
        if state is None:
            tail = self._tail_for(self.self_attn, ('self_attention', 'self_attn_prenorm'), 'self_attn_residual',
                                  'self_attn_postnorm', self.self_attn_layer_norm, residual)
        x, residual = self._prenorm(x, 'self_attn_prenorm', self.self_attn_layer_norm)
        x = self.self_attention(x, residual, padding_mask, self_attn_mask=self_attn_mask, state=state)
        self.self_attn._tail = None
        x = self._block_end(x, residual, self.self_attn_layer_norm, 'self_attn_residual', 'self_attn_postnorm', link,
                            tail, self.encoder_attn_layer_norm if state is None else None, 'cross_attn_prenorm')
        residual = x
        link = self._linked(('cross_attention', 'cross_attn_prenorm'), 'cross_attn_residual', 'cross_attn_postnorm',
                            self.encoder_attn_layer_norm, self.encoder_attn, '_residual_link')
        tail = self._tail_for(self.encoder_attn, ('cross_attention', 'cross_attn_prenorm'), 'cross_attn_residual',
                              'cross_attn_postnorm', self.encoder_attn_layer_norm, residual)
        x, residual = self._prenorm(x, 'cross_attn_prenorm', self.encoder_attn_layer_norm)
        x = self.cross_attention(x, residual, encoder_out, encoder_mask)
        self.encoder_attn._tail = None
        x = self._block_end(x, residual, self.encoder_attn_layer_norm, 'cross_attn_residual', 'cross_attn_postnorm',
                            link, tail, self.final_layer_norm if state is None else None, 'ffn_prenorm')
        residual = x
        link = self._linked(('ffn', 'ffn_prenorm'), 'ffn_residual', 'ffn_postnorm', self.final_layer_norm, self,
                            '_ffn_link')
        x = self._ffn_block(x, residual, padding_mask, link)
        self._wgrad_close((self.self_attn, self.encoder_attn))
        layer_outputs = self.layer_outputs
        self.layer_outputs = {}
        if self.name in return_layers:
            layer_outputs[self.name] = x
        return x, layer_outputs
