"""`adapter_transformer`: the Transformer with a bottleneck adapter after every layer (Bapna et al., 2019), the model the
IWSLT2023 recipes fine-tune (`adapter_nllb_1b3`, frozen backbone).  Mirror of pasero/models/adapters.py on top of the
pasero_amd layers: the same class names, constructor arguments, `adapters` ModuleDict / parameter names
(`...layers.N.adapters.<name>.{layer_norm,down,up}.*`), freezing rule and checkpoint handling, so adapter checkpoints are
interchangeable.  The adapter itself is one fused autograd function (`modules.AdapterLayer` -> `autograd.AdapterFn`);
frozen parameters cost no weight-gradient GEMMs (every backward here computes only the gradients autograd asks for).
"""
from typing import Optional

import torch.nn as nn
from torch import Tensor

from . import modules
from .config import register_model
from .modules import AdapterLayer
from .transformer import (DummyEncoder, Transformer, TransformerDecoder, TransformerDecoderLayer, TransformerEncoder,
                          TransformerEncoderLayer)

_LANG_PREFIX = 'lang:'      # adapters.py:33-34
_DOMAIN_PREFIX = 'domain:'
_ADAPTER_REGEX = r'(?P<module>encoder|decoder)\..*\.adapters\.(?P<uid>.*?)\.'


def _adapter_names(explicit, by_keys, task, batch_by: set):
    """adapters.py:71-101: explicit names win; else one adapter per language / domain of the task; else 'default'"""
    if explicit is not None:
        return list(dict.fromkeys(explicit)), False
    if by_keys:
        names = []
        for key in by_keys:
            prefix = _DOMAIN_PREFIX if key == 'domain' else _LANG_PREFIX
            names += [f'{prefix}{value}' for value in sorted(task.get_langs_or_domains(key))]
            batch_by.add(key)
        return list(dict.fromkeys(names)), True
    return ['default'], False


def _names_in_use(by_keys, meta) -> list:
    return [f"{_DOMAIN_PREFIX if key == 'domain' else _LANG_PREFIX}{meta[key]}" for key in by_keys]


class _AdapterLayerMixin:
    """a Transformer layer followed by its stack of adapters (adapters.py:232-301)"""

    def _build_adapters(self, cfg, dim: int, adapter_names):
        adapter_names = adapter_names or []
        self.adapters = nn.ModuleDict({
            uid: AdapterLayer(cfg.embed_dim, dim, zero_init=cfg.adapter_zero_init, activation_fn='relu')
            for uid in adapter_names})
        self.adapters_in_use = adapter_names

    def _apply_adapters(self, x: Tensor) -> Tensor:
        for uid in self.adapters_in_use:
            x = self.adapters[uid](x)
        return x


class AdapterTransformerEncoderLayer(_AdapterLayerMixin, TransformerEncoderLayer):
    def __init__(self, cfg, dist_cfg, layer_id: int, adapter_names):
        super().__init__(cfg, dist_cfg, layer_id)
        self._build_adapters(cfg, cfg.encoder_adapter_dim, adapter_names)

    def forward(self, x: Tensor, *args, **kwargs):
        x, layer_outputs = super().forward(x, *args, **kwargs)
        return self._apply_adapters(x), layer_outputs


class AdapterTransformerDecoderLayer(_AdapterLayerMixin, TransformerDecoderLayer):
    def __init__(self, cfg, dist_cfg, layer_id: int, adapter_names):
        super().__init__(cfg, dist_cfg, layer_id)
        self._build_adapters(cfg, cfg.decoder_adapter_dim, adapter_names)

    def forward(self, x: Tensor, *args, **kwargs):
        x, layer_outputs = super().forward(x, *args, **kwargs)
        return self._apply_adapters(x), layer_outputs


class AdapterTransformerEncoder(TransformerEncoder):
    def __init__(self, *args, adapter_names=None, **kwargs):
        self.adapter_names = adapter_names
        super().__init__(*args, **kwargs)

    def build_layer(self, layer_id: int) -> nn.Module:
        ids = self.cfg.encoder_adapter_layer_ids
        if (ids is None or layer_id in ids) and self.cfg.encoder_adapter_dim:
            layer = AdapterTransformerEncoderLayer(self.cfg, self.dist_cfg, layer_id, self.adapter_names)
        else:
            layer = TransformerEncoderLayer(self.cfg, self.dist_cfg, layer_id)
        return modules.checkpoint_wrapper(layer, activate=self.cfg.checkpoint_activations)

    def forward(self, *args, **kwargs):
        in_use = _names_in_use(self.cfg.encoder_adapters_by, kwargs.get('meta') or {})
        if in_use:
            for layer in self.layers:
                layer.adapters_in_use = in_use
        return super().forward(*args, **kwargs)


class AdapterTransformerDecoder(TransformerDecoder):
    def __init__(self, *args, adapter_names=None, **kwargs):
        self.adapter_names = adapter_names
        super().__init__(*args, **kwargs)

    def build_layer(self, layer_id: int) -> nn.Module:
        ids = self.cfg.decoder_adapter_layer_ids
        if (ids is None or layer_id in ids) and self.cfg.decoder_adapter_dim:
            layer = AdapterTransformerDecoderLayer(self.cfg, self.dist_cfg, layer_id, self.adapter_names)
        else:
            layer = TransformerDecoderLayer(self.cfg, self.dist_cfg, layer_id)
        return modules.checkpoint_wrapper(layer, activate=self.cfg.checkpoint_activations)

    def forward(self, *args, **kwargs):
        in_use = _names_in_use(self.cfg.decoder_adapters_by, kwargs.get('meta') or {})
        if in_use:
            for layer in self.layers:
                layer.adapters_in_use = in_use
        return super().forward(*args, **kwargs)


@register_model('adapter_transformer')
class AdapterTransformer(Transformer):
    """adapters.py:37-166"""

    def __init__(self, cfg, dist_cfg, task):
        batch_by = set()
        self.encoder_adapter_names, enc_by = _adapter_names(cfg.encoder_adapters, cfg.encoder_adapters_by, task, batch_by)
        self.decoder_adapter_names, dec_by = _adapter_names(cfg.decoder_adapters, cfg.decoder_adapters_by, task, batch_by)
        super().__init__(cfg, dist_cfg, task)
        # after super().__init__, which sets its own defaults for these two
        self.batch_by = sorted(batch_by)
        self.find_unused_parameters = enc_by or dec_by
        if not cfg.train_all_params:  # the usual adapter setup: everything but the adapters is frozen
            for name, param in self.named_parameters():
                if 'adapters' not in name.split('.'):
                    param.requires_grad = False
        self.extra_adapters = {}  # adapters of the checkpoint that this model instance does not use

    def build_encoder(self, embed=None):
        if self.cfg.model_type == 'decoder':
            return DummyEncoder()
        return AdapterTransformerEncoder(self.cfg, self.dist_cfg, self.task, embed=embed,
                                         adapter_names=self.encoder_adapter_names)

    def build_decoder(self, embed=None):
        return AdapterTransformerDecoder(self.cfg, self.dist_cfg, self.task, embed=embed,
                                         adapter_names=self.decoder_adapter_names)

    def update_state_dict(self, state_dict: dict) -> None:
        super().update_state_dict(state_dict)
        if self.training:  # new adapters start from their initialisation
            modules.add_missing_parameters(self, state_dict, _ADAPTER_REGEX)
        # adapters of other languages / domains in the checkpoint are kept aside and written back on save
        self.extra_adapters.update(modules.remove_unused_parameters(self, state_dict, _ADAPTER_REGEX))

    def clean_state_dict(self, state_dict: dict) -> None:
        super().clean_state_dict(state_dict)
        state_dict.update(self.extra_adapters)
