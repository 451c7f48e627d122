"""Minimal stand-in for the parts of `pasero/config.py` the model path reads, for use OUTSIDE the reference tree
(tests, bench.py, the GPU box where the reference does not exist).

Inside the reference tree nothing here is needed: `pasero_amd.transformer.Transformer` only duck-types its `cfg`
(`pasero.config.TransformerConfig`, config.py:1054-1299), `dist_cfg` (`DistributedConfig`, config.py:500-546) and
`task` arguments, and is registered under the reference's own registry.
Field names and defaults follow the reference; only model hyper-parameters are kept (no argparse/YAML layer).
"""
from dataclasses import dataclass, field
from typing import Optional

MODELS = {}
CONFIGS = {}


def register_model(*names):
    """same role as pasero.config.register_model (config.py:91-101).  Inside the reference tree (`pasero` importable)
    the class is ALSO entered in the reference's own registry, so that `pasero.config.get_architecture` — what
    `pasero-train` / `pasero-decode` call — resolves 'transformer' / 'adapter_transformer' to the classes of this package
    once `pasero_amd` has been imported."""
    def wrapper(cls):
        for name in names:
            MODELS[name] = cls
        try:
            from pasero.config import register_model as reference_register  # type: ignore
        except Exception:  # stand-alone (tests, bench.py, the GPU box)
            reference_register = None
        if reference_register is not None:
            for name in names:
                reference_register(name)(cls)
        return cls
    return wrapper


def register_model_config(*names):
    def wrapper(cls):
        for name in names:
            CONFIGS[name] = cls
        return cls
    return wrapper


def get_architecture(cfg):
    """walk the config class MRO for a registered model name (config.py:103-122)"""
    by_cls = {v: k for k, v in CONFIGS.items()}
    for cls in type(cfg).__mro__:
        for name in (by_cls.get(cls), cls.__dict__.get('_arch')):
            if name in MODELS:
                return MODELS[name]
    return MODELS['transformer']


@dataclass
class DistributedConfig:
    tp_size: Optional[int] = None
    tp_rank: Optional[int] = None
    dp_size: int = 1
    dp_rank: int = 0
    sequence_parallel: bool = False


@register_model_config('transformer')
@dataclass
class TransformerConfig:
    encoder_layers: int = 6
    decoder_layers: int = 6
    shared_embeddings: bool = True
    conv_kernel_sizes: Optional[list] = None
    conv_strides: Optional[list] = None
    conv_activation: Optional[str] = 'glu'
    input_dim: Optional[int] = None
    conv_input_dim: Optional[int] = None
    conv_channels: Optional[int] = None
    embed_dim: int = 512
    encoder_ffn_dim: int = 2048
    decoder_ffn_dim: int = 2048
    encoder_attention_heads: int = 8
    decoder_attention_heads: int = 8
    attention_heads_kv: Optional[int] = None
    sliding_window: Optional[int] = None
    scale_attn: bool = True
    check_inf: bool = False
    attention_key_bias: bool = True
    dropout: float = 0.1
    decoder_dropout: Optional[float] = None
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    label_smoothing: Optional[float] = 0.1     # task default for 'translation' (config.py:1146-1153)
    tied_output_projection: bool = True
    activation_fn: str = 'relu'
    has_bias: bool = True
    encoder_prenorm: bool = False
    decoder_prenorm: Optional[bool] = False
    encoder_embed_norm: bool = False
    decoder_embed_norm: bool = False
    rms_norm: bool = False
    norm_eps: float = 1e-5
    norm_bias: bool = True
    shared_norm: bool = False
    parallel_attention: bool = False
    encoder_positional_encoding: str = 'sinusoidal'
    decoder_positional_encoding: str = 'sinusoidal'
    alibi_max_bias: int = 8
    rope_base: int = 10000
    max_qkv: Optional[float] = None
    positional_encoding_shift: int = 2
    checkpoint_activations: bool = False
    model_type: Optional[str] = 'encoder_decoder'   # task default for 'translation' (config.py:1241-1248)
    prompt_loss: float = 1.0
    scale_embed: bool = True
    embed_dropout: Optional[float] = None
    encoder_max_len: int = 256
    decoder_max_len: Optional[int] = 256            # task default for 'translation' (config.py:1265-1272)
    lora_rank: int = 0
    lora_alpha: int = 8
    padding_idx: int = 1
    bos_idx: int = 2
    eos_idx: int = 2
    unk_idx: int = 3


@register_model_config('transformer_big', 'transformer_wmt_en_de_big', 'transformer_vaswani_wmt_en_de_big')
@dataclass
class TransformerBigConfig(TransformerConfig):   # config.py:2182-2188
    embed_dim: int = 1024
    encoder_ffn_dim: int = 4096
    decoder_ffn_dim: int = 4096
    encoder_attention_heads: int = 16
    decoder_attention_heads: int = 16


@register_model_config('transformer_small', 'transformer_iwslt_de_en')
@dataclass
class TransformerSmallConfig(TransformerConfig):  # config.py:2195-2201
    encoder_ffn_dim: int = 1024
    decoder_ffn_dim: int = 1024
    encoder_attention_heads: int = 4
    decoder_attention_heads: int = 4


@register_model_config('nllb_600m')
@dataclass
class NLLB600MConfig(TransformerBigConfig):      # config.py:2225-2230
    encoder_layers: int = 12
    decoder_layers: int = 12
    encoder_prenorm: bool = True
    decoder_prenorm: bool = True


@register_model_config('nllb_1b3')
@dataclass
class NLLB1B3Config(NLLB600MConfig):             # config.py:2232-2237
    encoder_layers: int = 24
    decoder_layers: int = 24
    encoder_ffn_dim: int = 8192
    decoder_ffn_dim: int = 8192


@register_model_config('whisper_base')
@dataclass
class WhisperConfig(TransformerConfig):          # config.py:2540-2560
    encoder_prenorm: bool = True
    decoder_prenorm: bool = True
    activation_fn: str = 'gelu'
    encoder_positional_encoding: str = 'learned'
    decoder_positional_encoding: str = 'learned'
    positional_encoding_shift: int = 0
    scale_embed: bool = False
    input_dim: int = 80
    conv_input_dim: int = 80
    conv_channels: int = 512
    conv_kernel_sizes: list = field(default_factory=lambda: [3, 3])
    conv_strides: list = field(default_factory=lambda: [1, 2])
    conv_activation: str = 'gelu'
    encoder_max_len: int = 3000
    decoder_max_len: int = 448
    attention_key_bias: bool = False
    padding_idx: int = 50256
    eos_idx: int = 50257
    bos_idx: int = 50258


@dataclass
class _AdapterOptions:   # config.py:1322-1383 (the options `adapter_transformer` adds to a backbone configuration)
    encoder_adapter_dim: int = 64
    decoder_adapter_dim: int = 64
    encoder_adapter_layer_ids: Optional[list] = None
    decoder_adapter_layer_ids: Optional[list] = None
    encoder_adapters: Optional[list] = None
    decoder_adapters: Optional[list] = None
    encoder_adapters_by: list = field(default_factory=list)
    decoder_adapters_by: list = field(default_factory=list)
    adapter_zero_init: bool = False
    train_all_params: bool = False
    _arch = 'adapter_transformer'  # the model every configuration with these options resolves to


@register_model_config('adapter_transformer')
@dataclass
class AdapterTransformerConfig(_AdapterOptions, TransformerConfig):
    pass


@register_model_config('transformer_wide')
@dataclass
class TransformerWideConfig(TransformerBigConfig):   # config.py:2190-2193
    encoder_ffn_dim: int = 8192
    decoder_ffn_dim: int = 8192


@register_model_config('mbart_large')
@dataclass
class MBARTConfig(TransformerBigConfig):             # config.py:2215-2226
    encoder_layers: int = 12
    decoder_layers: int = 12
    encoder_embed_norm: bool = True
    decoder_embed_norm: bool = True
    encoder_positional_encoding: str = 'learned'
    decoder_positional_encoding: str = 'learned'
    encoder_prenorm: bool = True
    decoder_prenorm: bool = True
    encoder_max_len: int = 1024
    decoder_max_len: int = 1024


@register_model_config('nllb_3b3')
@dataclass
class NLLB3B3Config(NLLB1B3Config):                  # config.py:2242-2244
    embed_dim: int = 2048


@register_model_config('whisper_large')
@dataclass
class WhisperLargeConfig(WhisperConfig):             # config.py:2568-2577
    encoder_layers: int = 32
    decoder_layers: int = 32
    embed_dim: int = 1280
    conv_channels: int = 1280
    encoder_ffn_dim: int = 5120
    decoder_ffn_dim: int = 5120
    encoder_attention_heads: int = 20
    decoder_attention_heads: int = 20


# adapter_* presets (config.py:2462-2509): the adapter options on top of each backbone preset
@register_model_config('adapter_transformer_big')
@dataclass
class AdapterTransformerBigConfig(_AdapterOptions, TransformerBigConfig):
    pass


@register_model_config('adapter_transformer_small')
@dataclass
class AdapterTransformerSmallConfig(_AdapterOptions, TransformerSmallConfig):
    pass


@register_model_config('adapter_transformer_wide')
@dataclass
class AdapterTransformerWideConfig(_AdapterOptions, TransformerWideConfig):
    pass


@register_model_config('adapter_nllb_600m')
@dataclass
class AdapterNLLB600MConfig(_AdapterOptions, NLLB600MConfig):
    pass


@register_model_config('adapter_nllb_1b3')
@dataclass
class AdapterNLLB1B3Config(_AdapterOptions, NLLB1B3Config):
    pass


@register_model_config('adapter_nllb_3b3')
@dataclass
class AdapterNLLB3B3Config(_AdapterOptions, NLLB3B3Config):
    pass


@register_model_config('adapter_mbart_large')
@dataclass
class AdapterMBARTConfig(_AdapterOptions, MBARTConfig):
    pass


class SyntheticTask:
    """What the model constructor reads from a `pasero.tasks.Task` (transformer.py:628-636,786)"""
    freeze_encoder_embed_mask = None

    def __init__(self, encoder_num_embeddings: int, decoder_num_embeddings: Optional[int] = None):
        self.encoder_num_embeddings = encoder_num_embeddings
        self.decoder_num_embeddings = (
            encoder_num_embeddings if decoder_num_embeddings is None else decoder_num_embeddings)

    # vocabulary re-mapping when fine-tuning a model trained with other dictionaries (pasero/tasks/task.py): identity
    def remap_encoder_embed(self, embed):
        return embed

    def remap_decoder_embed(self, embed):
        return embed
