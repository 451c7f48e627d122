"""Deterministic parameter / batch generator shared by `oracle/make_golden.py` (which loads the
values into the *reference* model) and by the tests (which load the same values into the oracle and
into the HIP-backed model).  Golden fixtures therefore only need to store names, shapes and outputs,
never the weights themselves.

Pure numpy: the stream of `np.random.RandomState(seed)` is stable across numpy versions.
"""
import zlib
import numpy as np


def _rs(seed: int, name: str) -> np.random.RandomState:
    # one independent stream per tensor, so that adding / removing a tensor does not shift the others
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2**31 - 1))


def make_param(seed: int, name: str, shape: tuple) -> np.ndarray:
    """fp32 value for parameter `name` of shape `shape`"""
    rs = _rs(seed, name)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    if 'layer_norm' in name or 'layernorm' in name:
        if leaf == 'weight':
            return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.05 * rs.standard_normal(shape)).astype(np.float32)
    if 'embed_tokens' in name or 'embed_positions' in name:
        return (rs.standard_normal(shape) * shape[-1] ** -0.5).astype(np.float32)
    if len(shape) == 3:  # Conv1d weight (out, in, k)
        fan_in = shape[1] * shape[2]
        return (rs.standard_normal(shape) * fan_in ** -0.5).astype(np.float32)
    if len(shape) == 2:  # Linear weight (out, in)
        return (rs.standard_normal(shape) * shape[1] ** -0.5).astype(np.float32)
    return (0.05 * rs.standard_normal(shape)).astype(np.float32)


def make_state_dict(seed: int, names_shapes) -> dict:
    """names_shapes: iterable of (name, shape). Returns {name: np.ndarray fp32}"""
    return {name: make_param(seed, name, shape) for name, shape in names_shapes}


def make_text_batch(seed: int, B: int, S: int, T: int, V: int, ragged: bool = True,
                    pad: int = 1, eos: int = 2, min_frac: float = 0.5, prompt_cols: int = 0) -> dict:
    """Synthetic padded (B, S) / (B, T+1) batch following SURVEY §8d: ids ~ U[4, V), last source token and
    first/last decoder tokens = EOS/BOS (2), pad = 1, `prompt_mask[:, 0] = True`.
    `decoder_input` has T+1 columns (BOS + T targets) like the batches of `pasero/tasks/task.py:564-571`."""
    rs = np.random.RandomState(seed)
    enc = rs.randint(4, V, size=(B, S)).astype(np.int64)
    dec = rs.randint(4, V, size=(B, T + 1)).astype(np.int64)
    if ragged:
        slen = rs.randint(max(1, int(S * min_frac)), S + 1, size=B)
        tlen = rs.randint(max(1, int(T * min_frac)), T + 1, size=B)
        slen[0] = S  # at least one full-length row
        tlen[B - 1] = T
    else:
        slen = np.full(B, S)
        tlen = np.full(B, T)
    for b in range(B):
        enc[b, slen[b] - 1] = eos
        enc[b, slen[b]:] = pad
        dec[b, 0] = eos  # BOS == EOS == 2
        dec[b, tlen[b]] = eos
        dec[b, tlen[b] + 1:] = pad
    prompt_mask = np.zeros((B, T + 1), dtype=bool)
    prompt_mask[:, 0] = True
    for b in range(B):  # `prompt_cols`: odd rows carry a prompt of up to that many target tokens (cfg.prompt_loss cases)
        if prompt_cols and b % 2 == 1:
            prompt_mask[b, : 1 + min(prompt_cols, max(0, tlen[b] - 1))] = True
    return {
        'encoder_input': enc,
        'encoder_input_length': slen.astype(np.int64),
        'decoder_input': dec,
        'prompt_mask': prompt_mask,
    }


def make_reverse_batch(seed: int, B: int, L: int, lo: int = 4, hi: int = 36, pad: int = 1, eos: int = 2) -> dict:
    """A learnable synthetic translation task (tests/golden/train_curve.npz): the target is the source read backwards.
    Sources are `n` tokens ~ U[lo, hi), n ~ U[L/2, L], + EOS; `decoder_input` = BOS + reversed tokens + EOS (+ pad)."""
    rs = np.random.RandomState(seed)
    n = rs.randint(max(1, L // 2), L + 1, size=B)
    n[0] = L
    enc = np.full((B, L + 1), pad, dtype=np.int64)
    dec = np.full((B, L + 2), pad, dtype=np.int64)
    for b in range(B):
        toks = rs.randint(lo, hi, size=n[b])
        enc[b, :n[b]] = toks
        enc[b, n[b]] = eos
        dec[b, 0] = eos
        dec[b, 1:n[b] + 1] = toks[::-1]
        dec[b, n[b] + 1] = eos
    prompt_mask = np.zeros((B, L + 2), dtype=bool)
    prompt_mask[:, 0] = True
    return {'encoder_input': enc, 'encoder_input_length': (n + 1).astype(np.int64), 'decoder_input': dec,
            'prompt_mask': prompt_mask}


def make_array(seed: int, name: str, shape: tuple, scale: float = 1.0) -> np.ndarray:
    return (_rs(seed, name).standard_normal(tuple(shape)) * scale).astype(np.float32)


def make_freeze_mask(seed: int, V: int, frac: float = 0.4) -> np.ndarray:
    """(V,) bool: which source embeddings are frozen (pasero/tasks/translation.py:141-146 builds it from a regex over the
    dictionary; here a seeded random subset)"""
    return _rs(seed, 'freeze_encoder_embed_mask').rand(V) < frac


# the configuration fields every fixture with a `cfg` entry stores (oracle/make_golden.py::cfg_json writes exactly these, plus
# the EXTRA_KEYS that are set): tests/test_oracle_golden.py::test_every_fixture_cfg_has_the_current_schema
CFG_KEYS = [
    'encoder_layers', 'decoder_layers', 'embed_dim', 'encoder_ffn_dim', 'decoder_ffn_dim',
    'encoder_attention_heads', 'decoder_attention_heads', 'dropout', 'attention_dropout', 'activation_dropout',
    'label_smoothing', 'activation_fn', 'encoder_prenorm', 'decoder_prenorm', 'encoder_embed_norm',
    'decoder_embed_norm', 'encoder_positional_encoding', 'decoder_positional_encoding',
    'positional_encoding_shift', 'scale_embed', 'encoder_max_len', 'decoder_max_len', 'shared_embeddings',
    'tied_output_projection', 'rope_base', 'input_dim', 'conv_input_dim', 'conv_channels', 'conv_kernel_sizes',
    'conv_strides', 'conv_activation', 'norm_eps', 'padding_idx', 'eos_idx', 'bos_idx', 'attention_key_bias',
    'has_bias', 'rms_norm', 'norm_bias', 'scale_attn', 'prompt_loss', 'shared_norm',
]


EXTRA_KEYS = ['lora_rank', 'lora_alpha', 'encoder_adapter_dim', 'decoder_adapter_dim', 'adapter_zero_init',
              'train_all_params']
