#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference (naver/pasero,
imported read-only from /root/reference) on its PyTorch-CPU fp32 path.

Test infrastructure only. Runs in the build container (the reference never travels to the GPU box; only
the .npz files written here do).  The reference has no tests / golden vectors of its own, so
these fixtures are what pins the oracle (oracle/ref_cpu.py) and, through it, the HIP path.

Two third-party imports of the reference are absent from this image and irrelevant to the model path;
they are stubbed in sys.modules exactly as SURVEY §8c describes: `sacrebleu` (pasero/evaluation.py:8,18)
and `stopes...text_normalizer` (pasero/preprocessing.py:20).

Usage:  python oracle/make_golden.py [--only NAME ...]
"""
import os
import sys
import types
import argparse
import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, OUT)
import paramgen  # noqa: E402


def import_reference():
    sb = types.ModuleType('sacrebleu')
    sbm = types.ModuleType('sacrebleu.metrics')

    class _B:
        TOKENIZERS = {}
    sbm.METRICS = {'BLEU': _B}
    sb.metrics = sbm
    sys.modules['sacrebleu'] = sb
    sys.modules['sacrebleu.metrics'] = sbm
    names = ['stopes', 'stopes.pipelines', 'stopes.pipelines.monolingual',
             'stopes.pipelines.monolingual.utils', 'stopes.pipelines.monolingual.utils.text_normalizer']
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    tn = sys.modules[names[-1]]
    tn.remove_non_printing_char = lambda s: s
    tn.replace_unicode_punct = lambda s: s
    sys.path.insert(0, '/root/reference')
    import torch
    from pasero.models import transformer, modules
    from pasero import config, decoding, optimization
    return torch, transformer, modules, config, decoding, optimization


torch, transformer, modules, config, decoding, optimization = import_reference()
torch.manual_seed(0)
torch.set_num_threads(8)


class FakeTask:
    freeze_encoder_embed_mask = None

    def __init__(self, V):
        self.encoder_num_embeddings = V
        self.decoder_num_embeddings = V


def build_model(V, arch='transformer', freeze_seed=None, **overrides):
    if arch == 'adapter_transformer':
        from pasero.models import adapters
        cfg_cls, model_cls = config.AdapterTransformerConfig, adapters.AdapterTransformer
    else:
        cfg_cls, model_cls = config.TransformerConfig, transformer.Transformer
    cfg = cfg_cls(**overrides)
    # task-dependent defaults (config.py:1146-1153,1241-1248,1265-1272) set by hand as in SURVEY §8c
    if cfg.label_smoothing is None:
        cfg.label_smoothing = 0.1
    cfg.model_type = cfg.model_type or 'encoder_decoder'
    cfg.decoder_max_len = cfg.decoder_max_len or 256
    task = FakeTask(V)
    if freeze_seed is not None:  # pasero/tasks/translation.py:141-146 -> Embedding(freeze_mask=...), modules.py:900-947
        task.freeze_encoder_embed_mask = torch.from_numpy(paramgen.make_freeze_mask(freeze_seed, V))
    model = model_cls(cfg, config.DistributedConfig(), task)
    return cfg, model


def load_params(model, seed):
    sd = model.state_dict()
    names_shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    new = paramgen.make_state_dict(seed, names_shapes)
    # tied tensors appear under several names (e.g. encoder./decoder.embed_tokens.weight when
    # shared_embeddings, transformer.py:151-153): every alias gets the value generated for its FIRST name
    first = {}
    for k, v in sd.items():
        new[k] = new[first.setdefault(v.data_ptr(), k)]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
    return names_shapes


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def npy(x):
    return x.detach().cpu().numpy().copy()  # copy: in-place updates (optimizer) must not alias saved arrays


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez(path, **arrays)
    print(f'wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB')


def names_shapes_arrays(names_shapes):
    return {
        'param_names': np.array([n for n, _ in names_shapes]),
        'param_shapes': np.array([','.join(map(str, s)) for _, s in names_shapes]),
    }


CFG_KEYS, EXTRA_KEYS = paramgen.CFG_KEYS, paramgen.EXTRA_KEYS  # the fixture schema: tests/test_oracle_golden.py checks every file against it


def cfg_json(cfg):
    import json
    d = {k: getattr(cfg, k) for k in CFG_KEYS}
    d.update({k: getattr(cfg, k) for k in EXTRA_KEYS if hasattr(cfg, k) and getattr(cfg, k) not in (0, None, False)})
    return np.array(json.dumps(d))


# ----------------------------------------------------------------------------------------------------------
def gen_encdec(name, V, B, S, T, seed, store_grads='full', arch='transformer', prompt_cols=0, freeze_seed=None, **overrides):
    """Whole Transformer.forward + backward (transformer.py:227-380), encoder (698-752), decoder (831-898)"""
    cfg, model = build_model(V, arch=arch, freeze_seed=freeze_seed, **overrides)
    names_shapes = load_params(model, seed)
    model.train()  # dropout probabilities are 0 in every fixture config, so train() == eval() numerically
    batch = paramgen.make_text_batch(seed, B, S, T, V, prompt_cols=prompt_cols)
    tb = {k: t(v) for k, v in batch.items()}
    loss, logs = model(**tb)
    loss.backward()
    out = {
        'cfg': cfg_json(cfg), 'arch': arch, 'V': V, 'B': B, 'S': S, 'T': T, 'seed': seed,
        'loss': npy(loss), 'logs_loss': logs['loss'], 'logs_nll_loss': logs['nll_loss'],
        'logs_num_tokens': logs['num_tokens'], 'logs_num_lines': logs['num_lines'],
        **names_shapes_arrays(names_shapes),
    }
    if prompt_cols:
        out['prompt_cols'] = prompt_cols
    if freeze_seed is not None:
        out['freeze_seed'] = freeze_seed
    for k in ('prompt_nll_loss', 'num_prompt_tokens'):  # the two-part loss of cfg.prompt_loss != 1 (transformer.py:283-321)
        if k in logs:
            out['logs_' + k] = logs[k]
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    out['frozen_names'] = np.array([k for k, p in model.named_parameters() if not p.requires_grad] or [''])
    out['grad_names'] = np.array(list(grads))
    out['grad_norms'] = np.array([g.double().norm().item() for g in grads.values()])  # fp64: the fp32 CPU norm of a
    # 4M-element tensor is off by 4e-4
    out['grad_sums'] = np.array([g.double().sum().item() for g in grads.values()])
    if store_grads == 'full':
        for k, g in grads.items():
            out['grad:' + k] = npy(g)
    else:  # strided sample only (large models)
        for k, g in grads.items():
            out['gradsample:' + k] = npy(g.reshape(-1)[::store_grads])
    with torch.no_grad():
        model.eval()
        enc_out, enc_mask, _ = model.encoder(tb['encoder_input'], tb['encoder_input_length'])
        logits, _ = model.decoder(enc_out, enc_mask, tb['decoder_input'][:, :-1])
    if store_grads == 'full':
        out['encoder_out'] = npy(enc_out)
        out['encoder_mask'] = npy(enc_mask)
        out['logits'] = npy(logits)
    else:
        out['encoder_out_sample'] = npy(enc_out.reshape(-1)[::store_grads])
        out['logits_sample'] = npy(logits.reshape(-1)[::store_grads])
    out['argmax'] = npy(logits.argmax(-1))
    save(name, **out)
    return cfg, model, tb


def gen_tiny_post():
    gen_encdec('tiny_encdec_post', V=101, B=3, S=7, T=5, seed=11,
               embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=192, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=2, decoder_layers=2, dropout=0.0)


def gen_tiny_pre():
    # whisper/NLLB-like variant: pre-norm, GELU(erf), learned positions, no embed scaling, embed layer norms
    gen_encdec('tiny_encdec_pre', V=67, B=4, S=9, T=6, seed=12,
               embed_dim=128, encoder_ffn_dim=160, decoder_ffn_dim=160, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0,
               encoder_prenorm=True, decoder_prenorm=True, activation_fn='gelu',
               encoder_positional_encoding='learned', decoder_positional_encoding='learned',
               positional_encoding_shift=0, scale_embed=False, encoder_embed_norm=True,
               decoder_embed_norm=True, label_smoothing=0.2, encoder_max_len=32, decoder_max_len=32)


def gen_tiny_freeze():
    """partially frozen source embeddings (`freeze_encoder_embed_regex`: Embedding(freeze_mask=...), modules.py:900-947): rows
    of the mask come from a second table `frozen_embedding`, the others from `weight`; separate source / target embeddings"""
    gen_encdec('tiny_freeze_embed', V=89, B=3, S=8, T=6, seed=17, freeze_seed=5,
               embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=192, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0, shared_embeddings=False)


def gen_tiny_freeze_shared():
    """partially frozen embeddings SHARED by encoder and decoder (transformer.py:151-153: the decoder reuses the encoder's
    Embedding object, so its lookup and its tied projection blend the two tables as well, modules.py:929-946) — plus the greedy
    tokens of `decoding.sample_on_the_fly` on that model: the incremental-decoding path must read the blended table too"""
    name = 'tiny_freeze_shared'
    cfg, model, tb = gen_encdec(name, V=89, B=3, S=8, T=6, seed=19, freeze_seed=7,
                                embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=192, encoder_attention_heads=2,
                                decoder_attention_heads=2, encoder_layers=1, decoder_layers=2, dropout=0.0,
                                shared_embeddings=True)
    assert model.decoder.embed_tokens is model.encoder.embed_tokens and model.decoder.embed_tokens.frozen_embedding is not None
    model.eval()
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(tb['encoder_input'], tb['encoder_input_length'])
        bos = torch.full((3, 1), cfg.bos_idx, dtype=torch.long)
        steps = [npy(o['tokens']).copy() for o in decoding.sample_on_the_fly(
            model.decoder, enc_out, enc_mask, 12, {}, decoder_input=bos, sampling_temperature=0)]
    path = os.path.join(OUT, name + '.npz')
    out = dict(np.load(path, allow_pickle=False))
    out['greedy_tokens'] = np.concatenate(steps, axis=1)
    out['max_output_len'] = 12
    save(name, **out)


def gen_tiny_adapter():
    """adapter_transformer (adapters.py:37-301): bottleneck adapters (LayerNorm -> down -> ReLU -> up -> + residual,
    modules.py:248-370) after every layer, frozen backbone — the IWSLT2023 fine-tuning setup"""
    gen_encdec('tiny_adapter', V=73, B=3, S=8, T=6, seed=15, arch='adapter_transformer',
               embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=192, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=2, decoder_layers=2, dropout=0.0,
               encoder_adapter_dim=16, decoder_adapter_dim=24)


def gen_tiny_lora():
    """LoRA branches on every Linear (modules.py:67-100, rank 4, alpha 8), pre-norm GELU stack, everything trained"""
    gen_encdec('tiny_lora', V=71, B=3, S=7, T=6, seed=16,
               embed_dim=128, encoder_ffn_dim=160, decoder_ffn_dim=160, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=2, dropout=0.0, lora_rank=4, lora_alpha=8,
               encoder_prenorm=True, decoder_prenorm=True, activation_fn='gelu')


def gen_tiny_lora_rotary():
    """LoRA branches together with rotary positions: the rotation is applied to the separate q / k projections"""
    gen_encdec('tiny_lora_rotary', V=71, B=3, S=7, T=6, seed=25,
               embed_dim=128, encoder_ffn_dim=160, decoder_ffn_dim=160, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0, lora_rank=4, lora_alpha=8,
               encoder_positional_encoding='rotary', decoder_positional_encoding='rotary')


def gen_tiny_hd128():
    """heads of 128 (transformer_small = transformer_iwslt_de_en: 4 x 128; nllb_3b3: 16 x 128)"""
    gen_encdec('tiny_hd128', V=79, B=3, S=9, T=7, seed=17,
               embed_dim=256, encoder_ffn_dim=128, decoder_ffn_dim=128, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0)


def gen_base_c1():
    """BASELINE config C1: `transformer` base 6+6 d=512 H=8 f=2048 V=8032, batch 8x(64,64), ragged.
    Weights are regenerated from the seed (193 MB), so only scalars / samples are stored."""
    gen_encdec('base_c1', V=8032, B=8, S=64, T=64, seed=21, store_grads=4099, dropout=0.0)


# ----------------------------------------------------------------------------------------------------------
def gen_mha():
    """modules.MultiheadAttention.forward (modules.py:579-739): self w/ key-padding mask, causal self where a
    2-D mask is dropped (modules.py:602-605), cross attention with ragged source; plus return_attn weights."""
    d, H, B, T, S = 128, 2, 3, 10, 13
    out = {}
    for variant in ('self_pad', 'self_causal', 'cross'):
        causal = variant == 'self_causal'
        mha = modules.MultiheadAttention(d, H, dropout=0.0, causal=causal)
        sd = mha.state_dict()
        ns = [(k, tuple(v.shape)) for k, v in sd.items()]
        mha.load_state_dict({k: t(v) for k, v in paramgen.make_state_dict(31, ns).items()})
        q = t(paramgen.make_array(31, variant + '.q', (B, T, d))).requires_grad_()
        if variant == 'cross':
            kv = t(paramgen.make_array(31, variant + '.kv', (B, S, d))).requires_grad_()
            lens = torch.tensor([S, 7, 1])
            src = S
        else:
            kv = q
            lens = torch.tensor([T, 6, 3])
            src = T
        mask = torch.arange(src)[None] >= lens[:, None]
        y, w = mha(query=q, key=kv, value=kv, attn_mask=mask)
        assert w is None
        dy = t(paramgen.make_array(31, variant + '.dy', (B, T, d)))
        y.backward(dy)
        out[variant + ':y'] = npy(y)
        out[variant + ':lens'] = npy(lens)
        out[variant + ':dq'] = npy(q.grad)
        if variant == 'cross':
            out[variant + ':dkv'] = npy(kv.grad)
        for k, p in mha.named_parameters():
            out[variant + ':grad:' + k] = npy(p.grad)
        with torch.no_grad():
            y2, w2 = mha(query=q, key=kv, value=kv, attn_mask=mask, return_attn=True)
        out[variant + ':attn_weights'] = npy(w2)  # (B, T, H, S)
        out[variant + ':y_return_attn'] = npy(y2)
        out[variant + ':param_names'] = np.array([n for n, _ in ns])
        out[variant + ':param_shapes'] = np.array([','.join(map(str, s)) for _, s in ns])
    save('mha', d=d, H=H, B=B, T=T, S=S, **out)


def gen_mha_rotary():
    """MultiheadAttention with positional_encoding='rotary' (modules.py:554-555,621-623,950-1025): causal self-attention,
    full sequence and the same sequence decoded incrementally (position offset = cached length)"""
    d, H, B, T = 128, 2, 2, 9
    mha = modules.MultiheadAttention(d, H, dropout=0.0, causal=True, positional_encoding='rotary')
    sd = mha.state_dict()
    ns = [(k, tuple(v.shape)) for k, v in sd.items()]
    mha.load_state_dict({k: t(v) for k, v in paramgen.make_state_dict(33, ns).items()})
    x = t(paramgen.make_array(33, 'rot.x', (B, T, d))).requires_grad_()
    y, _ = mha(query=x, key=x, value=x)
    dy = t(paramgen.make_array(33, 'rot.dy', (B, T, d)))
    y.backward(dy)
    out = {'y': npy(y), 'dx': npy(x.grad), 'param_names': np.array([n for n, _ in ns]),
           'param_shapes': np.array([','.join(map(str, s)) for _, s in ns])}
    for k, p in mha.named_parameters():
        out['grad:' + k] = npy(p.grad)
    with torch.no_grad():
        state, steps = {}, []
        for i in range(T):
            yi, _ = mha(query=x[:, i:i + 1], key=x[:, i:i + 1], value=x[:, i:i + 1], state=state)
            steps.append(yi)
        out['y_incremental'] = npy(torch.cat(steps, dim=1))
    save('mha_rotary', d=d, H=H, B=B, T=T, **out)


def gen_tiny_rotary():
    gen_encdec('tiny_encdec_rotary', V=59, B=3, S=8, T=6, seed=13,
               embed_dim=128, encoder_ffn_dim=128, decoder_ffn_dim=128, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0,
               encoder_positional_encoding='rotary', decoder_positional_encoding='rotary', activation_fn='gelu_tanh')


def gen_tiny_swiglu():
    gen_encdec('tiny_encdec_swiglu', V=61, B=3, S=7, T=6, seed=14,
               embed_dim=128, encoder_ffn_dim=160, decoder_ffn_dim=160, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0, activation_fn='swiglu',
               encoder_prenorm=True, decoder_prenorm=True)


def gen_tiny_rms():
    """llama-style parameterisation of the encoder-decoder: RMSNorm (modules.py:192-202), pre-norm, rotary positions,
    SwiGLU, no biases anywhere"""
    gen_encdec('tiny_encdec_rms', V=83, B=3, S=9, T=6, seed=18,
               embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=192, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=2, decoder_layers=1, dropout=0.0, activation_fn='swiglu',
               encoder_prenorm=True, decoder_prenorm=True, rms_norm=True, has_bias=False, norm_eps=1e-6,
               encoder_positional_encoding='rotary', decoder_positional_encoding='rotary')


def gen_tiny_opts():
    """configuration switches of the base path that no other fixture turns: untied output projection, separate
    encoder / decoder embeddings, LayerNorm without bias, unscaled attention scores, the two-part prompt loss
    (transformer.py:283-321) — and pre-norm layers sharing one norm (shared_norm), no biases, prompt tokens unscored"""
    gen_encdec('tiny_opts_a', V=89, B=4, S=8, T=7, seed=19, prompt_cols=2,
               embed_dim=128, encoder_ffn_dim=128, decoder_ffn_dim=192, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=2, dropout=0.0, activation_fn='gelu',
               shared_embeddings=False, tied_output_projection=False, norm_bias=False, scale_attn=False,
               prompt_loss=0.5)
    gen_encdec('tiny_opts_b', V=97, B=4, S=9, T=6, seed=20, prompt_cols=3,
               embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=128, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=2, decoder_layers=1, dropout=0.0,
               encoder_prenorm=True, decoder_prenorm=True, shared_norm=True, has_bias=False, prompt_loss=0.0,
               scale_embed=False, label_smoothing=0.0, decoder_positional_encoding='learned',
               encoder_embed_norm=True)


def gen_tiny_hd128_rotary():
    """heads of 128 with rotary positions (the llama-style head size), RMSNorm, SwiGLU, no biases"""
    gen_encdec('tiny_hd128_rotary', V=67, B=3, S=9, T=7, seed=24,
               embed_dim=256, encoder_ffn_dim=192, decoder_ffn_dim=192, encoder_attention_heads=2,
               decoder_attention_heads=2, encoder_layers=1, decoder_layers=1, dropout=0.0, activation_fn='swiglu',
               encoder_prenorm=True, decoder_prenorm=True, rms_norm=True, has_bias=False, norm_eps=1e-6,
               encoder_positional_encoding='rotary', decoder_positional_encoding='rotary')


def gen_ce():
    """Transformer.compute_loss (transformer.py:324-380): label-smoothed CE, sum reduction, pad ignored,
    logs in bits"""
    cfg, model = build_model(37, embed_dim=64, encoder_ffn_dim=64, decoder_ffn_dim=64, encoder_layers=1,
                             decoder_layers=1, encoder_attention_heads=1, decoder_attention_heads=1)
    B, T, V = 4, 16, 8032
    logits_np = paramgen.make_array(41, 'ce.logits', (B, T, V), scale=2.0)
    rs = np.random.RandomState(41)
    target = rs.randint(4, V, size=(B, T)).astype(np.int64)
    target[0, 10:] = 1
    target[2, 3:] = 1
    target[3, :] = 1  # a fully padded row
    out = {'B': B, 'T': T, 'V': V, 'target': target}
    for eps in (0.0, 0.1, 0.2):
        cfg.label_smoothing = eps
        logits = t(logits_np).requires_grad_()
        loss, logs = model.compute_loss(logits, t(target), {})
        loss.backward()
        tag = f'eps{eps}'
        out[tag + ':loss'] = npy(loss)
        out[tag + ':logs_loss'] = logs['loss']
        out[tag + ':logs_nll_loss'] = logs['nll_loss']
        out[tag + ':num_tokens'] = logs['num_tokens']
        out[tag + ':num_lines'] = logs['num_lines']
        out[tag + ':dlogits_rows'] = npy(logits.grad[:, :2])  # (B, 2, V) sample; rest checked by sums
        out[tag + ':dlogits_rowsum'] = npy(logits.grad.sum(-1))
        out[tag + ':dlogits_abs_rowsum'] = npy(logits.grad.abs().sum(-1))
    save('ce_ls', **out)


def gen_sinpos():
    """modules.SinusoidalPositionalEmbedding (modules.py:415-457)"""
    out = {}
    for d in (128, 512, 1024):
        pe = modules.SinusoidalPositionalEmbedding(300, d, shift=2)
        out[f'd{d}'] = npy(pe(40))[0]          # (40, d): positions 2..41
        out[f'd{d}_off'] = npy(pe(3, offset=17))[0]
        out[f'd{d}_rows'] = npy(pe.weight[[0, 1, 2, 150, 301]])
    save('sinpos', **out)


def gen_speech():
    """Speech path of TransformerEncoder.forward (transformer.py:731-744): in_linear (+ReLU) ->
    ConvolutionSubsampler (modules.py:774-834) -> scale -> positions -> layers"""
    for name, ov in (
        ('speech_whisper', dict(input_dim=80, conv_input_dim=80, conv_channels=128, conv_kernel_sizes=[3, 3],
                                conv_strides=[1, 2], conv_activation='gelu', encoder_prenorm=True,
                                decoder_prenorm=True, activation_fn='gelu', scale_embed=False,
                                encoder_positional_encoding='learned', decoder_positional_encoding='learned',
                                positional_encoding_shift=0, encoder_max_len=64, decoder_max_len=32,
                                attention_key_bias=False)),
        ('speech_iwslt', dict(input_dim=96, conv_input_dim=80, conv_channels=256, conv_kernel_sizes=[5],
                              conv_strides=[2], conv_activation='glu', encoder_prenorm=True,
                              decoder_prenorm=True, encoder_max_len=64, decoder_max_len=32)),
    ):
        V, B, S, T, seed = 53, 3, 37, 6, 51
        cfg, model = build_model(V, embed_dim=128, encoder_ffn_dim=128, decoder_ffn_dim=128,
                                 encoder_attention_heads=2, decoder_attention_heads=2, encoder_layers=1,
                                 decoder_layers=1, dropout=0.0, **ov)
        names_shapes = load_params(model, seed)
        model.train()
        feats = t(paramgen.make_array(seed, name + '.feats', (B, S, cfg.input_dim)))
        lens = torch.tensor([S, 20, 9])
        for b in range(B):
            feats[b, lens[b]:] = 0
        tb = paramgen.make_text_batch(seed, B, 4, T, V)
        feats.requires_grad_()
        loss, logs = model(encoder_input=feats, encoder_input_length=lens,
                           decoder_input=t(tb['decoder_input']), prompt_mask=t(tb['prompt_mask']))
        loss.backward()
        out = {'cfg': cfg_json(cfg), 'V': V, 'B': B, 'S': S, 'T': T, 'seed': seed, 'lens': npy(lens),
               'loss': npy(loss), 'logs_loss': logs['loss'], 'logs_nll_loss': logs['nll_loss'],
               'logs_num_tokens': logs['num_tokens'], 'dfeats': npy(feats.grad),
               **names_shapes_arrays(names_shapes)}
        for k, p in model.named_parameters():
            if 'subsample' in k or 'in_linear' in k:
                out['grad:' + k] = npy(p.grad)
        out['grad_names'] = np.array([k for k, _ in model.named_parameters()])
        out['grad_norms'] = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
        with torch.no_grad():
            x = feats.detach()
            if model.encoder.in_linear is not None:
                x = model.encoder.in_linear(x)
            sub, new_len = model.encoder.subsample(x, lens)
            enc_out, enc_mask, _ = model.encoder(feats.detach(), lens)
        out['subsample_out'] = npy(sub)
        out['new_len'] = npy(new_len)
        out['encoder_out'] = npy(enc_out)
        out['encoder_mask'] = npy(enc_mask)
        save(name, **out)


def gen_greedy():
    """decoding.sample_on_the_fly greedy path (decoding.py:1005-1221) with the decoder's incremental `state`
    (transformer.py:869-872,1263-1289; modules.py:625-641)"""
    V, B, S, seed = 101, 3, 7, 11
    cfg, model = build_model(V, embed_dim=128, encoder_ffn_dim=192, decoder_ffn_dim=192,
                             encoder_attention_heads=2, decoder_attention_heads=2, encoder_layers=2,
                             decoder_layers=2, dropout=0.0)
    names_shapes = load_params(model, seed)
    model.eval()
    batch = paramgen.make_text_batch(seed, B, S, 5, V)
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(t(batch['encoder_input']), t(batch['encoder_input_length']))
        steps = []
        # decoder_input given explicitly: EnsembleDecoder has no `bos_idx` (decoding.py:1068 would raise)
        bos = torch.full((B, 1), cfg.bos_idx, dtype=torch.long)
        for o in decoding.sample_on_the_fly(model.decoder, enc_out, enc_mask, 12, {}, decoder_input=bos,
                                            sampling_temperature=0):
            steps.append(npy(o['tokens']).copy())
    tokens = np.concatenate(steps, axis=1)
    save('greedy_decode', cfg=cfg_json(cfg), V=V, B=B, S=S, seed=seed, max_output_len=12, tokens=tokens,
         n_steps=len(steps), **names_shapes_arrays(names_shapes))


def gen_return_layers():
    """`return_layers` of the encoder / decoder (transformer.py:698-752,831-898; layers :1030-1099,1263-1417): hidden
    states after a layer ('enc_0', 'dec_1') and attention weights ('enc_0_self_attn', 'dec_0_self_attn',
    'dec_1_cross_attn'), full pass and one incremental step"""
    _gen_return_layers('return_layers')
    _gen_return_layers('return_layers_rotary', encoder_positional_encoding='rotary',
                       decoder_positional_encoding='rotary')


def _gen_return_layers(name, **overrides):
    V, B, S, T, seed = 71, 3, 8, 6, 23
    cfg, model = build_model(V, embed_dim=128, encoder_ffn_dim=128, decoder_ffn_dim=128, encoder_attention_heads=2,
                             decoder_attention_heads=2, encoder_layers=2, decoder_layers=2, dropout=0.0, **overrides)
    names_shapes = load_params(model, seed)
    model.eval()
    batch = paramgen.make_text_batch(seed, B, S, T, V)
    enc_names = ['enc_0', 'enc_1_self_attn']
    dec_names = ['dec_1', 'dec_0_self_attn', 'dec_1_cross_attn']
    out = {'cfg': cfg_json(cfg), 'V': V, 'B': B, 'S': S, 'T': T, 'seed': seed, 'enc_names': np.array(enc_names),
           'dec_names': np.array(dec_names), **names_shapes_arrays(names_shapes)}
    with torch.no_grad():
        enc_out, enc_mask, enc_layers = model.encoder(t(batch['encoder_input']), t(batch['encoder_input_length']),
                                                      return_layers=enc_names)
        dec_in = t(batch['decoder_input'])[:, :-1]
        logits, dec_layers = model.decoder(enc_out, enc_mask, dec_in, return_layers=dec_names)
        assert sorted(enc_layers) == sorted(enc_names) and sorted(dec_layers) == sorted(dec_names)
        for k, v in {**enc_layers, **dec_layers}.items():
            out['full:' + k] = npy(v)
        out['logits'] = npy(logits)
        state = {}
        model.decoder(enc_out, enc_mask, dec_in[:, :3], state=state)
        _, step_layers = model.decoder(enc_out, enc_mask, dec_in[:, 3:4], state=state, return_layers=dec_names)
        for k, v in step_layers.items():
            out['step:' + k] = npy(v)
    save(name, **out)


def gen_beam():
    """decoding.beam_search (decoding.py:1225-1657) as a TRACE: what the search feeds the decoder at every step
    (tokens, incremental `state`) and the beam re-ordering it applies to the state (`Decoder.reorder_state`, including the
    steps where finished sentences leave the batch), with the last-position logits the reference decoder returned.
    The GPU test replays the trace through pasero_amd's decoder and `reorder_state`; the search itself stays the
    reference's Python."""
    V, B, S, K, max_out = 61, 4, 7, 3, 10
    cfg, model = build_model(V, embed_dim=128, encoder_ffn_dim=128, decoder_ffn_dim=128,
                             encoder_attention_heads=2, decoder_attention_heads=2, encoder_layers=1,
                             decoder_layers=2, dropout=0.0)
    for seed in range(40, 80):  # first seed whose search drops finished sentences from the batch on the way
        names_shapes = load_params(model, seed)
        model.eval()
        batch = paramgen.make_text_batch(seed, B, S, 5, V)
        trace = {'dec_in': [], 'logits': [], 'reorder': [], 'rows': []}
        dec = model.decoder
        inner_forward = dec.forward

        def spy_forward(encoder_out, encoder_mask, decoder_input, **kw):
            out = inner_forward(encoder_out, encoder_mask, decoder_input, **kw)
            trace['dec_in'].append(npy(decoder_input))
            trace['logits'].append(npy(out[0][:, -1].float()))
            trace['rows'].append(encoder_out.size(0))
            return out
        inner_reorder = transformer.Decoder.reorder_state

        def spy_reorder(state, indices):
            trace['reorder'].append(npy(indices))
            return inner_reorder(state, indices)
        dec.forward = spy_forward
        decoding.Decoder.reorder_state = staticmethod(spy_reorder)
        try:
            with torch.no_grad():
                enc_out, enc_mask, _ = model.encoder(t(batch['encoder_input']), t(batch['encoder_input_length']))
                # decoder_input given explicitly: EnsembleDecoder has no `bos_idx` (decoding.py:1305 would raise)
                bos = torch.full((B, 1), cfg.bos_idx, dtype=torch.long)
                hyps = decoding.beam_search(dec, enc_out, enc_mask, max_out, K, {}, decoder_input=bos)
        finally:
            dec.forward = inner_forward
            decoding.Decoder.reorder_state = staticmethod(inner_reorder)
        shrinks = len(set(trace['rows'])) > 1
        print(f'seed {seed}: {len(trace["dec_in"])} decoder calls, rows per call {trace["rows"]}')
        if shrinks and len(trace['dec_in']) >= 5:
            break
    else:
        raise RuntimeError('no seed gave a search with a shrinking batch')
    out = {'cfg': cfg_json(cfg), 'V': V, 'B': B, 'S': S, 'K': K, 'seed': seed, 'max_output_len': max_out,
           'n_calls': len(trace['dec_in']), 'n_reorders': len(trace['reorder']), **names_shapes_arrays(names_shapes)}
    for i, (d, lg) in enumerate(zip(trace['dec_in'], trace['logits'])):
        out[f'dec_in_{i}'] = d
        out[f'logits_{i}'] = lg
    for i, r in enumerate(trace['reorder']):
        out[f'reorder_{i}'] = r
    for b, nbest in enumerate(hyps):
        out[f'best_{b}'] = npy(nbest[0]['tokens'])
    save('beam_trace', **out)


def gen_optim():
    """K9 ("next" row): optimization.Adam.step (optimization.py:56-149), clip_grad_norm_ (390-427),
    LRScheduler inverse-sqrt (21-52)"""
    rs = np.random.RandomState(61)
    shapes = [(33, 17), (129,), (8, 8, 3)]
    params = [torch.nn.Parameter(t(rs.standard_normal(s).astype(np.float32))) for s in shapes]
    grads = [[rs.standard_normal(s).astype(np.float32) for s in shapes] for _ in range(3)]
    out = {'n': len(shapes), 'steps': 3}
    for i, p in enumerate(params):
        out[f'p0:{i}'] = npy(p)
    opt = optimization.Adam(params, lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    for step in range(3):
        for i, p in enumerate(params):
            p.grad = t(grads[step][i]).clone()
            out[f'g{step}:{i}'] = grads[step][i]
        gnorm = optimization.clip_grad_norm_(params, 1.0)
        out[f'gnorm{step}'] = float(gnorm)
        opt.step()
        for i, p in enumerate(params):
            out[f'p{step + 1}:{i}'] = npy(p)
    save('adam_step', **out)


def gen_train_curve():
    """north_star: "loss curve matching CPU reference within 1e-3 on TED de-en" — the corpus is not in the image
    (examples/TED/de-en holds only dict.txt and bpecodes), so the closest thing the REAL reference can be run on here: a
    base-width 2 + 2-layer Transformer over the TED vocabulary size (8028 dict.txt entries + 4 specials = 8032) trained
    for 120 steps on a learnable synthetic translation task (target = source reversed, paramgen.make_reverse_batch), with
    the reference's own training arithmetic — Transformer.forward (loss summed over tokens), gradients / num_tokens
    (training.py:455-470), optimization.clip_grad_norm_ (390-427), optimization.Adam.step (56-149), LRScheduler
    (21-52: linear warm-up, inverse-sqrt decay).  Stored: the per-step loss sum, token count, gradient norm and lr."""
    V, B, L, STEPS = 8032, 32, 20, 120
    hp = dict(lr=3e-4, init_lr=1e-7, min_lr=1e-9, warmup=40, max_steps=STEPS, clip_norm=1.0, betas=(0.9, 0.98), eps=1e-8,
              weight_decay=0.0)
    cfg, model = build_model(V, encoder_layers=2, decoder_layers=2, dropout=0.0)
    ns = load_params(model, 77)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = optimization.Adam(params, lr=hp['lr'], betas=hp['betas'], eps=hp['eps'], weight_decay=hp['weight_decay'])
    tcfg = types.SimpleNamespace(warmup=hp['warmup'], init_lr=hp['init_lr'], lr=hp['lr'], max_steps=hp['max_steps'],
                                 min_lr=hp['min_lr'])
    # (the reference passes `verbose` to torch's scheduler base class, an argument torch 2.10 no longer has: the base
    # constructor is wrapped for the duration of this call so that the reference's own __init__ / get_lr run unchanged)
    base = torch.optim.lr_scheduler._LRScheduler
    base_init = base.__init__
    base.__init__ = lambda self, optimizer, last_epoch=-1, verbose=False: base_init(self, optimizer, last_epoch)
    try:
        sched = optimization.LRScheduler(tcfg, opt)
    finally:
        base.__init__ = base_init
    loss_sum, ntok, gnorms, lrs = [], [], [], []
    for step in range(STEPS):
        b = paramgen.make_reverse_batch(1000 + step, B, L)
        opt.zero_grad(set_to_none=True)
        loss, logs = model(**{k: t(v) for k, v in b.items()})
        loss.backward()
        for p in params:
            if p.grad is not None:
                p.grad.data.mul_(1.0 / logs['num_tokens'])
        gnorm = optimization.clip_grad_norm_(params, hp['clip_norm'])
        lrs.append(sched.get_last_lr()[0])
        opt.step()
        sched.step()
        loss_sum.append(float(loss.item()))
        ntok.append(int(logs['num_tokens']))
        gnorms.append(float(gnorm))
        if step % 20 == 0 or step == STEPS - 1:
            print(f'  step {step}: loss/token {loss_sum[-1] / ntok[-1]:.4f} gnorm {gnorms[-1]:.3f} lr {lrs[-1]:.2e}')
    assert loss_sum[-1] / ntok[-1] < 0.6 * loss_sum[0] / ntok[0], 'the task must be learnable in the stored steps'
    save('train_curve', V=V, B=B, L=L, S=L + 1, T=L + 1, steps=STEPS, seed=77, batch_seed0=1000, cfg=cfg_json(cfg),
         **names_shapes_arrays(ns), loss_sum=np.array(loss_sum, np.float64), num_tokens=np.array(ntok, np.int64), gnorm=np.array(gnorms, np.float64),
         lr=np.array(lrs, np.float64), hp=np.array([hp['lr'], hp['init_lr'], hp['min_lr'], hp['warmup'], hp['clip_norm'],
                                                    hp['betas'][0], hp['betas'][1], hp['eps'], hp['weight_decay']]))


def gen_logmel():
    """K8: third-party arithmetic — transformers.WhisperFeatureExtractor as called from
    examples/Whisper/extract-features.py:107-117 (pinned to the transformers version installed here)"""
    import transformers
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()  # defaults: 80 mel, 16 kHz, n_fft 400, hop 160, chunk 30 s
    rs = np.random.RandomState(0)
    n = 32000
    wav0 = (0.1 * rs.standard_normal(n)).astype(np.float32)
    tt = np.arange(n) / 16000.0
    wav1 = (0.5 * np.sin(2 * np.pi * 1000.0 * tt)).astype(np.float32)
    feats = fe([wav0, wav1], sampling_rate=16000, return_tensors='np')['input_features']  # (2, 80, 3000)
    feats = np.ascontiguousarray(feats.transpose(0, 2, 1))  # extract-features.py:116 -> (3000, 80)
    save('logmel', wav0=wav0, wav1=wav1, feats=feats.astype(np.float32)[:, :240],  # first 240 frames (rest = pad)
         feats_tail=feats.astype(np.float32)[:, -4:],
         mel_filters=np.asarray(fe.mel_filters, dtype=np.float64),
         transformers_version=np.array(transformers.__version__))



def gen_features_file():
    """the speech feature file format and its collate.  The REAL `NumpyFile.build` writes a file of
    ragged fp16 rows (pasero/files.py:122-159, as examples/Whisper/extract-features.py:164 uses it), the real
    `NumpyFile.__next__` / `seek` read it back (files.py:164-192), the real `utils.tokens_as_tensor` collates batches
    of its rows (utils.py:709-736).  Stored: the file's bytes (data the reference wrote), every row as the reference
    reads it, and the collated batches in fp32 / bf16 (bf16 as its 16-bit pattern)."""
    import tempfile
    from pasero.files import NumpyFile
    from pasero import utils
    rs = np.random.RandomState(31)
    lens = [37, 1, 12, 0, 25, 40, 3]  # a zero-length clip sits in the middle (same file position as its successor)
    D = 16
    feats = [(rs.standard_normal((n, D)) * 2.5).astype(np.float32) for n in lens]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'feats.bin')
        # num_feats larger than the number of arrays: trailing empty positions, skipped at load (files.py:113-116)
        f = NumpyFile.build(path, feats, dtype='float16', num_feats=len(feats) + 2)
        out['file_bytes'] = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
        idx, lengths = f.get_positions()
        out['indices'], out['lengths'] = idx, lengths
        rows = [next(f) for _ in range(len(idx))]
        for i, r in enumerate(rows):
            out[f'row{i}'] = r
        f.seek(4)
        out['seek4_tell'] = np.array(f.tell())
        out['seek4_row'] = next(f)
        f.close()
        batches = [[0, 2, 4], [5, 1], [6, 3, 0, 5]]
        out['batches'] = np.array([','.join(map(str, b)) for b in batches])
        for bi, b in enumerate(batches):
            for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
                tokens, ln = utils.tokens_as_tensor([rows[i] for i in b], padding_idx=1, dtype=dt)
                out[f'batch{bi}_{name}'] = (tokens.view(torch.int16).numpy() if dt == torch.bfloat16 else tokens.numpy())
                out[f'batch{bi}_len'] = ln.numpy()
    save('features_file', **out)

GENERATORS = {
    'tiny_encdec_post': gen_tiny_post,
    'tiny_encdec_pre': gen_tiny_pre,
    'base_c1': gen_base_c1,
    'tiny_adapter': gen_tiny_adapter,
    'tiny_freeze_embed': gen_tiny_freeze,
    'tiny_freeze_shared': gen_tiny_freeze_shared,
    'tiny_lora': gen_tiny_lora,
    'tiny_lora_rotary': gen_tiny_lora_rotary,
    'tiny_hd128': gen_tiny_hd128,
    'mha': gen_mha,
    'mha_rotary': gen_mha_rotary,
    'tiny_encdec_rotary': gen_tiny_rotary,
    'tiny_encdec_swiglu': gen_tiny_swiglu,
    'tiny_encdec_rms': gen_tiny_rms,
    'tiny_opts': gen_tiny_opts,
    'tiny_hd128_rotary': gen_tiny_hd128_rotary,
    'ce_ls': gen_ce,
    'sinpos': gen_sinpos,
    'speech': gen_speech,
    'greedy_decode': gen_greedy,
    'beam_trace': gen_beam,
    'return_layers': gen_return_layers,
    'adam_step': gen_optim,
    'train_curve': gen_train_curve,
    'logmel': gen_logmel,
    'features_file': gen_features_file,
}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*')
    args = ap.parse_args()
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only:
            continue
        print('==', name)
        fn()
