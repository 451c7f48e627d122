"""Pins the CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the real reference
(oracle/make_golden.py, PyTorch-CPU fp32).  Runs without a GPU."""
import numpy as np
import pytest
import torch

import paramgen
from conftest import load_golden, golden_cfg, golden_names_shapes
from oracle import ref_cpu as O

RTOL = 2e-5  # fp32 round-off between two orderings of the same sums


def close(a, b, rtol=RTOL, atol=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    err = np.abs(a - b).max() if b.size else 0.0
    assert err <= atol + rtol * scale, f'max abs err {err:.3e} vs scale {scale:.3e}'


def _state(g, seed):
    ns = golden_names_shapes(g)
    return O.to_torch_state(paramgen.make_state_dict(seed, ns))


def _run_encdec(name, full=True):
    g = load_golden(name)
    cfg = golden_cfg(g)
    seed = int(g['seed'])
    P = {k: v.requires_grad_() for k, v in _state(g, seed).items()}
    if cfg.shared_embeddings:  # one tensor under two names (transformer.py:151-153)
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
        if 'decoder.embed_tokens.frozen_embedding.weight' in P:
            P['decoder.embed_tokens.frozen_embedding.weight'] = P['encoder.embed_tokens.frozen_embedding.weight']
    if 'freeze_seed' in g.files:  # the task's (V,) bool mask of frozen source embeddings: an input, not a parameter
        P['encoder.embed_tokens.freeze_mask'] = torch.from_numpy(paramgen.make_freeze_mask(int(g['freeze_seed']), int(g['V'])))
    prompt_cols = int(g['prompt_cols']) if 'prompt_cols' in g.files else 0
    batch = paramgen.make_text_batch(seed, int(g['B']), int(g['S']), int(g['T']), int(g['V']), prompt_cols=prompt_cols)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    loss, logs = O.transformer_forward(P, cfg, **tb)
    loss.backward()
    for k in ('prompt_nll_loss', 'num_prompt_tokens'):  # two-part loss of cfg.prompt_loss != 1
        assert (k in logs) == ('logs_' + k in g.files)
        if k in logs:
            assert abs(logs[k] - float(g['logs_' + k])) <= 1e-5 * abs(float(g['logs_' + k])), k
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    assert abs(logs['loss'] - float(g['logs_loss'])) <= 1e-5 * abs(float(g['logs_loss']))
    assert abs(logs['nll_loss'] - float(g['logs_nll_loss'])) <= 1e-5 * abs(float(g['logs_nll_loss']))
    assert logs['num_tokens'] == int(g['logs_num_tokens'])
    assert logs['num_lines'] == int(g['logs_num_lines'])
    names = [str(n) for n in g['grad_names']]
    norms = g['grad_norms']
    for n, ref_norm in zip(names, norms):
        gr = P[n].grad
        assert gr is not None, n
        # (k_proj.bias gradients are mathematically zero: softmax is shift-invariant -> pure round-off)
        atol = 2e-5 if n.endswith('k_proj.bias') else 2e-6  # (unscaled scores: larger logits, larger round-off)
        assert abs(gr.double().norm().item() - ref_norm) <= 1e-4 * ref_norm + atol, n
        if full and n.endswith('k_proj.bias') and np.abs(g['grad:' + n]).max() < 1e-5:
            assert gr.abs().max().item() < 1e-5, n
        elif full:
            close(gr.numpy(), g['grad:' + n], rtol=1e-4)
        else:
            stride = 4099  # 12 layers deep: fp32 summation-order noise grows to a few 1e-4 relative
            close(gr.reshape(-1)[::stride].numpy(), g['gradsample:' + n], rtol=1e-3)
    with torch.no_grad():
        enc_out, enc_mask = O.encoder(P, cfg, tb['encoder_input'], tb['encoder_input_length'])
        logits = O.decoder(P, cfg, enc_out, enc_mask, tb['decoder_input'][:, :-1])
    if full:
        close(enc_out.numpy(), g['encoder_out'])
        assert (enc_mask.numpy() == g['encoder_mask']).all()
        close(logits.numpy(), g['logits'])
    else:
        close(logits.reshape(-1)[::4099].numpy(), g['logits_sample'], rtol=1e-4)
    assert (logits.argmax(-1).numpy() == g['argmax']).all()  # bit-exact token argmax


def test_tiny_encdec_postnorm():
    _run_encdec('tiny_encdec_post')


def test_tiny_encdec_prenorm_gelu_learned():
    _run_encdec('tiny_encdec_pre')


def test_tiny_encdec_rotary_gelu_tanh():
    _run_encdec('tiny_encdec_rotary')


def test_tiny_encdec_swiglu_prenorm():
    _run_encdec('tiny_encdec_swiglu')


def test_tiny_encdec_rmsnorm_rotary_swiglu_no_bias():
    _run_encdec('tiny_encdec_rms')


def test_tiny_partially_frozen_source_embeddings():
    """Embedding(freeze_mask=...) (modules.py:900-947): rows of the mask read `frozen_embedding.weight`"""
    _run_encdec('tiny_freeze_embed')


def test_tiny_partially_frozen_embeddings_shared_with_the_decoder():
    """shared_embeddings + freeze_mask: the decoder's lookup and tied projection blend the two tables too (the decoder IS the
    encoder's Embedding object, transformer.py:151-153) — loss, gradients, logits, and the reference's greedy tokens through
    the oracle's incremental decoder; with the frozen table ignored on the decoder side the tokens differ (the fixture tells)"""
    _run_encdec('tiny_freeze_shared')
    g = load_golden('tiny_freeze_shared')
    cfg = golden_cfg(g)
    P = _state(g, int(g['seed']))
    for k in ('weight', 'frozen_embedding.weight'):
        P['decoder.embed_tokens.' + k] = P['encoder.embed_tokens.' + k]
    P['encoder.embed_tokens.freeze_mask'] = torch.from_numpy(paramgen.make_freeze_mask(int(g['freeze_seed']), int(g['V'])))
    batch = paramgen.make_text_batch(int(g['seed']), int(g['B']), int(g['S']), int(g['T']), int(g['V']))
    with torch.no_grad():
        enc_out, enc_mask = O.encoder(P, cfg, torch.from_numpy(batch['encoder_input']),
                                      torch.from_numpy(batch['encoder_input_length']))
        tokens = O.greedy_decode(P, cfg, enc_out, enc_mask, int(g['max_output_len']))
        assert (tokens.numpy() == g['greedy_tokens']).all()
        Q = {k: v for k, v in P.items() if k != 'decoder.embed_tokens.frozen_embedding.weight'}
        wrong = O.greedy_decode(Q, cfg, enc_out, enc_mask, int(g['max_output_len']))
        assert wrong.shape != tokens.shape or (wrong.numpy() != g['greedy_tokens']).any()


def test_tiny_heads_of_128_rotary():
    _run_encdec('tiny_hd128_rotary')


@pytest.mark.parametrize('name', ['tiny_opts_a', 'tiny_opts_b'])
def test_tiny_configuration_switches(name):
    """untied projection / unshared embeddings / LayerNorm without bias / unscaled scores / prompt_loss 0.5 (a);
    shared_norm / no biases / prompt_loss 0 / no label smoothing / embedding norm (b)"""
    _run_encdec(name)


def test_mha_rotary_full_and_incremental():
    g = load_golden('mha_rotary')
    d, H, B, T = (int(g[k]) for k in 'dHBT')
    names = [str(n) for n in g['param_names']]
    shapes = [tuple(int(x) for x in str(s).split(',')) for s in g['param_shapes']]
    P = {'a.' + k: v.requires_grad_() for k, v in
         O.to_torch_state(paramgen.make_state_dict(33, list(zip(names, shapes)))).items()}
    x = torch.from_numpy(paramgen.make_array(33, 'rot.x', (B, T, d))).requires_grad_()
    y, _ = O.multihead_attention(P, 'a', x, x, x, H, None, causal=True, rope_base=10000.0)
    y.backward(torch.from_numpy(paramgen.make_array(33, 'rot.dy', (B, T, d))))
    close(y.detach().numpy(), g['y'])
    close(x.grad.numpy(), g['dx'], rtol=1e-4)
    for n in names:
        close(P['a.' + n].grad.numpy(), g['grad:' + n], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        state, steps = {}, []
        for i in range(T):
            yi, _ = O.multihead_attention(P, 'a', x[:, i:i + 1], x[:, i:i + 1], x[:, i:i + 1], H, None, causal=True,
                                          state=state, rope_base=10000.0)
            steps.append(yi)
    close(torch.cat(steps, 1).numpy(), g['y_incremental'])


def test_base_c1():
    _run_encdec('base_c1', full=False)


@pytest.mark.parametrize('variant', ['self_pad', 'self_causal', 'cross'])
def test_mha(variant):
    g = load_golden('mha')
    d, H, B, T, S = (int(g[k]) for k in 'dHBTS')
    names = [str(n) for n in g[variant + ':param_names']]
    shapes = [tuple(int(x) for x in str(s).split(',')) for s in g[variant + ':param_shapes']]
    P = {'a.' + k: v.requires_grad_() for k, v in
         O.to_torch_state(paramgen.make_state_dict(31, list(zip(names, shapes)))).items()}
    q = torch.from_numpy(paramgen.make_array(31, variant + '.q', (B, T, d))).requires_grad_()
    if variant == 'cross':
        kv = torch.from_numpy(paramgen.make_array(31, variant + '.kv', (B, S, d))).requires_grad_()
        src = S
    else:
        kv, src = q, T
    lens = torch.from_numpy(g[variant + ':lens'])
    mask = O.len_to_mask(lens, src)
    y, w = O.multihead_attention(P, 'a', q, kv, kv, H, mask, causal=(variant == 'self_causal'))
    y.backward(torch.from_numpy(paramgen.make_array(31, variant + '.dy', (B, T, d))))
    close(y.detach().numpy(), g[variant + ':y'])
    close(y.detach().numpy(), g[variant + ':y_return_attn'])
    close(w.detach().numpy(), g[variant + ':attn_weights'])
    close(q.grad.numpy(), g[variant + ':dq'], rtol=1e-4)
    if variant == 'cross':
        close(kv.grad.numpy(), g[variant + ':dkv'], rtol=1e-4)
    for n in names:
        close(P['a.' + n].grad.numpy(), g[variant + ':grad:' + n], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('eps', [0.0, 0.1, 0.2])
def test_label_smoothed_ce(eps):
    g = load_golden('ce_ls')
    B, T, V = int(g['B']), int(g['T']), int(g['V'])
    logits = torch.from_numpy(paramgen.make_array(41, 'ce.logits', (B, T, V), scale=2.0)).requires_grad_()
    target = torch.from_numpy(g['target'])
    loss, nll, ntok = O.label_smoothed_ce(logits.view(-1, V), target.view(-1), 1, eps)
    loss.backward()
    tag = f'eps{eps}'
    assert abs(loss.item() - float(g[tag + ':loss'])) <= 1e-5 * abs(float(g[tag + ':loss']))
    assert abs(loss.item() / O.LN2 - float(g[tag + ':logs_loss'])) <= 1e-5 * abs(float(g[tag + ':logs_loss']))
    assert abs(nll.item() / O.LN2 - float(g[tag + ':logs_nll_loss'])) <= 1e-5 * abs(float(g[tag + ':logs_nll_loss']))
    assert int(ntok) == int(g[tag + ':num_tokens'])
    close(logits.grad[:, :2].numpy(), g[tag + ':dlogits_rows'], rtol=1e-4)
    close(logits.grad.sum(-1).numpy(), g[tag + ':dlogits_rowsum'], rtol=1e-4, atol=1e-5)
    close(logits.grad.abs().sum(-1).numpy(), g[tag + ':dlogits_abs_rowsum'], rtol=1e-4)


@pytest.mark.parametrize('d', [128, 512, 1024])
def test_sinusoidal_positions(d):
    g = load_golden('sinpos')
    table = O.sinusoidal_table(300, d, shift=2)
    # fp32 sin/cos/exp of the host libm: the last ulp of a 301-rad angle is 2e-5 absolute (another CPU's libm differs
    # by that much; on the container that made the fixture the tables agree to 1e-6)
    close(table[2:42].numpy(), g[f'd{d}'], rtol=0, atol=1e-5)
    close(table[19:22].numpy(), g[f'd{d}_off'], rtol=0, atol=1e-5)
    close(table[[0, 1, 2, 150, 301]].numpy(), g[f'd{d}_rows'], rtol=0, atol=5e-5)


@pytest.mark.parametrize('name', ['speech_whisper', 'speech_iwslt'])
def test_speech_frontend(name):
    g = load_golden(name)
    cfg = golden_cfg(g)
    seed, B, S, T, V = (int(g[k]) for k in ('seed', 'B', 'S', 'T', 'V'))
    P = {k: v.requires_grad_() for k, v in _state(g, seed).items()}
    if cfg.shared_embeddings:
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    feats = torch.from_numpy(paramgen.make_array(seed, name + '.feats', (B, S, cfg.input_dim)))
    lens = torch.from_numpy(g['lens'])
    for b in range(B):
        feats[b, lens[b]:] = 0
    feats.requires_grad_()
    tb = paramgen.make_text_batch(seed, B, 4, T, V)
    loss, logs = O.transformer_forward(P, cfg, feats, lens, torch.from_numpy(tb['decoder_input']))
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    assert logs['num_tokens'] == int(g['logs_num_tokens'])
    close(feats.grad.numpy(), g['dfeats'], rtol=1e-4)
    for n, ref_norm in zip(g['grad_names'], g['grad_norms']):
        n = str(n)
        assert abs(P[n].grad.double().norm().item() - ref_norm) <= 1e-4 * ref_norm + 2e-6, n
        if 'grad:' + n in g:
            close(P[n].grad.numpy(), g['grad:' + n], rtol=1e-4)
    with torch.no_grad():
        x = feats.detach()
        if 'encoder.in_linear.0.weight' in P:
            x = torch.clamp(O.linear(x, P['encoder.in_linear.0.weight'], P['encoder.in_linear.0.bias']), min=0)
        sub, new_len = O.conv_subsampler(P, 'encoder.subsample', x, lens, cfg.conv_kernel_sizes,
                                         cfg.conv_strides, cfg.conv_activation)
        enc_out, enc_mask = O.encoder(P, cfg, feats.detach(), lens)
    close(sub.numpy(), g['subsample_out'])
    assert (new_len.numpy() == g['new_len']).all()
    close(enc_out.numpy(), g['encoder_out'])
    assert (enc_mask.numpy() == g['encoder_mask']).all()


def test_greedy_decode_incremental_state():
    g = load_golden('greedy_decode')
    cfg = golden_cfg(g)
    seed, B, S, V = (int(g[k]) for k in ('seed', 'B', 'S', 'V'))
    P = _state(g, seed)
    if cfg.shared_embeddings:
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    batch = paramgen.make_text_batch(seed, B, S, 5, V)
    with torch.no_grad():
        enc_out, enc_mask = O.encoder(P, cfg, torch.from_numpy(batch['encoder_input']),
                                      torch.from_numpy(batch['encoder_input_length']))
        tokens = O.greedy_decode(P, cfg, enc_out, enc_mask, int(g['max_output_len']))
    assert tokens.shape == g['tokens'].shape
    assert (tokens.numpy() == g['tokens']).all()


def test_beam_search_trace_replay_through_the_oracle():
    """tests/golden/beam_trace.npz (what the reference's beam_search fed its decoder, and the reorderings of the
    incremental state — decoding.py:1225-1657, Decoder.reorder_state) replayed through the oracle's incremental decoder:
    the last-position logits of every step match, including after the batch shrinks"""
    g = load_golden('beam_trace')
    cfg = golden_cfg(g)
    seed, B, S, V, K = (int(g[k]) for k in ('seed', 'B', 'S', 'V', 'K'))
    P = _state(g, seed)
    if cfg.shared_embeddings:
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    batch = paramgen.make_text_batch(seed, B, S, 5, V)
    with torch.no_grad():
        enc_out, enc_mask = O.encoder(P, cfg, torch.from_numpy(batch['encoder_input']),
                                      torch.from_numpy(batch['encoder_input_length']))
        enc_out, enc_mask = enc_out.repeat_interleave(K, dim=0), enc_mask.repeat_interleave(K, dim=0)
        state = {}
        for i in range(int(g['n_calls'])):
            dec_in = torch.from_numpy(g[f'dec_in_{i}'])
            logits = O.decoder(P, cfg, enc_out, enc_mask, dec_in, state=state)[:, -1]
            close(logits.numpy(), g[f'logits_{i}'], rtol=2e-5)
            assert (logits.argmax(-1).numpy() == g[f'logits_{i}'].argmax(-1)).all()
            if i < int(g['n_reorders']):
                idx = torch.from_numpy(g[f'reorder_{i}'])
                enc_out, enc_mask = enc_out.index_select(0, idx), enc_mask.index_select(0, idx)
                for k, v in list(state.items()):  # Decoder.reorder_state (transformer.py): every tensor of the state
                    if torch.is_tensor(v):
                        state[k] = v.index_select(0, idx)


def test_adam_and_clip():
    g = load_golden('adam_step')
    n, steps = int(g['n']), int(g['steps'])
    ps = [torch.from_numpy(g[f'p0:{i}']) for i in range(n)]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for s in range(steps):
        grads = [torch.from_numpy(g[f'g{s}:{i}']) for i in range(n)]
        gnorm, grads = O.clip_grad_norm(grads, 1.0)
        assert abs(gnorm.item() - float(g[f'gnorm{s}'])) <= 1e-5 * float(g[f'gnorm{s}'])
        for i in range(n):
            ps[i], ms[i], vs[i] = O.adam_step(ps[i], grads[i], ms[i], vs[i], s + 1, 1e-3, 0.9, 0.98, 1e-8, 0.01)
            close(ps[i].numpy(), g[f'p{s + 1}:{i}'], rtol=1e-5)


def test_log_mel():
    g = load_golden('logmel')
    close(O.mel_filter_bank(), g['mel_filters'], rtol=1e-6, atol=1e-9)
    for i, key in enumerate(('wav0', 'wav1')):
        f = O.log_mel(g[key])
        assert f.shape == (3000, 80)
        # fp32-vs-fp64 STFT round-off on a log scale; features are O(1)
        assert np.abs(f[:240] - g['feats'][i]).max() < 2e-4
        assert np.abs(f[-4:] - g['feats_tail'][i]).max() < 2e-4


def test_first_steps_of_the_reference_training_curve():
    """tests/golden/train_curve.npz (120 steps of the REAL reference: model + gradient normalisation + clip + Adam +
    warm-up schedule on a reverse-the-source task): the oracle with its restatement of the optimizer reproduces the first
    12 steps — loss per token within 1e-4, gradient norm within 1e-3 (the CPU suite's share; the HIP path runs all 120
    steps on the GPU box, tests/test_training_curve_gpu.py)."""
    g = load_golden('train_curve')
    cfg = golden_cfg(g)
    P = {k: v.clone().requires_grad_() for k, v in O.to_torch_state(paramgen.make_state_dict(int(g['seed']), golden_names_shapes(g))).items()}
    P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    names = [n for n in P if n != 'decoder.embed_tokens.weight']
    m = {n: torch.zeros_like(P[n]) for n in names}
    v = {n: torch.zeros_like(P[n]) for n in names}
    lr, init_lr, min_lr, warmup, clip, b1, b2, eps, wd = [float(x) for x in g['hp']]
    for step in range(12):
        b = paramgen.make_reverse_batch(int(g['batch_seed0']) + step, int(g['B']), int(g['L']))
        for n in names:
            P[n].grad = None
        loss, logs = O.transformer_forward(P, cfg, **{k: torch.from_numpy(x) for k, x in b.items()})
        loss.backward()
        assert logs['num_tokens'] == int(g['num_tokens'][step])
        ref = float(g['loss_sum'][step]) / int(g['num_tokens'][step])
        assert abs(loss.item() / logs['num_tokens'] - ref) <= 1e-4 * ref, (step, loss.item() / logs['num_tokens'], ref)
        used = [n for n in names if P[n].grad is not None]
        total, grads = O.clip_grad_norm([P[n].grad / logs['num_tokens'] for n in used], clip)
        assert abs(float(total) - float(g['gnorm'][step])) <= 1e-3 * float(g['gnorm'][step]), step
        cur = init_lr + step * (lr - init_lr) / warmup if step < warmup else lr * (warmup / step) ** 0.5
        assert abs(max(cur, min_lr) - float(g['lr'][step])) <= 1e-12
        with torch.no_grad():
            for n, gr in zip(used, grads):
                p_new, m[n], v[n] = O.adam_step(P[n].detach(), gr, m[n], v[n], step + 1, max(cur, min_lr), b1, b2, eps, wd)
                P[n].copy_(p_new)


def test_every_fixture_cfg_has_the_current_schema():
    """a fixture written before a field was added to the schema would hand the model a configuration that silently lacks it
    (conftest.golden_cfg builds the namespace from the stored JSON): every `cfg` holds exactly paramgen.CFG_KEYS plus the
    EXTRA_KEYS that are set, and every whole-model fixture names its architecture and frozen set"""
    import glob
    import json
    import os
    from conftest import GOLDEN
    seen = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, '*.npz'))):
        g = np.load(path, allow_pickle=False)
        if 'cfg' not in g.files:
            continue
        seen += 1
        keys = set(json.loads(str(g['cfg'])))
        assert set(paramgen.CFG_KEYS) <= keys, (os.path.basename(path), sorted(set(paramgen.CFG_KEYS) - keys))
        assert keys <= set(paramgen.CFG_KEYS) | set(paramgen.EXTRA_KEYS), (os.path.basename(path), sorted(keys))
        if os.path.basename(path).startswith(('tiny_', 'base_')):  # gen_encdec's layout
            assert 'arch' in g.files and 'frozen_names' in g.files, os.path.basename(path)
    assert seen >= 20


def test_tiny_adapter_transformer_frozen_backbone():
    """adapter_transformer (pasero/models/adapters.py:232-301, modules.py:248-370): the oracle's bottleneck adapter behind
    every layer against the real reference — loss, every adapter gradient (the fixture stores gradients of trained parameters
    only), encoder output, logits, argmax"""
    g = load_golden('tiny_adapter')
    frozen = {str(n) for n in g['frozen_names']}
    assert frozen and all('adapters.' in str(n) for n in g['grad_names'])
    assert all('adapters.' not in n for n in frozen)
    _run_encdec('tiny_adapter')
