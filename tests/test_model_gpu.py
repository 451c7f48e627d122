"""End-to-end parity (GPU): pasero_amd.Transformer (HIP kernels through the C ABI) against
  (a) the golden vectors produced by the real reference on its PyTorch-CPU fp32 path, and
  (b) the CPU oracle on the same seeded inputs.
Bars (BASELINE.json north_star): bit-exact token argmax; loss within 1e-4 relative (fp32 kernels).  The bf16 kernels
are held to bf16 storage precision: loss within 2e-2 relative of the fp32 reference (stated per test)."""
import numpy as np
import pytest
import torch

import paramgen
from conftest import load_golden
from model_utils import build_model, build_cfg, oracle_state, text_batch, rel
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def _check_encdec(name, full=True):
    g = load_golden(name)
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.train()  # dropout probabilities are 0 in the fixture configs
    batch = text_batch(g, 'cuda')
    loss, logs = model(**batch)
    loss.backward()
    ref_loss = float(g['loss'])
    assert loss.dtype == torch.float32 and loss.dim() == 0
    assert abs(loss.item() - ref_loss) <= 1e-4 * abs(ref_loss)          # north_star: fp loss within 1e-4 rel
    assert abs(logs['loss'] - float(g['logs_loss'])) <= 1e-4 * abs(float(g['logs_loss']))
    assert abs(logs['nll_loss'] - float(g['logs_nll_loss'])) <= 1e-4 * abs(float(g['logs_nll_loss']))
    assert logs['num_tokens'] == int(g['logs_num_tokens']) and logs['num_lines'] == int(g['logs_num_lines'])
    for k in ('prompt_nll_loss', 'num_prompt_tokens'):  # the two-part loss of cfg.prompt_loss != 1
        assert (k in logs) == ('logs_' + k in g.files)
        if k in logs:
            assert abs(logs[k] - float(g['logs_' + k])) <= 1e-4 * abs(float(g['logs_' + k])), k
    grads = dict(model.named_parameters())
    for n, ref_norm in zip(g['grad_names'], g['grad_norms']):
        n = str(n)
        gr = grads[n].grad
        assert gr is not None, n
        # (a key bias shifts all scores of a row equally: its gradient is mathematically zero, what is left is the
        # round-off of one particular summation order)
        atol = 2e-5 if n.endswith('k_proj.bias') else 2e-6
        assert abs(gr.double().norm().item() - ref_norm) <= 2e-4 * ref_norm + atol, n
        if full:
            assert rel(gr, g['grad:' + n]) < 2e-4 or np.abs(g['grad:' + n]).max() < 1e-5, n
        else:
            # strided sample (as small as ONE element for a bias): tolerance relative to the tensor's typical
            # element magnitude, the norm above being the tight check.  1e-2: a ReLU pre-activation within fp32
            # round-off of 0 flips its 0/1 derivative, which moves single weight-gradient entries by ~3e-3 of max
            ref = g['gradsample:' + n]
            typical = ref_norm / np.sqrt(gr.numel())
            err = np.abs(gr.reshape(-1)[::4099].cpu().numpy() - ref).max()
            assert err <= 1e-2 * max(np.abs(ref).max(), typical) + (atol if n.endswith('k_proj.bias') else 1e-7), n
    model.eval()
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        logits, _ = model.decoder(enc_out, enc_mask, batch['decoder_input'][:, :-1])
        loss2, _ = model(**batch)
    assert abs(loss2.item() - ref_loss) <= 1e-4 * abs(ref_loss)
    if full:
        assert rel(enc_out, g['encoder_out']) < 2e-5
        assert (enc_mask.cpu().numpy() == g['encoder_mask']).all()
        assert rel(logits, g['logits']) < 2e-5
    else:
        assert rel(logits.reshape(-1)[::4099], g['logits_sample']) < 1e-4
    assert (logits.argmax(-1).cpu().numpy() == g['argmax']).all()       # north_star: bit-exact token argmax
    return g, cfg, model, batch


def test_tiny_postnorm_fp32_vs_reference():
    _check_encdec('tiny_encdec_post')


def test_tiny_prenorm_gelu_learned_fp32_vs_reference():
    _check_encdec('tiny_encdec_pre')


def test_tiny_rotary_gelu_tanh_fp32_vs_reference():
    """RoPE on the packed q|k projection + tanh-GELU FFN (north_star: "QKV-projection+RoPE", "bias+GELU FFN")"""
    _check_encdec('tiny_encdec_rotary')


@pytest.mark.parametrize('name', ['tiny_encdec_rotary', 'tiny_encdec_rms', 'tiny_hd128_rotary', 'tiny_lora_rotary'])
def test_rotary_fixtures_run_without_a_rope_pass(name, monkeypatch):
    """the rotary fixtures above pass with the rotation INSIDE the attention kernels: a training step (forward and backward,
    encoder and decoder self-attention, with and without LoRA branches) never calls pk_rope — and gives the same loss as
    the separate pass it replaces (PASERO_ROPE_PASS=1)"""
    from pasero_amd import functional as F, modules
    g = load_golden(name)
    calls = []
    real = F.rope
    monkeypatch.setattr(F, 'rope', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    losses = []
    for as_pass in (False, True):
        monkeypatch.setattr(modules, '_ROPE_PASS', as_pass)
        cfg, model = build_model(g, torch.float32, 'cuda')
        model.train()
        del calls[:]
        loss, _ = model(**text_batch(g, 'cuda'))
        loss.backward()
        losses.append(loss.item())
        assert (len(calls) > 0) == as_pass, (name, as_pass, len(calls))
    assert abs(losses[0] - losses[1]) <= 2e-6 * abs(losses[1]), losses


def test_tiny_swiglu_prenorm_fp32_vs_reference():
    """gated FFN (fc3): the gate product is the fc1 GEMM's epilogue"""
    _check_encdec('tiny_encdec_swiglu')


def test_tiny_rmsnorm_rotary_swiglu_no_bias_fp32_vs_reference():
    """llama-style parameterisation of the encoder-decoder: RMSNorm (the LayerNorm kernel with the mean fixed at 0),
    rotary positions, SwiGLU, no bias on any projection"""
    _check_encdec('tiny_encdec_rms')


def test_heads_of_128_with_rotary_positions_fp32_vs_reference():
    """RoPE on heads of 128 (the llama-style head size) together with RMSNorm / SwiGLU / no biases"""
    _check_encdec('tiny_hd128_rotary')


@pytest.mark.parametrize('name', ['tiny_opts_a', 'tiny_opts_b'])
def test_configuration_switches_fp32_vs_reference(name):
    """(a) untied output projection, separate encoder / decoder embeddings, LayerNorm without bias, unscaled attention
    scores, prompt tokens scored at weight 0.5 (the non-fused two-pass loss, transformer.py:283-321);
    (b) pre-norm layers sharing one norm, no biases, prompt tokens unscored, no label smoothing, embedding LayerNorm"""
    _check_encdec(name)


def test_partially_frozen_source_embeddings_fp32_vs_reference():
    """`--freeze-encoder-embed-regex`: Embedding(freeze_mask=...) (pasero/models/modules.py:900-947; VERDICT r3 missing 5) —
    tokens of the mask read the second table.  Loss, every gradient — of BOTH tables: masked rows of `weight` and unmasked
    rows of `frozen_embedding.weight` get exactly zero — encoder output, logits and argmax against the real reference."""
    g, cfg, model, batch = _check_encdec('tiny_freeze_embed')
    mask = torch.from_numpy(paramgen.make_freeze_mask(int(g['freeze_seed']), int(g['V']))).cuda()
    emb = model.encoder.embed_tokens
    assert emb.frozen_embedding is not None and mask.any() and (~mask).any()
    assert (emb.weight.grad[mask] == 0).all() and (emb.frozen_embedding.weight.grad[~mask] == 0).all()
    assert emb.frozen_embedding.weight.grad[mask].abs().max() > 0


def test_partially_frozen_embeddings_shared_with_the_decoder_fp32_vs_reference():
    """shared_embeddings + freeze_mask: the decoder carries the encoder's Embedding, so its lookup, its tied
    projection AND incremental decoding read the blended table — loss / gradients / logits against the real reference, and the
    reference's greedy tokens bit for bit (the native decoding plan holds one raw table pointer and must stand aside); the
    three readers of a training step share ONE merge of the two tables"""
    from pasero_amd import decode
    from pasero_amd.modules import MergeTablesFn
    g, cfg, model, batch = _check_encdec('tiny_freeze_shared')
    emb = model.decoder.embed_tokens
    assert emb is model.encoder.embed_tokens and emb.frozen_embedding is not None
    model.eval()
    assert decode.get_plan(model.decoder) is None
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        tokens = _greedy(model, enc_out, enc_mask, int(g['max_output_len']))
    assert tokens.shape == g['greedy_tokens'].shape and (tokens.cpu().numpy() == g['greedy_tokens']).all()
    # one merge node per parameter state
    model.train()
    model.zero_grad(set_to_none=True)
    loss, _ = model(**batch)
    merges, seen, todo = 0, set(), [loss.grad_fn]
    while todo:
        fn = todo.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        merges += type(fn).__name__.startswith(MergeTablesFn.__name__)
        todo += [f for f, _ in fn.next_functions]
    assert merges == 1, merges


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_tied_table_gradient_is_one_tensor(dtype, monkeypatch):
    """shared embeddings + tied projection (transformer.py:151-153): the table's gradient is the projection's dense dW with the
    rows of BOTH lookups added into it by the table's hook (pk_embed_bwd_acc; autograd.tie_table) — the same gradient as three
    autograd contributions (fp32: to summation-order round-off; bf16: one rounding instead of three), no session left behind
    after the pass, a second backward pass (retained graph) gives the same result again, and two graphs back-propagated
    together (`(l1 + l2).backward()`: two dense contributions, four lookups, ONE session) give the sum of their gradients"""
    from pasero_amd import autograd, functional as PF
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, dtype, 'cuda')
    assert cfg.shared_embeddings and cfg.tied_output_projection
    model.train()
    batch = text_batch(g, 'cuda')
    calls = []
    real = PF.embed_bwd
    monkeypatch.setattr(PF, 'embed_bwd', lambda *a, into=None, **k: (calls.append(into is not None), real(*a, into=into, **k))[1])

    def grads(retain=False):
        model.zero_grad(set_to_none=True)
        loss, _ = model(**batch)
        loss.backward(retain_graph=retain)
        first = model.encoder.embed_tokens.weight.grad.float().clone()
        if retain:
            model.zero_grad(set_to_none=True)
            loss.backward()
            assert torch.equal(model.encoder.embed_tokens.weight.grad.float(), first)
        return first
    sunk = grads(retain=True)
    assert calls == [True, True, True, True] and not autograd._table_sessions, calls
    del calls[:]
    monkeypatch.setattr(autograd, '_NO_GRAD_SINK', True)
    plain = grads()
    assert calls == [False, False]
    tol = 1e-6 if dtype == torch.float32 else 1.2e-2
    assert (sunk - plain).abs().max().item() <= tol * plain.abs().max().item()
    if dtype == torch.float32:  # and both are the reference's gradient
        assert rel(sunk, g['grad:encoder.embed_tokens.weight']) < 2e-4
    # two graphs in one backward pass
    monkeypatch.setattr(autograd, '_NO_GRAD_SINK', False)
    b2 = {k: (v.flip(0).contiguous() if v.dim() else v) for k, v in batch.items()}
    model.zero_grad(set_to_none=True)
    model(**b2)[0].backward()
    other = model.encoder.embed_tokens.weight.grad.float().clone()
    model.zero_grad(set_to_none=True)
    del calls[:]
    (model(**batch)[0] + model(**b2)[0]).backward()
    both = model.encoder.embed_tokens.weight.grad.float()
    assert calls == [True] * 4 and not autograd._table_sessions
    assert (both - (sunk + other)).abs().max().item() <= 2 * tol * (sunk + other).abs().max().item()


def test_adapter_transformer_frozen_backbone_fp32_vs_reference():
    """adapter_transformer (pasero/models/adapters.py): bottleneck adapters after every layer, only they are trained;
    loss, every adapter gradient, logits and argmax against the real reference; frozen parameters get no gradient"""
    g, cfg, model, batch = _check_encdec('tiny_adapter')
    frozen = {str(n) for n in g['frozen_names']}
    assert frozen and all(not p.requires_grad and p.grad is None for n, p in model.named_parameters() if n in frozen)
    assert all(p.requires_grad == (n not in frozen) for n, p in model.named_parameters())
    assert any('adapters.default.down.weight' in str(n) for n in g['grad_names'])


def test_lora_branches_fp32_vs_reference():
    """LoRA on every Linear (q/k/v/out projections, fc1, fc2), rank 4, alpha 8: modules.py:67-100"""
    g, cfg, model, batch = _check_encdec('tiny_lora')
    assert any(str(n).endswith('q_proj.lora.up.weight') for n in g['grad_names'])
    assert any(str(n).endswith('fc2.lora.down.weight') for n in g['grad_names'])


def test_lora_with_rotary_positions_fp32_vs_reference():
    """the separate-projection attention path (LoRA) with RoPE applied to its q and k outputs"""
    _check_encdec('tiny_lora_rotary')


def test_heads_of_128_fp32_vs_reference():
    """embed_dim 256 with 2 heads: the head_dim-128 instantiations of the attention kernels inside the whole model"""
    _check_encdec('tiny_hd128')


def test_mha_rotary_incremental_offsets():
    """rotary self-attention decoded one token at a time (offset = cached length) equals the full causal pass"""
    from pasero_amd.modules import MultiheadAttention
    g = load_golden('mha_rotary')
    d, H, B, T = (int(g[k]) for k in 'dHBT')
    mha = MultiheadAttention(d, H, causal=True, positional_encoding='rotary')
    names = [str(n) for n in g['param_names']]
    shapes = [tuple(int(x) for x in str(s).split(',')) for s in g['param_shapes']]
    assert [(k, tuple(v.shape)) for k, v in mha.state_dict().items()] == list(zip(names, shapes))
    mha.load_state_dict({k: torch.from_numpy(v) for k, v in paramgen.make_state_dict(33, list(zip(names, shapes))).items()})
    mha = mha.cuda()
    x = torch.from_numpy(paramgen.make_array(33, 'rot.x', (B, T, d))).cuda().requires_grad_()
    y, _ = mha(query=x, key=x, value=x)
    y.backward(torch.from_numpy(paramgen.make_array(33, 'rot.dy', (B, T, d))).cuda())
    assert rel(y, g['y']) < 2e-5
    assert rel(x.grad, g['dx']) < 2e-4
    for n, p in mha.named_parameters():
        assert rel(p.grad, g['grad:' + n]) < 2e-4 or np.abs(g['grad:' + n]).max() < 1e-5, n
    with torch.no_grad():
        state, steps = {}, []
        for i in range(T):
            xi = x[:, i:i + 1].contiguous()
            yi, _ = mha(query=xi, key=xi, value=xi, state=state)
            steps.append(yi)
    assert rel(torch.cat(steps, 1), g['y_incremental']) < 2e-5


def test_base_c1_fp32_vs_reference():
    """BASELINE configs[0]: Transformer-base 6+6 d=512 V=8032, batch 8x(64,64)"""
    _check_encdec('base_c1', full=False)


@pytest.mark.parametrize('name', ['tiny_encdec_post', 'tiny_encdec_pre'])
def test_fp32_vs_oracle_same_inputs(name):
    """the oracle on the same seeded inputs, a different batch than the golden one (ragged, other seed)"""
    g = load_golden(name)
    cfg, model = build_model(g, torch.float32, 'cuda')
    P = {k: v.requires_grad_() for k, v in oracle_state(g, cfg).items()}
    if cfg.shared_embeddings:
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    b = paramgen.make_text_batch(777, 5, 11, 9, int(g['V']))
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    ref_loss, ref_logs = O.transformer_forward(P, cfg, **tb)
    ref_loss.backward()
    model.train()
    loss, logs = model(**{k: v.cuda() for k, v in tb.items()})
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item())
    assert logs['num_tokens'] == ref_logs['num_tokens']
    for n, p in model.named_parameters():
        assert rel(p.grad, P[n].grad) < 2e-4 or P[n].grad.abs().max() < 1e-5, n


@pytest.mark.parametrize('name', ['tiny_encdec_post', 'tiny_encdec_pre', 'base_c1'])
def test_bf16_model_vs_reference(name):
    """bf16 kernels (MFMA attention / GEMMs): weights, activations and P/dS tiles are rounded to bf16 (2^-8 relative
    per rounding, ~30 roundings deep) -> loss within 2e-2 relative of the fp32 reference, grads within 10 % in norm"""
    g = load_golden(name)
    cfg, model = build_model(g, torch.bfloat16, 'cuda')
    model.train()
    batch = text_batch(g, 'cuda')
    loss, logs = model(**batch)
    loss.backward()
    ref_loss = float(g['loss'])
    assert loss.dtype == torch.float32
    assert abs(loss.item() - ref_loss) <= 2e-2 * abs(ref_loss)
    assert logs['num_tokens'] == int(g['logs_num_tokens'])
    grads = dict(model.named_parameters())
    tot_ref = float(np.sqrt((g['grad_norms'] ** 2).sum()))
    tot = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in grads.values())).item())
    assert abs(tot - tot_ref) <= 0.1 * tot_ref
    assert all(torch.isfinite(p.grad.float()).all() for p in grads.values())


@pytest.mark.parametrize('name', ['speech_whisper', 'speech_iwslt'])
def test_speech_frontend_fp32_vs_reference(name):
    g = load_golden(name)
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.train()
    seed, B, S, T, V = (int(g[k]) for k in ('seed', 'B', 'S', 'T', 'V'))
    feats = torch.from_numpy(paramgen.make_array(seed, name + '.feats', (B, S, cfg.input_dim)))
    lens = torch.from_numpy(g['lens'])
    for b in range(B):
        feats[b, lens[b]:] = 0
    feats = feats.cuda().requires_grad_()
    tb = paramgen.make_text_batch(seed, B, 4, T, V)
    loss, logs = model(encoder_input=feats, encoder_input_length=lens.cuda(),
                       decoder_input=torch.from_numpy(tb['decoder_input']).cuda(),
                       prompt_mask=torch.from_numpy(tb['prompt_mask']).cuda())
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    assert logs['num_tokens'] == int(g['logs_num_tokens'])
    assert rel(feats.grad, g['dfeats']) < 2e-4
    grads = dict(model.named_parameters())
    for n, ref_norm in zip(g['grad_names'], g['grad_norms']):
        n = str(n)
        assert abs(grads[n].grad.double().norm().item() - ref_norm) <= 2e-4 * ref_norm + 2e-6, n
        if 'grad:' + n in g:
            assert rel(grads[n].grad, g['grad:' + n]) < 2e-4, n
    model.eval()
    with torch.no_grad():
        x = feats.detach()
        if model.encoder.in_linear is not None:
            from pasero_amd.autograd import LinearFn
            lin = model.encoder.in_linear[0]
            x = LinearFn.apply(x, lin.weight, lin.bias, 'relu')
        sub, new_len = model.encoder.subsample(x, lens.cuda())
        enc_out, enc_mask, _ = model.encoder(feats.detach(), lens.cuda())
    assert rel(sub, g['subsample_out']) < 2e-5
    assert (new_len.cpu().numpy() == g['new_len']).all()
    assert rel(enc_out, g['encoder_out']) < 2e-5
    assert (enc_mask.cpu().numpy() == g['encoder_mask']).all()


def _greedy(model, enc_out, enc_mask, max_output_len):
    """greedy search with the decoder's incremental `state` (decoding.py:1119-1221 restricted to one BOS column)"""
    cfg = model.cfg
    B = enc_out.size(0)
    max_len = min(cfg.decoder_max_len, 1 + max_output_len)
    tokens = torch.full((B, max_len), cfg.padding_idx, dtype=torch.long, device=enc_out.device)
    tokens[:, 0] = cfg.bos_idx
    has_eos = torch.zeros(B, dtype=torch.bool, device=enc_out.device)
    state, prev, last = {}, 0, 0
    for step in range(1, max_len):
        has_eos = has_eos | (step >= 1 + max_output_len)
        logits, _ = model.decoder(enc_out, enc_mask, tokens[:, prev:step], state=state)
        logits = logits[:, -1].float().clone()
        pad_logit = logits[:, cfg.padding_idx].clone()
        logits[has_eos] = -float('inf')
        logits[:, cfg.padding_idx] = pad_logit
        tokens[:, step] = logits.argmax(-1)
        last = step
        has_eos = (has_eos | (tokens[:, step] == cfg.eos_idx)) & (step >= 1)
        prev = step
        if bool(has_eos.all()):
            break
    return tokens[:, 1:last + 1]


def test_greedy_decode_incremental_state_bit_exact():
    g = load_golden('greedy_decode')
    g2 = dict(g)
    g2['T'] = 5
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.eval()
    b = paramgen.make_text_batch(int(g['seed']), int(g['B']), int(g['S']), 5, int(g['V']))
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(torch.from_numpy(b['encoder_input']).cuda(),
                                             torch.from_numpy(b['encoder_input_length']).cuda())
        tokens = _greedy(model, enc_out, enc_mask, int(g['max_output_len']))
    assert tokens.shape == g['tokens'].shape
    assert (tokens.cpu().numpy() == g['tokens']).all()


def test_dropout_training_step_runs_and_is_reproducible():
    """dropout 0.1 (the real training configuration): same seed -> same loss and grads; another seed differs"""
    from pasero_amd import rng
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.bfloat16, 'cuda')
    cfg.dropout = 0.1
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1
    model.train()
    batch = text_batch(g, 'cuda')
    out = []
    for seed in (5, 5, 6):
        rng.manual_seed(seed)
        model.zero_grad(set_to_none=True)
        loss, _ = model(**batch)
        loss.backward()
        out.append((loss.item(), model.decoder.layers[0].fc1.weight.grad.float().norm().item()))
    assert out[0] == out[1]
    assert out[0] != out[2]
    assert abs(out[0][0] - float(g['loss'])) < 0.5 * float(g['loss'])


def test_subclass_hooks_are_honoured():
    """adapters / MoE override `ffn` and the residual hooks (adapters.py:232-301, mixture_of_experts.py:434): an
    overriding subclass must be called, and must switch the fused residual+LayerNorm path off"""
    from pasero_amd.transformer import TransformerEncoderLayer
    calls = []

    class Layer(TransformerEncoderLayer):
        def ffn(self, x, residual, padding_mask):
            calls.append('ffn')
            return super().ffn(x, residual, padding_mask)

        def ffn_residual(self, x, residual):
            calls.append('ffn_residual')
            return super().ffn_residual(x, residual)

    g = load_golden('tiny_encdec_post')
    cfg = build_cfg(g)
    from pasero_amd.config import DistributedConfig
    torch.manual_seed(0)
    a = TransformerEncoderLayer(cfg, DistributedConfig(), 0).cuda()
    b = Layer(cfg, DistributedConfig(), 0).cuda()
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 5, cfg.embed_dim, device='cuda')
    mask = torch.zeros(2, 5, dtype=torch.bool, device='cuda')
    ya, _ = a(x, mask)
    yb, _ = b(x, mask)
    assert calls == ['ffn', 'ffn_residual']
    assert rel(yb, ya) < 1e-5


def _decode_logits(model, enc_out, enc_mask, tokens, reorder_at=None, indices=None):
    """teacher-forced incremental decoding: logits of every step (first call: 2 tokens, then one token per call)"""
    state, out = {}, []
    with torch.no_grad():
        lg, _ = model.decoder(enc_out, enc_mask, tokens[:, :2], state=state)
        out.append(lg[:, -1].float())
        for t in range(2, tokens.size(1)):
            if reorder_at == t:
                model.decoder.reorder_state(state, indices)
                enc_out, enc_mask, tokens = enc_out[indices], enc_mask[indices], tokens[indices]
            lg, _ = model.decoder(enc_out, enc_mask, tokens[:, t:t + 1], state=state)
            out.append(lg[:, -1].float())
    return out, state


@pytest.mark.parametrize('name,dtype', [('tiny_encdec_post', torch.float32), ('tiny_encdec_pre', torch.float32),
                                        ('base_c1', torch.bfloat16)])
def test_native_decoding_step_matches_per_op_path(name, dtype, monkeypatch):
    """pk_decoder_step (preallocated KV caches, cross K/V computed once) against the per-op incremental path that is
    pinned bit-exactly to the reference's greedy tokens: same kernels underneath -> same logits up to the different
    summation order of two GEMM shapes (fp32 2e-5, bf16 2e-2 of the logit range); the state stays reference-shaped"""
    g = load_golden(name)
    cfg, model = build_model(g, dtype, 'cuda')
    model.eval()
    B, S, V = int(g['B']), int(g['S']), int(g['V'])
    b = paramgen.make_text_batch(int(g['seed']), B, S, 12, V)
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(torch.from_numpy(b['encoder_input']).cuda(),
                                             torch.from_numpy(b['encoder_input_length']).cuda())
    tokens = torch.from_numpy(b['decoder_input']).cuda().clamp(min=2)  # no pad tokens inside the prefix
    native, st = _decode_logits(model, enc_out, enc_mask, tokens)
    assert '_pk_decode' in st, 'the native step did not engage'
    H = cfg.decoder_attention_heads
    assert st['dec_0_self_attn_key'].shape == (B, tokens.size(1), H, 64) and st['offset'] == tokens.size(1)
    monkeypatch.setenv('PASERO_NO_NATIVE_DECODE', '1')
    per_op, st2 = _decode_logits(model, enc_out, enc_mask, tokens)
    assert '_pk_decode' not in st2
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for a, r in zip(native, per_op):
        assert (a - r).abs().max().item() <= tol * r.abs().max().item()
        if dtype == torch.float32:
            assert torch.equal(a.argmax(-1), r.argmax(-1))
    last = f'dec_{cfg.decoder_layers - 1}_self_attn_value'
    assert rel(st[last].float(), st2[last].float()) <= tol
    # beam-search style reordering of the state in the middle of a sentence
    monkeypatch.delenv('PASERO_NO_NATIVE_DECODE')
    idx = torch.tensor([B - 1] + list(range(B - 1)) + [0], device='cuda')
    a, _ = _decode_logits(model, enc_out, enc_mask, tokens, reorder_at=6, indices=idx)
    monkeypatch.setenv('PASERO_NO_NATIVE_DECODE', '1')
    r, _ = _decode_logits(model, enc_out, enc_mask, tokens, reorder_at=6, indices=idx)
    for x, y in zip(a, r):
        assert x.shape == y.shape and (x - y).abs().max().item() <= tol * y.abs().max().item()


def test_native_decoding_plan_follows_replaced_parameters(monkeypatch):
    """the native step holds raw pointers to the weights: in-place updates are seen as they are; a parameter whose
    storage is REPLACED between two sentences (p.data = ...) must invalidate the plan — and a model left with mixed
    parameter dtypes must fall back to the per-op path, which refuses it loudly"""
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.eval()
    B, S, V = int(g['B']), int(g['S']), int(g['V'])
    b = paramgen.make_text_batch(int(g['seed']), B, S, 8, V)
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(torch.from_numpy(b['encoder_input']).cuda(),
                                             torch.from_numpy(b['encoder_input_length']).cuda())
    tokens = torch.from_numpy(b['decoder_input']).cuda().clamp(min=2)
    before, st = _decode_logits(model, enc_out, enc_mask, tokens)
    assert '_pk_decode' in st
    w = model.decoder.layers[1].fc2.weight
    with torch.no_grad():
        w.mul_(1.5)                                   # in place: same storage
    inplace, _ = _decode_logits(model, enc_out, enc_mask, tokens)
    w.data = (w.data / 1.5).clone()                   # new storage, original values
    replaced, st = _decode_logits(model, enc_out, enc_mask, tokens)
    assert '_pk_decode' in st
    monkeypatch.setenv('PASERO_NO_NATIVE_DECODE', '1')
    per_op, _ = _decode_logits(model, enc_out, enc_mask, tokens)
    monkeypatch.delenv('PASERO_NO_NATIVE_DECODE')
    assert (inplace[-1] - before[-1]).abs().max().item() > 1e-3       # the in-place update was seen
    for a, r0, r1 in zip(replaced, before, per_op):
        assert (a - r0).abs().max().item() <= 2e-5 * r0.abs().max().item()
        assert (a - r1).abs().max().item() <= 2e-5 * r1.abs().max().item()
    model.decoder.layers[0].self_attn_layer_norm.double()              # one module in another dtype
    with pytest.raises((TypeError, AssertionError)):
        _decode_logits(model, enc_out, enc_mask, tokens)


@pytest.mark.parametrize('native', [True, False])
def test_beam_search_trace_replay(native, monkeypatch):
    """tests/golden/beam_trace.npz: what the reference's beam_search (decoding.py:1225-1657) fed ITS decoder at every
    step of a real search — tokens, incremental state, and the beam re-ordering of the state, including the step where
    a finished sentence leaves the batch (12 -> 9 rows) — with the logits it got back.  Replayed here through
    pasero_amd's decoder and `Decoder.reorder_state`, with the native decoding step and with the per-op path:
    same last-position logits at every step (fp32, 2e-5 of the range) and the same best next token"""
    from pasero_amd.transformer import Decoder
    g = load_golden('beam_trace')
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.eval()
    if not native:
        monkeypatch.setenv('PASERO_NO_NATIVE_DECODE', '1')
    B, S, V, K, seed = (int(g[k]) for k in ('B', 'S', 'V', 'K', 'seed'))
    b = paramgen.make_text_batch(seed, B, S, 5, V)
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(torch.from_numpy(b['encoder_input']).cuda(),
                                             torch.from_numpy(b['encoder_input_length']).cuda())
        enc_out, enc_mask = enc_out.repeat_interleave(K, dim=0), enc_mask.repeat_interleave(K, dim=0)
        state, engaged, rows = {}, False, []
        for i in range(int(g['n_calls'])):
            dec_in = torch.from_numpy(g[f'dec_in_{i}']).cuda()
            assert dec_in.size(0) == enc_out.size(0)
            rows.append(dec_in.size(0))
            logits, _ = model.decoder(enc_out, enc_mask, dec_in, state=state)
            engaged |= '_pk_decode' in state
            want = torch.from_numpy(g[f'logits_{i}']).cuda()
            got = logits[:, -1].float()
            assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item(), i
            assert torch.equal(got.argmax(-1), want.argmax(-1)), i
            if i < int(g['n_reorders']):
                idx = torch.from_numpy(g[f'reorder_{i}']).cuda()
                enc_out, enc_mask = enc_out.index_select(0, idx), enc_mask.index_select(0, idx)
                Decoder.reorder_state(state, idx)
    assert len(set(rows)) > 1          # the trace does contain a shrinking batch
    assert engaged == native


@pytest.mark.parametrize('fixture', ['return_layers', 'return_layers_rotary'])
def test_return_layers_hidden_states_and_attention_weights(fixture):
    """`return_layers` (transformer.py:698-752,831-898): hidden states after chosen layers and (B,T,H,S) attention
    weights of chosen attention blocks, for a full pass and for one incremental step (where the native decoding step
    must stand aside, the per-op path produces the weights) — against the real reference"""
    g = load_golden(fixture)
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.eval()
    batch = text_batch(g, 'cuda')
    enc_names, dec_names = [str(n) for n in g['enc_names']], [str(n) for n in g['dec_names']]
    with torch.no_grad():
        enc_out, enc_mask, enc_layers = model.encoder(batch['encoder_input'], batch['encoder_input_length'],
                                                      return_layers=enc_names)
        dec_in = batch['decoder_input'][:, :-1]
        logits, dec_layers = model.decoder(enc_out, enc_mask, dec_in, return_layers=dec_names)
        assert sorted(enc_layers) == sorted(enc_names) and sorted(dec_layers) == sorted(dec_names)
        for k, v in {**enc_layers, **dec_layers}.items():
            want = g['full:' + k]
            assert tuple(v.shape) == want.shape, k
            assert rel(v, want) < 2e-5, k
        assert rel(logits, g['logits']) < 2e-5
        state = {}
        model.decoder(enc_out, enc_mask, dec_in[:, :3], state=state)
        _, step_layers = model.decoder(enc_out, enc_mask, dec_in[:, 3:4], state=state, return_layers=dec_names)
        assert sorted(step_layers) == sorted(dec_names)
        for k, v in step_layers.items():
            want = g['step:' + k]
            assert tuple(v.shape) == want.shape, k
            assert rel(v, want) < 2e-5, k
        # without return_layers nothing is collected
        _, none = model.decoder(enc_out, enc_mask, dec_in)
        assert none == {}


def test_argmax_rows_first_maximum():
    import ctypes
    from pasero_amd import lib
    x = torch.randn(37, 8032, device='cuda').bfloat16()
    x[3, 100] = x[3, 7000] = 50.0   # tie: lowest index wins
    x[5] = -float('inf')
    out = torch.full((37, 2), -1, dtype=torch.long, device='cuda')
    lib.check(lib.load().pk_argmax_rows(x.data_ptr(), 37, 8032, 8032, out.data_ptr(), 2, lib.PK_BF16,
                                        torch.cuda.current_stream().cuda_stream), 'pk_argmax_rows')
    want = x.float().argmax(-1)
    want[5] = 0
    assert torch.equal(out[:, 0], want) and out[3, 0] == 100 and (out[:, 1] == -1).all()


def test_activation_checkpointing_replays_the_dropout_masks():
    """cfg.checkpoint_activations (modules.py:386-391): layers are recomputed in the backward pass; with dropout on, the
    recomputation has to draw the masks of the first run — loss and every gradient equal the non-checkpointed step"""
    from pasero_amd import rng
    from pasero_amd.config import DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    from model_utils import load_paramgen
    g = load_golden('tiny_encdec_post')
    results = []
    for ckpt in (False, True):
        cfg = build_cfg(g)
        cfg.dropout, cfg.attention_dropout, cfg.checkpoint_activations = 0.2, 0.1, ckpt
        model = Transformer(cfg, DistributedConfig(), SyntheticTask(int(g['V'])))
        load_paramgen(model, int(g['seed']))
        model = model.cuda().train()
        rng.manual_seed(77)
        loss, _ = model(**text_batch(g, 'cuda'))
        loss.backward()
        results.append((loss.item(), {n: p.grad.clone() for n, p in model.named_parameters()}, rng.get_state()))
    (l0, g0, s0), (l1, g1, s1) = results
    assert l0 == l1 and s0 == s1  # same masks in the forward pass, and the offset stream ends at the same place
    for n in g0:
        if 'embed_tokens' in n:   # fp32 atomics: summation order varies from run to run
            assert rel(g1[n], g0[n]) < 1e-5, n
        else:
            assert torch.equal(g0[n], g1[n]), n


def test_inference_runs_no_backward_only_work():
    """`ctx.needs_input_grad` stays True for parameters under torch.no_grad(): the forward passes must look at the
    caller's grad mode instead (pasero_amd.autograd.wants_grad), or scoring at inference would run the gradient GEMMs of
    the fused vocabulary loss and write every z = x + residual.  Counted with the library's own GEMM sampler."""
    from pasero_amd import lib
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.float32, 'cuda')
    batch = text_batch(g, 'cuda')
    L = lib.load()

    def gemm_calls(fn):
        lib.check(L.pk_gemm_timing_start(4096, 1), 'pk_gemm_timing_start')
        out = fn()
        torch.cuda.synchronize()
        return L.pk_gemm_timing_stop(), out

    model.eval()
    with torch.no_grad():
        n_eval, (loss_eval, _) = gemm_calls(lambda: model(**batch))
    n_train_fwd, (loss_train, _) = gemm_calls(lambda: model(**batch))
    # per encoder layer qkv, out, fc1, fc2; per decoder layer qkv, out, q, kv, out, fc1, fc2; + the vocabulary projection
    forward_gemms = 4 * cfg.encoder_layers + 7 * cfg.decoder_layers + 1
    assert n_eval == forward_gemms
    assert n_train_fwd == forward_gemms + 2  # dX and dW of the fused loss, made while the logits chunk is cache-resident
    assert loss_eval.item() == loss_train.item()


def test_edge_cases_empty_single_token_and_all_pad_targets():
    """degenerate batches the reference handles: one-token sentences, a batch whose targets are all padding (loss 0,
    num_tokens 0, finite zero gradients), an empty batch at the kernel boundary"""
    from pasero_amd import functional as F
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.train()
    V = int(g['V'])
    # one source token (EOS) and one target token per sentence: S = 1, T = 1
    batch = {'encoder_input': torch.full((3, 1), 2, device='cuda'), 'encoder_input_length': torch.ones(3, dtype=torch.long, device='cuda'),
             'decoder_input': torch.tensor([[2, 7], [2, 9], [2, 2]], device='cuda'),
             'prompt_mask': torch.tensor([[True, False]] * 3, device='cuda')}
    loss, logs = model(**batch)
    loss.backward()
    assert logs['num_tokens'] == 3 and torch.isfinite(loss) and loss.item() > 0
    # the same through the oracle
    P = oracle_state(g, cfg)
    ref, ref_logs = O.transformer_forward(P, cfg, **{k: v.cpu() for k, v in batch.items()})
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())
    # all targets are padding: nothing to predict
    model.zero_grad(set_to_none=True)
    batch['decoder_input'] = torch.tensor([[2, 1], [2, 1], [2, 1]], device='cuda')
    loss, logs = model(**batch)
    loss.backward()
    assert loss.item() == 0.0 and logs['num_tokens'] == 0
    for n, p in model.named_parameters():
        assert p.grad is None or (torch.isfinite(p.grad).all() and p.grad.abs().max().item() == 0.0), n
    # empty batches at the C ABI: every entry point accepts zero rows
    e = torch.empty(0, 128, device='cuda')
    w = torch.randn(64, 128, device='cuda')
    assert F.gemm(e, w).shape == (0, 64)
    y, z, mean, rstd = F.residual_ln_fwd(e, None, torch.ones(128, device='cuda'), torch.zeros(128, device='cuda'), 1e-5)
    assert y.shape == (0, 128) and mean.numel() == 0
    o, lse = F.attn_fwd(torch.empty(0, 4, 128, device='cuda'), torch.empty(0, 5, 128, device='cuda'),
                        torch.empty(0, 5, 128, device='cuda'), 2, None, False, 0.125)
    assert o.shape == (0, 4, 128)
    rl, rn = torch.empty(0, device='cuda'), torch.empty(0, device='cuda')
    F.ce_rows(torch.empty(0, V, device='cuda'), torch.empty(0, dtype=torch.long, device='cuda'), 1, 0.1, rl, rn)
    assert F.ce_finalize(rl, rn, torch.empty(0, dtype=torch.long, device='cuda'), 1).tolist() == [0.0, 0.0, 0.0]


def test_sequence_longer_than_the_positional_table_is_rejected_like_the_reference():
    """modules.py:441-446 / :467-472: positions beyond the table raise instead of reading out of bounds"""
    g = load_golden('tiny_encdec_pre')  # learned positions
    cfg, model = build_model(g, torch.float32, 'cuda')
    n = cfg.decoder_max_len + 5
    batch = {'encoder_input': torch.full((1, 4), 5, device='cuda'), 'encoder_input_length': torch.tensor([4], device='cuda'),
             'decoder_input': torch.full((1, n + 1), 5, device='cuda'), 'prompt_mask': torch.zeros(1, n + 1, dtype=torch.bool, device='cuda')}
    with pytest.raises(AssertionError, match='too long'):
        model(**batch)


def test_activation_dropout_keeps_the_residual_branch_gradient(monkeypatch):
    """activation_dropout > 0 in a post-norm model (transformer.py:1005-1008): the FFN runs module by module, and the
    gradient of the residual branch, which the fused block end parks on a ResidualLink, must still reach the layer
    input.  Same seeds -> same masks, so the step with the links switched off (plain autograd accumulation) is the
    exact reference; fp32, every gradient."""
    from pasero_amd import rng
    from pasero_amd.transformer import _LayerBase
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.float32, 'cuda')
    for layer in list(model.encoder.layers) + list(model.decoder.layers):
        layer.activation_dropout.p = 0.25
    model.train()
    batch = text_batch(g, 'cuda')

    def step():
        rng.manual_seed(11)
        model.zero_grad(set_to_none=True)
        loss, _ = model(**batch)
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    loss_a, grads_a = step()
    monkeypatch.setattr(_LayerBase, '_linked', lambda self, *a, **k: None)
    loss_b, grads_b = step()
    assert loss_a == loss_b and grads_a.keys() == grads_b.keys()
    for k in grads_a:
        assert rel(grads_a[k], grads_b[k]) < 2e-5, k
    # and the dropped activations do change the step
    for layer in list(model.encoder.layers) + list(model.decoder.layers):
        layer.activation_dropout.p = 0.0
    assert step()[0] != loss_a


def test_attention_dropout_training_step():
    """attention_dropout 0.1 (the IWSLT2023 recipes): the step runs through the dropout instantiations of the attention
    kernels, is reproducible from the seed, differs from the dropout-free step and from another seed, and in eval mode
    nothing is dropped"""
    from pasero_amd import rng
    from pasero_amd.modules import MultiheadAttention
    g = load_golden('tiny_encdec_post')
    cfg, model = build_model(g, torch.bfloat16, 'cuda')
    batch = text_batch(g, 'cuda')
    model.train()
    base, _ = model(**batch)
    for m in model.modules():
        if isinstance(m, MultiheadAttention):
            m.dropout = 0.1
    out = []
    for seed in (5, 5, 6):
        rng.manual_seed(seed)
        model.zero_grad(set_to_none=True)
        loss, _ = model(**batch)
        loss.backward()
        gq = model.decoder.layers[0].encoder_attn.q_proj.weight.grad.float()
        assert torch.isfinite(loss) and torch.isfinite(gq).all()
        out.append((loss.item(), gq.norm().item()))
    assert out[0] == out[1] and out[0] != out[2] and out[0][0] != base.item()
    assert abs(out[0][0] - base.item()) < 0.2 * base.item()
    model.eval()
    with torch.no_grad():
        a, _ = model(**batch)
        b, _ = model(**batch)
    assert a.item() == b.item() and abs(a.item() - base.item()) <= 1e-6 * base.item()


@pytest.mark.parametrize('name', ['tiny_encdec_post', 'tiny_encdec_pre', 'base_c1'])
def test_fp16_model_vs_reference(name):
    """float16 — the reference's default `--dtype` (config.py:518-523): the same kernels with the f16 MFMA and
    conversions; 2^-11 relative per rounding -> loss within 3e-3 of the fp32 reference, total gradient norm within 2 %,
    every gradient finite (no loss scaling needed at these magnitudes), greedy argmax of the tiny models unchanged"""
    g = load_golden(name)
    cfg, model = build_model(g, torch.float16, 'cuda')
    model.train()
    batch = text_batch(g, 'cuda')
    loss, logs = model(**batch)
    loss.backward()
    ref_loss = float(g['loss'])
    assert loss.dtype == torch.float32
    assert abs(loss.item() - ref_loss) <= 3e-3 * abs(ref_loss)
    assert logs['num_tokens'] == int(g['logs_num_tokens'])
    grads = dict(model.named_parameters())
    assert all(p.grad.dtype == torch.float16 and torch.isfinite(p.grad.float()).all() for p in grads.values())
    tot_ref = float(np.sqrt((g['grad_norms'] ** 2).sum()))
    tot = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in grads.values())).item())
    assert abs(tot - tot_ref) <= 2e-2 * tot_ref
    # and the native decoding step in fp16
    model.eval()
    with torch.no_grad():
        enc_out, enc_mask, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        tokens = batch['decoder_input'].clamp(min=2)
        state = {}
        lg, _ = model.decoder(enc_out, enc_mask, tokens[:, :2], state=state)
        lg2, _ = model.decoder(enc_out, enc_mask, tokens[:, 2:3], state=state)
        full, _ = model.decoder(enc_out, enc_mask, tokens[:, :3])
    assert '_pk_decode' in state
    assert (lg2[:, -1].float() - full[:, -1].float()).abs().max().item() <= 2e-2 * full[:, -1].float().abs().max().item()


@pytest.mark.parametrize('name', ['tiny_encdec_post', 'tiny_encdec_rms', 'speech_whisper'])
def test_amp_autocast_fp32_parameters_fp16_compute(name):
    """`--amp` (pasero/training.py:27-31,107-112,379): torch.autocast(float16) around the forward pass of an fp32 model.
    The custom functions follow the custom_fwd contract — fp32 tensors are cast on the way in, the kernels run in fp16,
    parameter gradients come back in fp32"""
    g = load_golden(name)
    cfg, model = build_model(g, torch.float32, 'cuda')
    model.train()
    if name.startswith('speech'):
        seed, B, S, T, V = (int(g[k]) for k in ('seed', 'B', 'S', 'T', 'V'))
        feats = torch.from_numpy(paramgen.make_array(seed, name + '.feats', (B, S, cfg.input_dim)))
        lens = torch.from_numpy(g['lens'])
        for b in range(B):
            feats[b, lens[b]:] = 0
        tb = paramgen.make_text_batch(seed, B, 4, T, V)
        batch = dict(encoder_input=feats.cuda(), encoder_input_length=lens.cuda(),
                     decoder_input=torch.from_numpy(tb['decoder_input']).cuda(),
                     prompt_mask=torch.from_numpy(tb['prompt_mask']).cuda())
    else:
        batch = text_batch(g, 'cuda')
    with torch.autocast('cuda', dtype=torch.float16):
        enc_out, _, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        loss, logs = model(**batch)
    assert enc_out.dtype == torch.float16       # the kernels really ran in 16 bits
    loss.backward()
    ref_loss = float(g['loss'])
    assert loss.dtype == torch.float32
    assert abs(loss.item() - ref_loss) <= 3e-3 * abs(ref_loss)
    assert logs['num_tokens'] == int(g['logs_num_tokens'])
    grads = dict(model.named_parameters())
    assert all(p.dtype == torch.float32 and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all()
               for p in grads.values())
    tot_ref = float(np.sqrt((g['grad_norms'] ** 2).sum()))
    tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in grads.values())).item())
    assert abs(tot - tot_ref) <= 2e-2 * tot_ref


def test_lora_weights_merge_into_the_linear_layers_at_inference():
    """transformer.py:484-497: at inference the low-rank updates are folded into the weights when the checkpoint is loaded
    (cfg.lora_rank is reset to 0 by setup_for_inference) — the merged model must compute what the LoRA model computes"""
    import dataclasses
    g = load_golden('tiny_lora')
    cfg, lora_model = build_model(g, torch.float32, 'cuda')
    batch = text_batch(g, 'cuda')
    lora_model.eval()
    with torch.no_grad():
        enc_out, enc_mask, _ = lora_model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        want, _ = lora_model.decoder(enc_out, enc_mask, batch['decoder_input'][:, :-1])
    sd = {k: v.detach().cpu().clone() for k, v in lora_model.state_dict().items()}
    from pasero_amd.config import DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    plain = Transformer(dataclasses.replace(cfg, lora_rank=0), DistributedConfig(), SyntheticTask(int(g['V'])))
    plain.eval()
    plain.update_state_dict(sd)
    assert not any('.lora.' in k for k in sd)
    plain.load_state_dict(sd)
    plain = plain.cuda()
    with torch.no_grad():
        enc_out2, enc_mask2, _ = plain.encoder(batch['encoder_input'], batch['encoder_input_length'])
        got, _ = plain.decoder(enc_out2, enc_mask2, batch['decoder_input'][:, :-1])
    assert rel(got, want) < 1e-4
    assert torch.equal(got.argmax(-1), want.argmax(-1))
