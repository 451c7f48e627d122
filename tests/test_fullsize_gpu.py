"""Parity at BASELINE.json's FULL sizes (configs[1] = C2: Transformer-base, V = 8032, B = 256, S = T = 128) through
size-independent properties — the CPU oracle needs minutes at this size, so instead of comparing against it the tests
check what must hold for ANY correct implementation of the path:
  * the loss is a sum over target tokens: a batch equals the sum of its halves, rows may be permuted, trailing padding
    changes nothing (reference semantics: pasero/models/transformer.py:355-377 sums the label-smoothed NLL over
    non-pad targets; attention masks keys by length, modules.py:662-683);
  * GEMM: scaling an operand by a power of two scales the output bit-exactly; row checksums match an fp32 reference;
  * attention: every row of softmax weights sums to one (V = 1 gives O = 1); lse matches an fp32 logsumexp;
  * LayerNorm (gamma = 1, beta = 0): rows have zero mean and unit variance;
  * cross-entropy: the gradient of a row sums to zero, pad rows have zero gradient.
Tolerances are written at each check; bf16 results are held to bf16 storage precision (2^-8 relative per element)."""
import numpy as np
import pytest
import torch

from model_utils import load_paramgen

pytestmark = pytest.mark.gpu

V, B, S, T = 8032, 256, 128, 128


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def _model(dtype, dropout=0.0):
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    cfg = TransformerConfig(dropout=dropout)
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V))
    load_paramgen(model, 11)
    return model.to(dtype).cuda().train()


def _batch(b=B, s=S, t=T, ragged=True, seed=5):
    import paramgen
    return {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(seed, b, s, t, V, ragged=ragged).items()}


def _rows(batch, idx):
    return {k: v[idx].contiguous() for k, v in batch.items()}


WATCH = ['encoder.layers.0.fc1.weight', 'encoder.layers.5.self_attn.q_proj.weight', 'decoder.layers.3.fc2.weight',
         'decoder.layers.5.encoder_attn.k_proj.weight', 'decoder.layers.0.self_attn_layer_norm.weight',
         'encoder.embed_tokens.weight']


def _step(model, batch):
    model.zero_grad(set_to_none=True)
    loss, logs = model(**batch)
    loss.backward()
    params = dict(model.named_parameters())
    return loss.item(), logs, {n: params[n].grad.float().clone() for n in WATCH}


def test_c2_loss_and_grads_are_additive_over_the_batch():
    model = _model(torch.bfloat16)
    batch = _batch()
    full, logs, g = _step(model, batch)
    h1, logs1, g1 = _step(model, _rows(batch, slice(0, B // 2)))
    h2, logs2, g2 = _step(model, _rows(batch, slice(B // 2, B)))
    assert logs['num_tokens'] == logs1['num_tokens'] + logs2['num_tokens'] == int((batch['decoder_input'][:, 1:] != 1).sum())
    # every row is computed independently of its batch neighbours; only the final fp64 -> fp32 sum differs
    assert abs(full - (h1 + h2)) <= 1e-5 * abs(full)
    for n in WATCH:  # bf16 gradients: each half is rounded once more before the sum
        err = (g[n] - (g1[n] + g2[n])).abs().max().item()
        assert err <= 2e-2 * g[n].abs().max().item(), (n, err)


def test_c2_row_permutation_and_trailing_padding_change_nothing():
    model = _model(torch.bfloat16)
    batch = _batch(s=96, t=96, ragged=True)
    base, logs, g = _step(model, batch)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    p_loss, p_logs, pg = _step(model, _rows(batch, perm))
    assert abs(p_loss - base) <= 1e-5 * abs(base) and p_logs['num_tokens'] == logs['num_tokens']
    # pad source to S = 128 and targets to T = 128: masked keys carry weight exp(-inf) = 0, pad targets no loss
    padded = {
        'encoder_input': torch.nn.functional.pad(batch['encoder_input'], (0, S - 96), value=1),
        'encoder_input_length': batch['encoder_input_length'],
        'decoder_input': torch.nn.functional.pad(batch['decoder_input'], (0, T - 96), value=1),
        'prompt_mask': torch.nn.functional.pad(batch['prompt_mask'], (0, T - 96), value=False),
    }
    q_loss, q_logs, qg = _step(model, padded)
    assert q_logs['num_tokens'] == logs['num_tokens']
    assert abs(q_loss - base) <= 1e-4 * abs(base)  # other key-tile grouping of the online softmax (bf16 P)
    for n in WATCH:
        scale = g[n].abs().max().item()
        assert (pg[n] - g[n]).abs().max().item() <= 2e-2 * scale, n
        assert (qg[n] - g[n]).abs().max().item() <= 2e-2 * scale, n


def test_c2_dropout_step_is_reproducible_and_bf16_tracks_fp32():
    from pasero_amd import rng
    model = _model(torch.bfloat16, dropout=0.1)
    batch = _batch(ragged=False)  # the bench.py workload
    rng.manual_seed(9)
    a, _, ga = _step(model, batch)
    rng.manual_seed(9)
    b, _, gb = _step(model, batch)
    assert a == b
    for n in WATCH:  # (no float atomics anywhere on the path: embedding and LayerNorm gradients included)
        assert torch.equal(ga[n], gb[n]), n
    # no dropout: the bf16 step against the fp32 kernels on the same weights and batch
    m16, m32 = _model(torch.bfloat16), _model(torch.float32)
    l16, _, g16 = _step(m16, batch)
    l32, _, g32 = _step(m32, batch)
    assert abs(l16 - l32) <= 2e-2 * abs(l32)  # bf16 storage precision through 12 layers
    for n in WATCH:
        # relative L2 error of a bf16 gradient.  Measured: 1.3-3.2 %, of which ~2.2 % is the rounding of the weights
        # themselves (fp32 kernels on bf16-rounded weights differ from fp32 by that much).  The cross-attention key
        # projection is the exception (11.6 %, identical with the fused and the two-kernel backward): at random init
        # the attention is near uniform and dK = dS^T Q cancels almost completely, so the bf16 rounding of dS — which
        # every bf16-MFMA flash backward shares — is large relative to what is left.
        tol = 0.2 if 'encoder_attn.k_proj' in n else 0.05
        num = (g16[n] - g32[n]).norm().item()
        assert num <= tol * g32[n].norm().item(), (n, num)


@pytest.mark.parametrize('a_col,b_col', [(False, False), (False, True), (True, True)])
def test_fullsize_gemm_pow2_scaling_is_exact_and_checksums_match(a_col, b_col):
    from pasero_amd import functional as F
    # the C2 shapes of the three forms: y = x W^T (32768 x 2048 x 512), dX = dY W, dW = dY^T X (K = 32768)
    M, N, K = (2048, 512, 32768) if a_col else (32768, 2048, 512)
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    Bm = torch.randn(N, K, device='cuda', generator=g).bfloat16()
    a = A.t().contiguous() if a_col else A
    b = Bm.t().contiguous() if b_col else Bm
    sk = F.choose_splitk(M, N, K) if a_col else 1
    c1 = F.gemm(a, b, a_col=a_col, b_col=b_col, splitk=sk)
    c2 = F.gemm(a * 2, b * 0.5, a_col=a_col, b_col=b_col, splitk=sk)
    assert torch.equal(c1, c2)  # powers of two commute with every rounding on the way
    # checksum of checksums: C 1 = A (B^T 1), in fp32
    want = A.float() @ Bm.float().sum(0)
    got = c1.float().sum(1)
    tol = 2.0 ** -8 * (A.float().abs() @ Bm.float().abs().sum(0)) / N ** 0.5 * 4 + 1e-3  # N independent bf16 roundings
    assert ((got - want).abs() <= tol).all()


@pytest.mark.parametrize('causal', [False, True])
def test_fullsize_attention_rows_sum_to_one_and_lse_matches(causal):
    from pasero_amd import functional as F
    H, D = 8, 512
    g = torch.Generator(device='cuda').manual_seed(2)
    q = torch.randn(B, T, D, device='cuda', generator=g).bfloat16()
    k = torch.randn(B, S, D, device='cuda', generator=g).bfloat16()
    v = torch.ones(B, S, D, device='cuda', dtype=torch.bfloat16)
    lens = torch.randint(S // 2, S + 1, (B,), device='cuda', generator=g)
    pad = None if causal else torch.arange(S, device='cuda')[None, :] >= lens[:, None]
    o, lse = F.attn_fwd(q, k, v, H, pad, causal, 0.125)
    # weights are rounded to bf16 before they multiply V: |sum_j bf16(p_j) - 1| <= 2^-9 sum_j p_j
    assert (o.float() - 1).abs().max().item() <= 2.0 ** -7
    for bi, h in [(0, 0), (B - 1, H - 1), (101, 3)]:
        s = (q[bi, :, 64 * h: 64 * h + 64].float() @ k[bi, :, 64 * h: 64 * h + 64].float().t()) * 0.125
        if causal:
            s = s.masked_fill(torch.ones(T, S, device='cuda', dtype=torch.bool).triu(1), float('-inf'))
        else:
            s = s.masked_fill(pad[bi][None, :], float('-inf'))
        assert (lse[bi, h] - torch.logsumexp(s, 1)).abs().max().item() <= 1e-3  # fp32 accumulation of bf16 products


def test_fullsize_layernorm_rows_are_standardised():
    from pasero_amd import functional as F
    rows, d = B * T, 512
    g = torch.Generator(device='cuda').manual_seed(4)
    x = (3 * torch.randn(rows, d, device='cuda', generator=g) + 1.5).bfloat16()
    res = torch.randn(rows, d, device='cuda', generator=g).bfloat16()
    gamma, beta = torch.ones(d, device='cuda', dtype=torch.bfloat16), torch.zeros(d, device='cuda', dtype=torch.bfloat16)
    y = F.residual_ln_fwd(x, res, gamma, beta, 1e-5)[0].float()
    assert y.shape == (rows, d)
    assert y.mean(1).abs().max().item() <= 2e-3        # mean of 512 values each rounded to bf16 (|y| <~ 4)
    assert (y.var(1, unbiased=False) - 1).abs().max().item() <= 2e-2


def test_fullsize_cross_entropy_gradient_rows_sum_to_zero():
    from pasero_amd import functional as F
    rows = B * T
    g = torch.Generator(device='cuda').manual_seed(6)
    logits = (2 * torch.randn(rows, V, device='cuda', generator=g)).bfloat16()
    target = torch.randint(4, V, (rows,), device='cuda', generator=g)
    target[::7] = 1  # pad rows
    row_loss = torch.empty(rows, device='cuda')
    row_nll = torch.empty(rows, device='cuda')
    dlogits = torch.empty_like(logits)
    F.ce_rows(logits, target, 1, 0.1, row_loss, row_nll, dlogits)
    sums = dlogits.float().sum(1)
    assert sums[target == 1].abs().max().item() == 0.0 and row_loss[target == 1].abs().max().item() == 0.0
    # softmax - smoothed one-hot sums to zero; V bf16 roundings of ~1/V plus one of ~0.9
    assert sums.abs().max().item() <= 1e-2
    lse = torch.logsumexp(logits[:4096].float(), 1)
    nll = lse - logits[:4096].float().gather(1, target[:4096, None])[:, 0]
    keep = target[:4096] != 1
    assert (row_nll[:4096] - nll)[keep].abs().max().item() <= 1e-3 * nll[keep].abs().max().item()
    sums3 = F.ce_finalize(row_loss, row_nll, target, 1)
    assert int(sums3[2].item()) == int((target != 1).sum())
    assert abs(sums3[1].item() - row_nll.double().sum().item()) <= 1e-5 * row_nll.double().sum().item()


@pytest.mark.parametrize('vocab,layers,b', [(70376, 2, 64), (256206, 1, 32)])
def test_c3_c5_vocabularies_chunked_loss(vocab, layers, b, monkeypatch):
    """BASELINE configs C3 (transformer_big, V = 70376) and C5 (NLLB, V = 256206) at their widths (d = 1024, 16 heads,
    S = T = 128) with fewer layers: the fused vocabulary loss walks the rows in logits chunks — the result must not depend on the chunking, and stays additive over the batch"""
    from pasero_amd import autograd
    from pasero_amd.config import TransformerBigConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    import paramgen
    cfg = TransformerBigConfig(encoder_layers=layers, decoder_layers=layers, dropout=0.0)
    torch.manual_seed(3)
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(vocab)).to(torch.bfloat16).cuda().train()
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(9, b, 128, 128, vocab).items()}
    watch = ['encoder.embed_tokens.weight', 'decoder.layers.0.fc1.weight']

    def step(rows=slice(None)):
        model.zero_grad(set_to_none=True)
        loss, logs = model(**{k: v[rows].contiguous() for k, v in batch.items()})
        loss.backward()
        params = dict(model.named_parameters())
        return loss.item(), logs['num_tokens'], {n: params[n].grad.float().clone() for n in watch}

    chunks = []
    real = autograd._ce_chunk_rows
    # (a quarter of the default budget of 1200 MiB: these rows in several chunks, cut by the same rule)
    monkeypatch.setattr(autograd, '_ce_chunk_rows',
                        lambda rows, V, itemsize: chunks.append(real(rows, V, itemsize, budget_bytes=300 << 20)) or chunks[-1])
    full, ntok, g = step()
    assert chunks[-1] < b * 128, 'a 300 MiB budget should split these rows into several chunks'
    h1, n1, g1 = step(slice(0, b // 2))
    h2, n2, g2 = step(slice(b // 2, b))
    assert ntok == n1 + n2 and abs(full - (h1 + h2)) <= 1e-5 * abs(full)
    for n in watch:
        assert (g[n] - (g1[n] + g2[n])).abs().max().item() <= 2e-2 * g[n].abs().max().item(), n
    monkeypatch.setattr(autograd, '_ce_chunk_rows', lambda rows, V, itemsize, budget_bytes=0: min(rows, 640))
    small, ntok2, gs = step()   # many small chunks (640 rows: not a multiple of the 256-row tiles)
    assert ntok2 == ntok and abs(small - full) <= 1e-6 * abs(full)   # per-row losses are the same numbers
    for n in watch:  # the gradient chunks are rounded to bf16 before they are summed: more chunks, more roundings
        assert (gs[n] - g[n]).abs().max().item() <= 2e-2 * g[n].abs().max().item(), n


def test_c2_shape_layer_pair_against_the_cpu_oracle():
    """The fast 16-bit kernels only engage at base width with thousands of rows (the B-stationary and the phase-interleaved
    GEMMs, the GEMM with the LayerNorm epilogue, the grouped weight gradients, the fused attention backward), which none of
    the reference fixtures reaches: ONE encoder + ONE decoder layer at the C2 shape (B = 256, S = T = 128, d = 512,
    V = 8032, bf16) against oracle/ref_cpu.py in fp32 on the same (bf16-representable) weights and batch — loss within
    2e-2 relative, every gradient within 5 % relative L2 (cross-attention k_proj 20 %: see the test above)."""
    import paramgen
    from oracle import ref_cpu as O
    from pasero_amd import lib
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    import ctypes
    cfg = TransformerConfig(dropout=0.0, encoder_layers=1, decoder_layers=1)
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V))
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = {k: torch.from_numpy(v).bfloat16().float() for k, v in paramgen.make_state_dict(21, names_shapes).items()}
    state['decoder.embed_tokens.weight'] = state['encoder.embed_tokens.weight']
    model.load_state_dict(state)
    model = model.to(torch.bfloat16).cuda().train()
    b = paramgen.make_text_batch(6, B, S, T, V, ragged=True)
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    # which GEMM kernels ran (the library's launch sampling, every launch): the fast ones must be among them
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(256, 1), 'start')
    model.zero_grad(set_to_none=True)
    loss, logs = model(**batch)
    loss.backward()
    n = L.pk_gemm_timing_stop()
    tags = set()
    for i in range(n):
        ints = [ctypes.c_int() for _ in range(5)]
        fl, ms = ctypes.c_double(), ctypes.c_float()
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
        tags.add(ints[0].value)
    assert any(t & 0x200 for t in tags), tags              # gemmbs
    assert (8 | 0x40) in tags and (8 | 0x80) in tags, tags  # grouped weight gradients, GEMM + LayerNorm
    assert any((t & 0x2CF) == 8 for t in tags), tags        # gemm8p
    got = {k: p.grad.float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    P = {k: v.clone().requires_grad_() for k, v in state.items() if k != 'decoder.embed_tokens.weight'}
    P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    ref_loss, ref_logs = O.transformer_forward(P, cfg, **{k: torch.from_numpy(v) for k, v in b.items()})
    ref_loss.backward()
    assert logs['num_tokens'] == ref_logs['num_tokens']
    assert abs(loss.item() - ref_loss.item()) <= 2e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    checked = 0
    for k, g in got.items():
        if k == 'decoder.embed_tokens.weight':
            continue
        r = P[k].grad
        tol = 0.2 if 'encoder_attn.k_proj' in k else 0.05
        err = (g - r).norm().item()
        if k.endswith('k_proj.bias'):
            # softmax does not see a constant added to every key: this gradient is zero in exact arithmetic (the
            # reference's fp32 value is 1e-6 of round-off), what the bf16 path leaves is judged against the gradient of
            # the projection's weight
            assert err <= 0.05 * P[k.replace('bias', 'weight')].grad.norm().item(), (k, err)
        else:
            assert err <= tol * r.norm().item() + 1e-6, (k, err, r.norm().item())
        checked += 1
    assert checked >= 40, checked


# ------------------------------------------------------------------------------------------------------------
# the natively run PRE-NORM 16-bit layer (what C4 and C5 run in bench.py) and the d = 1024 kernels against the oracle
# ------------------------------------------------------------------------------------------------------------
def _gemm_tags(run):
    """run() under the library's launch sampling (every GEMM launch): the set of kernel tags that ran"""
    import ctypes
    from pasero_amd import lib
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(1024, 1), 'start')
    out = run()
    n = L.pk_gemm_timing_stop()
    tags = set()
    for i in range(n):
        ints = [ctypes.c_int() for _ in range(5)]
        fl, ms = ctypes.c_double(), ctypes.c_float()
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
        tags.add(ints[0].value)
    return out, tags


def _count_native_calls(fn):
    from pasero_amd import native_layer
    calls = {'n': 0}
    orig = native_layer.NativeLayerFn.forward

    def counted(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    native_layer.NativeLayerFn.forward = staticmethod(counted)
    try:
        out = fn()
    finally:
        native_layer.NativeLayerFn.forward = staticmethod(orig)
    return out, calls['n']


def _oracle_grads(state, cfg, batch, act_dtype=None):
    from oracle import ref_cpu as O
    P = {k: v.clone().requires_grad_() for k, v in state.items() if k != 'decoder.embed_tokens.weight'}
    if 'decoder.embed_tokens.weight' in state:
        P['decoder.embed_tokens.weight'] = P['encoder.embed_tokens.weight']
    O.ACT_DTYPE = act_dtype
    try:
        loss, logs = O.transformer_forward(P, cfg, **batch)
        loss.backward()
    finally:
        O.ACT_DTYPE = None
    grads = {k: v.grad for k, v in P.items() if v.grad is not None and k != 'decoder.embed_tokens.weight'}
    if act_dtype is not None:  # the reference's 16-bit parameters receive 16-bit gradients
        grads = {k: g.to(act_dtype).float() for k, g in grads.items()}
    return loss.item(), logs, grads


def _check_grads(got, ref, what, tol=0.05, kproj_tol=0.2):
    checked = 0
    for k, g in got.items():
        if k == 'decoder.embed_tokens.weight' or k not in ref:
            continue
        r = ref[k]
        err = (g - r).norm().item()
        if k.endswith('k_proj.bias'):  # zero in exact arithmetic (softmax ignores a constant added to every key)
            assert err <= 0.05 * ref[k.replace('bias', 'weight')].norm().item(), (what, k, err)
        else:
            t = kproj_tol if 'encoder_attn.k_proj' in k else tol
            assert err <= t * r.norm().item() + 1e-6, (what, k, err, r.norm().item())
        checked += 1
    return checked


def _bf16_bracket(what, loss, got, state, cfg, cpu_batch, true_loss, true, bound=1.5):
    """the bracket of test_c2_pair_bf16_error_is_the_error_of_16_bit_storage for any layer pair — per
    gradient, the HIP bf16 path's distance from the fp32 truth within `bound` x the distance of the oracle with bf16 module
    boundaries (oracle/ref_cpu.py ACT_DTYPE), + 0.3 % of the gradient's norm for tensors both get almost exactly"""
    o16_loss, _, o16 = _oracle_grads(state, cfg, cpu_batch, torch.bfloat16)
    e_hip, e_o16 = abs(loss - true_loss) / abs(true_loss), abs(o16_loss - true_loss) / abs(true_loss)
    assert e_hip <= bound * e_o16 + 1e-3, (what, e_hip, e_o16)
    worst, rows = (0.0, None), []
    for k, r in true.items():
        if k.endswith('k_proj.bias') or k not in got:
            continue
        n = r.norm().item()
        if n == 0:
            continue
        h, o = (got[k] - r).norm().item() / n, (o16[k] - r).norm().item() / n
        rows.append((h / max(o, 1e-12), k, h, o))
        if o > 0 and h / o > worst[0]:
            worst = (h / o, k)
    # (the cross-attention key projection's gradient is a near-total cancellation — softmax ignores a constant added to every
    # key — over S keys: what is left of it amplifies any difference in summation order; its plain bound is 20 % where the
    # others have 5 %, and its bracket is 2 x theirs.  Measured at the C4 pair, S = 1500: 0.84 % against the bf16 oracle's 0.34 %)
    bad = [(k, round(h, 5), round(o, 5)) for ratio, k, h, o in rows
           if h > (2 * bound if 'encoder_attn.k_proj' in k else bound) * o + 3e-3]
    print(f'{what} bf16 error bracket: loss hip {e_hip:.2e} / oracle-bf16 {e_o16:.2e}; worst gradient ratio '
          f'{worst[0]:.2f} ({worst[1]}); {len(rows)} gradients')
    assert not bad, (what, bad)


def test_c4_shape_prenorm_pair_bf16_native_layers_against_the_cpu_oracle():
    """the natively run pre-norm 16-bit layer had no oracle comparison (its rows >= 256 / d >= 256 rule
    keeps every d = 128 reference fixture off it).  The whole C4 path at its own rows with ONE encoder + ONE decoder layer:
    whisper_base shapes — (16, 3000, 80) features -> conv k3 s1 + conv k3 s2 (GELU) -> 1500 positions, learned positions,
    pre-norm GELU layers of d = 512, f = 2048, key projections without bias, T = 64, V = 51 865 — bf16, native layer calls
    on, against oracle/ref_cpu.py in fp32 on the same bf16-representable weights and features: loss 2e-2, every gradient
    5 % relative L2 (cross-attention k_proj 20 %), and both layers must have run as native calls."""
    import paramgen
    from pasero_amd.config import WhisperConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    V_, B_, T_ = 51865, 16, 64
    cfg = WhisperConfig(dropout=0.0, encoder_layers=1, decoder_layers=1)
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V_))
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = {k: torch.from_numpy(v).bfloat16().float() for k, v in paramgen.make_state_dict(23, names_shapes).items()}
    if cfg.shared_embeddings:
        state['decoder.embed_tokens.weight'] = state['encoder.embed_tokens.weight']
    model.load_state_dict(state)
    model = model.to(torch.bfloat16).cuda().train()
    feats = torch.randn(B_, 3000, 80, generator=torch.Generator().manual_seed(4)).bfloat16().float()
    tb = paramgen.make_text_batch(11, B_, 4, T_, V_)
    cpu_batch = {'encoder_input': feats, 'encoder_input_length': torch.full((B_,), 3000, dtype=torch.int64),
                 'decoder_input': torch.from_numpy(tb['decoder_input']), 'prompt_mask': torch.from_numpy(tb['prompt_mask'])}
    batch = {k: (v.bfloat16() if v.is_floating_point() else v).cuda() for k, v in cpu_batch.items()}

    def step():
        model.zero_grad(set_to_none=True)
        loss, logs = model(**batch)
        loss.backward()
        return loss, logs
    ((loss, logs), tags), native = _count_native_calls(lambda: _gemm_tags(step))
    assert native == 2, native                                   # the encoder and the decoder layer, one call each
    assert (8 | 0x40) in tags and any(t & 0x200 for t in tags), tags   # grouped weight gradients, the B-stationary GEMM (GELU inside)
    got = {k: p.grad.float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    ref_loss, ref_logs, ref = _oracle_grads(state, cfg, cpu_batch)
    assert logs['num_tokens'] == ref_logs['num_tokens']
    assert abs(loss.item() - ref_loss) <= 2e-2 * abs(ref_loss), (loss.item(), ref_loss)
    assert _check_grads(got, ref, 'c4') >= 40
    _bf16_bracket('c4', loss.item(), got, state, cfg, cpu_batch, ref_loss, ref)


def test_c5_width_prenorm_pair_bf16_native_layers_against_the_cpu_oracle():
    """the d = 1024 bf16 kernels at the shapes where they engage (gemm8p at K = 1024 / 8192, the grouped
    weight gradients of d = 1024, the two-strip LayerNorm kernels) met the oracle only through kernel-level fuzz.  ONE
    encoder + ONE decoder layer of the NLLB-1.3B preset (d = 1024, f = 8192, 16 heads, pre-norm, ReLU) at 4096 rows
    (B = 32, S = T = 128), bf16, native layer calls on, against the fp32 oracle: loss 2e-2, every gradient 5 %."""
    import paramgen
    from pasero_amd import config as C
    from pasero_amd.transformer import Transformer
    V_, B_, S_, T_ = 1000, 32, 128, 128
    cfg = C.NLLB1B3Config(encoder_layers=1, decoder_layers=1, dropout=0.0)
    model = Transformer(cfg, C.DistributedConfig(), C.SyntheticTask(V_))
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = {k: torch.from_numpy(v).bfloat16().float() for k, v in paramgen.make_state_dict(43, names_shapes).items()}
    if cfg.shared_embeddings:
        state['decoder.embed_tokens.weight'] = state['encoder.embed_tokens.weight']
    model.load_state_dict(state)
    model = model.to(torch.bfloat16).cuda().train()
    b = paramgen.make_text_batch(44, B_, S_, T_, V_, ragged=True)
    cpu_batch = {k: torch.from_numpy(v) for k, v in b.items()}
    batch = {k: v.cuda() for k, v in cpu_batch.items()}

    def step():
        model.zero_grad(set_to_none=True)
        loss, logs = model(**batch)
        loss.backward()
        return loss, logs
    ((loss, logs), tags), native = _count_native_calls(lambda: _gemm_tags(step))
    assert native == 2, native
    assert (8 | 0x40) in tags and any((t & 0x2CF) == 8 for t in tags), tags   # grouped weight gradients, gemm8p
    got = {k: p.grad.float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    ref_loss, ref_logs, ref = _oracle_grads(state, cfg, cpu_batch)
    assert logs['num_tokens'] == ref_logs['num_tokens']
    assert abs(loss.item() - ref_loss) <= 2e-2 * abs(ref_loss), (loss.item(), ref_loss)
    assert _check_grads(got, ref, 'c5') >= 40
    _bf16_bracket('c5', loss.item(), got, state, cfg, cpu_batch, ref_loss, ref)


def test_iwslt_recipe_layer_pair_frozen_backbone_bf16_against_the_cpu_oracle(monkeypatch):
    """the frozen-backbone per-op route of the IWSLT2023 recipe at ITS rows and width, against the
    oracle: (32, 1000, 1024) features -> in_linear 1024 -> 80 + ReLU -> conv k5 s2 + GLU -> 500 positions = 16 000 rows of
    d = 1024; ONE frozen pre-norm encoder layer with a trained bottleneck adapter (d -> 64 -> d) behind it, ONE frozen decoder
    layer (T = 64: 2048 rows, f = 8192), bf16; only in_linear, the subsampler and the adapter train — so every gradient that
    is checked has crossed the frozen decoder and encoder layers backwards.  oracle/ref_cpu.py in fp32 on the same
    bf16-representable weights: loss 2e-2, gradients 5 % relative L2; and the routes the recipe's timed step depends on must
    have run: the pre-norm fork node, the forward K-slabs of fc2 at 2048 rows, the few-rows kernel for the adapter's
    16 000 x 64 down-projection."""
    import paramgen
    import bench
    from pasero_amd import autograd, config as C, functional as PF
    from pasero_amd import adapters  # noqa: F401
    V_, B_, S_, T_ = 1000, 32, 1000, 64
    over = {**bench.IWSLT_OVERRIDES, 'encoder_layers': 1, 'decoder_layers': 1, 'encoder_adapter_layer_ids': [0],
            'dropout': 0.0, 'attention_dropout': 0.0}
    cfg = C.AdapterNLLB1B3Config(**over)
    model = C.get_architecture(cfg)(cfg, C.DistributedConfig(), C.SyntheticTask(V_))
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = {k: torch.from_numpy(v).bfloat16().float() for k, v in paramgen.make_state_dict(47, names_shapes).items()}
    if cfg.shared_embeddings:
        state['decoder.embed_tokens.weight'] = state['encoder.embed_tokens.weight']
    model.load_state_dict(state)
    model = model.to(torch.bfloat16).cuda().train()
    trained_re = r'(.*\.in_linear|.*\.subsample|.*\.adapters)'
    import re
    for n, p in model.named_parameters():
        p.requires_grad = bool(re.match(trained_re, n))
    trained = {n for n, p in model.named_parameters() if p.requires_grad}
    assert any('adapters.default.down.weight' in n for n in trained) and not any('.fc1.' in n for n in trained)
    feats = torch.randn(B_, S_, 1024, generator=torch.Generator().manual_seed(8)).bfloat16().float()
    tb = paramgen.make_text_batch(13, B_, 4, T_, V_)
    cpu_batch = {'encoder_input': feats, 'encoder_input_length': torch.full((B_,), S_, dtype=torch.int64),
                 'decoder_input': torch.from_numpy(tb['decoder_input']), 'prompt_mask': torch.from_numpy(tb['prompt_mask'])}
    batch = {k: (v.bfloat16() if v.is_floating_point() else v).cuda() for k, v in cpu_batch.items()}
    forks, splits = {'n': 0}, []
    orig_fork, orig_split = autograd.LayerNormForkFn.forward, PF.fwd_split

    def fork(*a, **k):
        forks['n'] += 1
        return orig_fork(*a, **k)
    monkeypatch.setattr(autograd.LayerNormForkFn, 'forward', staticmethod(fork))
    orig_end = autograd.ResidualDropoutLnFn.forward  # (round 6: inside a layer, block end + the next block's LayerNorm as one node)

    def end(*a, **k):
        forks['n'] += 1
        return orig_end(*a, **k)
    monkeypatch.setattr(autograd.ResidualDropoutLnFn, 'forward', staticmethod(end))
    monkeypatch.setattr(PF, 'fwd_split', lambda M, N, K, dt: splits.append((M, N, K, orig_split(M, N, K, dt))) or splits[-1][3])

    def step():
        model.zero_grad(set_to_none=True)
        loss, logs = model(**batch)
        loss.backward()
        return loss, logs
    (loss, logs), tags = _gemm_tags(step)
    # (the encoder layer's two pre-norm sub-blocks + the decoder sub-blocks whose input carries a gradient: with a frozen
    # embedding that is the feed-forward block, behind the cross-attention)
    assert 3 <= forks['n'] <= 5, forks
    assert (2048, 1024, 8192, 4) in splits, splits          # the decoder's fc2 as K-slabs
    assert 64 in tags, tags                                 # the few-rows kernel (the adapter's down-projection)
    got = {k: p.grad.float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == trained, set(got) ^ trained
    ref_loss, ref_logs, ref = _oracle_grads(state, cfg, cpu_batch)
    ref = {k: v for k, v in ref.items() if k in trained}
    assert logs['num_tokens'] == ref_logs['num_tokens']
    assert abs(loss.item() - ref_loss) <= 2e-2 * abs(ref_loss), (loss.item(), ref_loss)
    assert _check_grads(got, ref, 'iwslt') == len(trained)


def test_c2_pair_bf16_error_is_the_error_of_16_bit_storage():
    """"loss 2e-2, gradients 5 %" is what 16-bit storage costs, not slack of these kernels.  The C2
    layer pair three ways on the same bf16-representable weights — the fp32 oracle (truth), the oracle with every module
    output and every gradient between modules rounded to bf16 (oracle/ref_cpu.py ACT_DTYPE: what the reference's own bf16
    arithmetic does at best), and the HIP bf16 path.  Per gradient, the HIP path's distance from the truth must stay within
    1.5 x the bf16 oracle's (+ 0.3 % of the gradient's norm for tensors both get almost exactly)."""
    import paramgen
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    cfg = TransformerConfig(dropout=0.0, encoder_layers=1, decoder_layers=1)
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V))
    names_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = {k: torch.from_numpy(v).bfloat16().float() for k, v in paramgen.make_state_dict(21, names_shapes).items()}
    state['decoder.embed_tokens.weight'] = state['encoder.embed_tokens.weight']
    model.load_state_dict(state)
    model = model.to(torch.bfloat16).cuda().train()
    b = paramgen.make_text_batch(6, B // 2, S, T, V, ragged=True)   # (half the C2 batch: two oracle passes on the host)
    cpu_batch = {k: torch.from_numpy(v) for k, v in b.items()}
    model.zero_grad(set_to_none=True)
    loss, logs = model(**{k: v.cuda() for k, v in cpu_batch.items()})
    loss.backward()
    got = {k: p.grad.float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    true_loss, _, true = _oracle_grads(state, cfg, cpu_batch)
    o16_loss, _, o16 = _oracle_grads(state, cfg, cpu_batch, torch.bfloat16)
    e_hip, e_o16 = abs(loss.item() - true_loss) / abs(true_loss), abs(o16_loss - true_loss) / abs(true_loss)
    assert e_hip <= 1.5 * e_o16 + 1e-3, (e_hip, e_o16)
    worst = (0.0, None)
    for k, r in true.items():
        if k.endswith('k_proj.bias') or k not in got:
            continue
        n = r.norm().item()
        h, o = (got[k] - r).norm().item() / n, (o16[k] - r).norm().item() / n
        assert h <= 1.5 * o + 3e-3, (k, h, o)
        if o > 0 and h / o > worst[0]:
            worst = (h / o, k)
    print(f'bf16 error bracket: loss hip {e_hip:.2e} / oracle-bf16 {e_o16:.2e}; worst gradient ratio {worst[0]:.2f} ({worst[1]})')
