"""The native layer path (pasero_amd/native_layer.py, csrc/layer.cpp: one C call per layer and direction) against the
per-op path it replaces (PASERO_NO_NATIVE_LAYER, the same kernels dispatched from Python): loss, EVERY gradient and the
dropout masks must agree BIT FOR BIT — both paths issue the same launches with the same arguments — for the base model
(fused block ends, d = 512), a d = 1024 model (stand-alone LayerNorm block ends), GELU feed-forwards (saved
pre-activation), ragged batches, with and without dropout; models outside the stock layer must stay on the per-op path."""
import pytest
import torch

import paramgen
from pasero_amd import functional as F
from model_utils import load_paramgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def _model(V, seed=3, dtype=torch.bfloat16, **over):
    from pasero_amd.config import TransformerConfig, DistributedConfig, SyntheticTask
    from pasero_amd.transformer import Transformer
    cfg = TransformerConfig(**over)
    model = Transformer(cfg, DistributedConfig(), SyntheticTask(V))
    load_paramgen(model, seed)
    return model.to(dtype).cuda().train()


def _step(model, batch, native: bool, seed=11, chain: bool = False):
    # chain: the decoder layers sum their encoder-output gradients inside the kv dX GEMMs (one rounding per layer) instead of
    # leaving five bf16 additions to autograd (two roundings per layer) — the default; off for the bit-for-bit comparisons
    from pasero_amd import native_layer, rng
    native_layer._OFF = not native
    native_layer._NO_DENC_CHAIN = not chain
    calls = {'n': 0}
    orig = native_layer.NativeLayerFn.forward

    def counted(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    native_layer.NativeLayerFn.forward = staticmethod(counted)
    try:
        rng.manual_seed(seed)
        model.zero_grad(set_to_none=True)
        loss, logs = model(**batch)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        native_layer.NativeLayerFn.forward = staticmethod(orig)
        native_layer._OFF = False
        native_layer._NO_DENC_CHAIN = False
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    return loss.item(), logs['num_tokens'], grads, calls['n']


@pytest.mark.parametrize('over,B,S,T', [
    (dict(dropout=0.1, encoder_layers=2, decoder_layers=2), 48, 40, 36),                      # base width: fused block ends
    (dict(dropout=0.0, encoder_layers=1, decoder_layers=2), 16, 33, 47),
    (dict(dropout=0.1, encoder_layers=1, decoder_layers=1, activation_fn='gelu'), 24, 24, 24),
    (dict(dropout=0.1, encoder_layers=1, decoder_layers=1, embed_dim=1024, encoder_ffn_dim=2048, decoder_ffn_dim=2048,
          encoder_attention_heads=16, decoder_attention_heads=16), 16, 32, 32),                # stand-alone LayerNorm ends
])
def test_native_layer_equals_the_per_op_path_bit_for_bit(over, B, S, T):
    V = 1000
    model = _model(V, **over)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, B, S, T, V, ragged=True).items()}
    l1, n1, g1, c1 = _step(model, batch, native=True)
    l0, n0, g0, c0 = _step(model, batch, native=False)
    layers = over['encoder_layers'] + over['decoder_layers']
    assert c1 == layers and c0 == 0, (c1, c0)
    assert l1 == l0 and n1 == n0
    assert set(g1) == set(g0)
    for k in g0:
        assert torch.equal(g1[k].view(torch.int16), g0[k].view(torch.int16)), k
    # and twice in a row (grow-only scratch buffers reused across layers and steps)
    l2, _, g2, _ = _step(model, batch, native=True)
    assert l2 == l1 and all(torch.equal(g2[k], g1[k]) for k in g1)
    # the default: the encoder-output gradient summed by the decoder layers' kv dX GEMMs.  Everything on the decoder side is
    # untouched (bit for bit); what flows into the encoder differs by the roundings it no longer has
    l3, _, g3, _ = _step(model, batch, native=True, chain=True)
    assert l3 == l1
    for k in g1:
        if k.startswith('decoder.layers') or over['decoder_layers'] == 1:
            assert torch.equal(g3[k].view(torch.int16), g1[k].view(torch.int16)), k
        elif not k.endswith('k_proj.bias'):  # (a key bias's gradient is mathematically zero: round-off only)
            a, b_ = g3[k].float(), g1[k].float()
            assert (a - b_).norm().item() <= 3e-2 * b_.norm().item() + 1e-6, (k, (a - b_).norm().item(), b_.norm().item())
    l4, _, g4, _ = _step(model, batch, native=True, chain=True)
    assert l4 == l3 and all(torch.equal(g4[k], g3[k]) for k in g3)


def test_what_is_not_the_stock_layer_stays_on_the_per_op_path():
    V = 500
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 16, 32, 32, V).items()}
    for over in (dict(attention_dropout=0.1), dict(activation_fn='swiglu'),
                 dict(encoder_positional_encoding='rotary', decoder_positional_encoding='rotary'), dict(has_bias=False)):
        model = _model(V, encoder_layers=1, decoder_layers=1, **over)
        _, _, _, calls = _step(model, batch, native=True)
        assert calls == 0, over
    model = _model(V, dtype=torch.float32, encoder_layers=1, decoder_layers=1)
    assert _step(model, batch, native=True)[3] == 0
    model = _model(V, encoder_layers=1, decoder_layers=1)
    with torch.no_grad():
        from pasero_amd import native_layer
        assert not native_layer.takes(model.encoder.layers[0], torch.zeros(16, 32, 512, device='cuda', dtype=torch.bfloat16), None,
                                      None, [], False)
    tiny = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 2, 8, 8, V).items()}
    assert _step(model, tiny, native=True)[3] == 0  # too few rows for the grouped weight-gradient launch
    # a row count that is no multiple of 8 (the weight gradients contract over the rows: no whole 16-byte columns) — one 30 s
    # Whisper clip is 1500 rows; it used to pass `takes` and fail inside pk_layer_bwd
    odd = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 3, 100, 100, V, ragged=False).items()}
    loss, _, _, calls = _step(model, odd, native=True)
    assert calls == 0 and torch.isfinite(torch.as_tensor(loss))


def test_second_backward_over_a_retained_graph_and_a_gradient_towards_the_input():
    """the node keeps what it needs after a backward (retain_graph), and `autograd.grad` towards the layer input works"""
    V = 600
    model = _model(V, encoder_layers=1, decoder_layers=1, dropout=0.1)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 24, 20, 20, V).items()}
    from pasero_amd import rng
    rng.manual_seed(2)
    model.zero_grad(set_to_none=True)
    loss, _ = model(**batch)
    loss.backward(retain_graph=True)
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    loss.backward()
    for n, p in model.named_parameters():
        assert torch.equal(p.grad, g1[n]), n


@pytest.mark.parametrize('over,B,S,T', [
    (dict(dropout=0.1, encoder_layers=2, decoder_layers=2, encoder_prenorm=True, decoder_prenorm=True), 32, 40, 36),
    (dict(dropout=0.0, encoder_layers=1, decoder_layers=1, encoder_prenorm=True, decoder_prenorm=True, activation_fn='gelu',
          attention_key_bias=False), 24, 33, 20),                                              # the Whisper layer
    (dict(dropout=0.1, encoder_layers=1, decoder_layers=1, encoder_prenorm=True, decoder_prenorm=False), 16, 24, 24),
])
def test_native_prenorm_layer_against_the_per_op_path(over, B, S, T):
    """pre-norm layers: the native call feeds the residual branch's gradient into the LayerNorm backward kernel
    (`dz_extra`) where the per-op path lets autograd add it in a separate pass over bf16 values, and masks the gradient for
    the block below inside the LayerNorm backward — fewer roundings, so the comparison is to bf16 round-off, not bit for
    bit: loss 1e-3, every gradient within 4 % of its norm.  (Measured against an fp32 run of the same model: both paths
    sit 0.9 % from it on average and 2.0 % at worst — the attention projections' weights — and up to 1.6 % apart.)"""
    V = 1000
    model = _model(V, **over)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, B, S, T, V, ragged=True).items()}
    l1, n1, g1, c1 = _step(model, batch, native=True)
    l0, n0, g0, c0 = _step(model, batch, native=False)
    assert c1 == over['encoder_layers'] + over['decoder_layers'] and c0 == 0, (c1, c0)
    assert n1 == n0 and abs(l1 - l0) <= 1e-3 * abs(l0), (l1, l0)
    assert set(g1) == set(g0)
    for k in g0:
        a, r = g1[k].float(), g0[k].float()
        # (the key bias has no gradient in exact arithmetic — softmax ignores a constant added to every key: what both
        # paths leave there is round-off, judged against the gradient of the projection's weight)
        ref = g0[k.replace('bias', 'weight')].float().norm().item() if k.endswith('k_proj.bias') else r.norm().item()
        assert (a - r).norm().item() <= 4e-2 * ref + 1e-6, (k, (a - r).norm().item(), ref)
    l2, _, g2, _ = _step(model, batch, native=True)
    assert l2 == l1 and all(torch.equal(g2[k], g1[k]) for k in g1)


def test_adapter_behind_a_prenorm_layer_offers_the_masked_gradient(monkeypatch):
    """the bottleneck adapter that follows a pre-norm layer (`adapter_transformer`: x = layer(x); x = adapter(x)) reads the
    layer's output twice — through its LayerNorm and as its residual — and its LayerNorm backward produces the whole gradient
    of that tensor: with the layer's DropLink it writes the gradient through the layer's last dropout mask too, and
    ResidualDropoutFn takes it instead of drawing the mask again.  Same forward; gradients to bf16 round-off."""
    from pasero_amd import autograd, rng
    from pasero_amd.autograd import AdapterFn, ResidualDropoutFn, DropLink
    B, T, d, r = 8, 40, 512, 64
    gen = torch.Generator().manual_seed(5)

    def mk(*shape, scale=1.0):
        return (scale * torch.randn(*shape, generator=gen)).bfloat16().cuda()
    x0, r0, dy = mk(B, T, d), mk(B, T, d), mk(B, T, d)
    ws = [mk(d).abs() + 0.5, mk(d, scale=0.1), mk(r, d, scale=d ** -0.5), mk(r, scale=0.1), mk(d, r, scale=r ** -0.5), mk(d, scale=0.1)]
    hits = {'n': 0}
    orig_take = DropLink.take

    def counted(self, dz):
        m = orig_take(self, dz)
        hits['n'] += m is not None
        return m
    monkeypatch.setattr(DropLink, 'take', counted)
    outs = []
    for linked in (True, False):
        rng.manual_seed(3)
        hits['n'] = 0
        x, res = x0.clone().requires_grad_(), r0.clone().requires_grad_()
        w = [t.clone().requires_grad_() for t in ws]
        link = DropLink() if linked else None
        z = ResidualDropoutFn.apply(x, res, 0.1, link)
        y = AdapterFn.apply(z, z, w[0], w[1], 1e-5, w[2], w[3], w[4], w[5], 'relu', 1.0, link)
        y.backward(dy)
        outs.append([y.detach(), x.grad, res.grad] + [t.grad for t in w])
        assert hits['n'] == (1 if linked else 0)
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert (a.float() - b.float()).norm().item() <= 1e-2 * b.float().norm().item() + 1e-6


@pytest.mark.parametrize('per_op_layer', [0, 1])
def test_masked_gradient_hand_over_between_a_native_and_a_per_op_layer(monkeypatch, per_op_layer):
    """the same note travels between the two kinds of layer: a natively run pre-norm layer under a per-op one (whose
    LayerNormForkFn offers the masked gradient) and a per-op layer (ResidualDropoutFn takes it) under a native one — every
    hand-over found, gradients as close to the all-per-op run as the all-native run is"""
    from pasero_amd import native_layer, autograd
    V = 700
    model = _model(V, dropout=0.1, encoder_layers=2, decoder_layers=1, encoder_prenorm=True, decoder_prenorm=True)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 16, 40, 24, V, ragged=True).items()}
    l0, _, g0, _ = _step(model, batch, native=False)
    hits = {'n': 0}
    orig_take = autograd.DropLink.take

    def counted(self, dz):
        m = orig_take(self, dz)
        hits['n'] += m is not None
        return m
    monkeypatch.setattr(autograd.DropLink, 'take', counted)
    mixed_out = model.encoder.layers[per_op_layer]
    orig_takes = native_layer.takes
    monkeypatch.setattr(native_layer, 'takes', lambda layer, *a, **k: layer is not mixed_out and orig_takes(layer, *a, **k))
    l1, _, g1, c1 = _step(model, batch, native=True)
    assert c1 == 2  # one encoder layer and the decoder layer run natively
    # (inside the per-op layer the block end and the next LayerNorm are one node — nothing to hand over there; what is counted is
    # the boundary between the two encoder layers, served in both directions)
    assert hits['n'] == 1, hits
    assert abs(l1 - l0) <= 1e-3 * abs(l0)
    for k in g0:
        a, r = g1[k].float(), g0[k].float()
        ref = g0[k.replace('bias', 'weight')].float().norm().item() if k.endswith('k_proj.bias') else r.norm().item()
        assert (a - r).norm().item() <= 4e-2 * ref + 1e-6, (k, (a - r).norm().item(), ref)


def test_stacked_prenorm_layers_hand_the_masked_gradient_down(monkeypatch):
    """pre-norm layers with dropout, stacked: the last LayerNorm backward of layer l + 1 writes the gradient of its input a second
    time, through the feed-forward dropout mask of layer l (PkLayer.dx_masked -> dy_masked, autograd.DropLink), and layer l
    does not draw that mask again.  Same forward (equal loss); the masked copy is rounded once instead of twice, so gradients
    agree to bf16 round-off; the hand-over must actually happen between every pair of stacked layers and nowhere else."""
    from pasero_amd import native_layer, autograd
    V = 800
    model = _model(V, dropout=0.1, encoder_layers=3, decoder_layers=3, encoder_prenorm=True, decoder_prenorm=True)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 24, 30, 28, V, ragged=True).items()}
    hits = {'n': 0}
    orig = autograd.DropLink.take

    def counted(self, dz):
        m = orig(self, dz)
        hits['n'] += m is not None
        return m
    monkeypatch.setattr(autograd.DropLink, 'take', counted)
    l1, _, g1, c1 = _step(model, batch, native=True)
    assert c1 == 6 and hits['n'] == 4, (c1, hits)
    hits['n'] = 0
    monkeypatch.setattr(native_layer, '_NO_DROP_LINK', True)
    l0, _, g0, _ = _step(model, batch, native=True)
    assert hits['n'] == 0
    assert l1 == l0
    for k in g0:
        a, r = g1[k].float(), g0[k].float()
        ref = g0[k.replace('bias', 'weight')].float().norm().item() if k.endswith('k_proj.bias') else r.norm().item()
        assert (a - r).norm().item() <= 2e-2 * ref + 1e-6, (k, (a - r).norm().item(), ref)
    monkeypatch.setattr(native_layer, '_NO_DROP_LINK', False)
    l2, _, g2, _ = _step(model, batch, native=True)
    assert l2 == l1 and all(torch.equal(g2[k], g1[k]) for k in g1)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2.5e-2)])
def test_prenorm_input_fork_against_autograds_addition(dtype, tol):
    """per-op path, pre-norm layers: `residual = x; x = *_prenorm(x)` as ONE autograd node (autograd.LayerNormForkFn — the
    residual branch's gradient enters the LayerNorm backward kernel as `dz_extra`) against the reference's two uses of x whose
    gradients autograd adds: same loss (the forward is untouched), gradients equal to round-off — fp32 2e-5, bf16 one rounding
    less per sub-block; and the node really is in the graph (no elementwise addition kernel left for those sums)."""
    from pasero_amd import transformer, autograd
    V = 1000
    over = dict(dropout=0.1, encoder_layers=2, decoder_layers=2, encoder_prenorm=True, decoder_prenorm=True)
    model = _model(V, dtype=dtype, **over)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 24, 40, 36, V, ragged=True).items()}
    calls = {'n': 0, 'ends': 0}
    orig = autograd.LayerNormForkFn.backward
    orig_end = autograd.ResidualDropoutLnFn.backward
    drops = []
    real_dropout = F.dropout

    def counted(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)

    def counted_end(*a, **k):
        calls['ends'] += 1
        return orig_end(*a, **k)
    autograd.LayerNormForkFn.backward = staticmethod(counted)
    autograd.ResidualDropoutLnFn.backward = staticmethod(counted_end)
    F.dropout = lambda *a, **k: (drops.append(1), real_dropout(*a, **k))[1]
    try:
        l1, n1, g1, c1 = _step(model, batch, native=False)
        forks, drops_fork = calls['n'], len(drops)
        transformer._NO_LN_FORK = True
        del drops[:]
        l0, n0, g0, c0 = _step(model, batch, native=False)
        drops_plain = len(drops)
    finally:
        transformer._NO_LN_FORK = False
        autograd.LayerNormForkFn.backward = staticmethod(orig)
        autograd.ResidualDropoutLnFn.backward = staticmethod(orig_end)
        F.dropout = real_dropout
    # round 6, 16-bit: inside a layer the block end and the LayerNorm of the next block are ONE node (autograd.ResidualDropoutLnFn:
    # 1 per encoder layer, 2 per decoder layer); the fork is left at the head of every layer
    inner = 2 * 1 + 2 * 2 if dtype != torch.float32 else 0
    assert c1 == 0 and c0 == 0 and forks == 2 * 2 + 2 * 3 - inner and calls['n'] == forks and calls['ends'] == inner, (c1, c0, forks, calls)
    # round 5: a `residual + dropout(.)` whose output goes straight into the next fork gets its masked gradient from that fork's
    # LayerNorm backward kernel (autograd.DropLink): 2 x 1 encoder + 2 x 2 decoder stand-alone dropout launches fewer
    # (16-bit only: fp32 is the parity path and keeps the stand-alone mask); round 6: the note survives the layer's entry node
    # (WGradSinkFn), so the feed-forward block end of a layer is served by the first fork of the layer above: + 1 per stack of two
    assert drops_plain - drops_fork == (2 * 1 + 2 * 2 + 2 if dtype != torch.float32 else 0), (drops_plain, drops_fork)
    assert n1 == n0 and l1 == l0, (l1, l0)
    assert set(g1) == set(g0)
    for k in g0:
        a, r = g1[k].float(), g0[k].float()
        ref = g0[k.replace('bias', 'weight')].float().norm().item() if k.endswith('k_proj.bias') else r.norm().item()
        assert (a - r).norm().item() <= tol * ref + 1e-7, (k, (a - r).norm().item(), ref)


def test_prenorm_input_fork_with_one_output_unused():
    """the fork node does not materialise a missing output gradient as zeros — only the LayerNorm branch consumed
    -> the LayerNorm gradient alone (no dz_extra operand); only the residual branch consumed -> the identity; both against
    the two-use form"""
    from pasero_amd.autograd import LayerNormForkFn, ResidualLayerNormFn
    torch.manual_seed(3)
    x = torch.randn(96, 512, device='cuda', dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(512, device='cuda', dtype=torch.bfloat16, requires_grad=True)
    b = torch.randn(512, device='cuda', dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn(96, 512, device='cuda', dtype=torch.bfloat16)

    def grads(fn):
        for t in (x, g, b):
            t.grad = None
        fn().backward()
        return [None if t.grad is None else t.grad.clone() for t in (x, g, b)]
    y_only = grads(lambda: (LayerNormForkFn.apply(x, g, b, 1e-5)[0] * w).float().sum())
    ref = grads(lambda: (ResidualLayerNormFn.apply(x, None, g, b, 1e-5, 0.0) * w).float().sum())
    for a, r in zip(y_only, ref):
        assert torch.equal(a, r)
    res_only = grads(lambda: (LayerNormForkFn.apply(x, g, b, 1e-5)[1] * w).float().sum())
    assert torch.equal(res_only[0], w) and res_only[1] is None and res_only[2] is None


def test_forward_split_boundary_moves_rows_by_round_off_only():
    """functional.fwd_split gates on M (512..2048 rows) and pk_gemm re-derives its own slab count from the tile count,
    so the number of partial sums in fc2 of NLLB-1.3B (8192 -> 1024) depends on the batch's row count: one chain at 4096 rows,
    8 slabs of the 256-tile kernel at 2048, 4 slabs of the 128-tile kernel at 1024.  What holds, and is pinned here: the same
    row in batches of different sizes agrees to the fp32 round-off of the accumulation — at most one bf16 ulp, on a small
    fraction of the outputs — and every variant is within one ulp of the fp64 product."""
    torch.manual_seed(5)
    K, N = 8192, 1024
    a = (torch.randn(4096, K, device='cuda') * 0.5).bfloat16()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
    assert F.fwd_split(4096, N, K, a.dtype) == 1 and F.fwd_split(2048, N, K, a.dtype) == 4 == F.fwd_split(1024, N, K, a.dtype)
    whole = F.gemm(a, w, splitk=F.fwd_split(4096, N, K, a.dtype))
    halves = torch.cat([F.gemm(a[i:i + 2048], w, splitk=4) for i in (0, 2048)])
    quarters = torch.cat([F.gemm(a[i:i + 1024], w, splitk=4) for i in range(0, 4096, 1024)])
    ref = a.double() @ w.double().t()
    # one bf16 ulp of the exact value (8 significant bits); below 2^-6 the fp32 accumulation error of an 8192-term sum of
    # O(0.5 x 0.01) products (~1e-5) is no longer small against an ulp of the value, so the scale stops shrinking there
    ulp = 2.0 ** (torch.floor(torch.log2(ref.abs().clamp_min(2.0 ** -6))) - 7)
    for other in (halves, quarters):
        d = (whole.double() - other.double()).abs()
        assert (d <= ulp * 1.001).all(), (d / ulp).max().item()
        assert (d > 0).float().mean().item() < 0.05
    for t in (whole, halves, quarters):
        assert ((t.double() - ref).abs() <= ulp * 1.001).all()
    # the rule itself is deterministic: the same batch twice is the same bits
    assert torch.equal(halves, torch.cat([F.gemm(a[i:i + 2048], w, splitk=4) for i in (0, 2048)]))


def test_encoder_gradient_chain_belongs_to_the_decoder_pass():
    """the tally of the chained encoder-output gradients lived on the encoder TENSOR — a decoder pass over it that
    was never back-propagated left a stale count behind and the next pass over the same tensor lost the encoder's gradient
    silently.  It belongs to the pass now: (i) an abandoned decoder pass changes nothing for the next one, (ii) a second
    backward over a retained graph gives the same gradients, (iii) a backward that visits only some of the chained layers
    is refused loudly."""
    from pasero_amd import rng
    V = 600
    model = _model(V, encoder_layers=1, decoder_layers=3, dropout=0.0)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(4, 24, 20, 20, V).items()}
    dec_in = batch['decoder_input'][:, :-1].contiguous()

    def enc_grad(abandon_first: bool):
        model.zero_grad(set_to_none=True)
        enc_out, enc_mask, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
        if abandon_first:
            model.decoder(enc_out, enc_mask, dec_in, project=False)   # a grad-mode pass nobody back-propagates
        feats, _ = model.decoder(enc_out, enc_mask, dec_in, project=False)
        feats.float().pow(2).sum().backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in model.encoder.named_parameters() if p.grad is not None}

    rng.manual_seed(2)
    plain = enc_grad(False)
    rng.manual_seed(2)
    after = enc_grad(True)
    assert plain and plain.keys() == after.keys()
    for n in plain:
        assert plain[n].float().abs().max() > 0 and torch.equal(plain[n], after[n]), n
    # (iii) a gradient towards the input of the TOP decoder layer's successor only touches that layer's node
    enc_out, enc_mask, _ = model.encoder(batch['encoder_input'], batch['encoder_input_length'])
    mid = {}
    def grab(m, i, o):
        mid['x'] = o[0]   # (returns None: the output stays as it is)
    h = model.decoder.layers[1].register_forward_hook(grab)
    try:
        feats, _ = model.decoder(enc_out, enc_mask, dec_in, project=False)
    finally:
        h.remove()
    with pytest.raises(RuntimeError, match='only some of the decoder layers'):
        torch.autograd.grad(feats.float().sum(), [mid['x']])
    # and the chain is usable again afterwards
    rng.manual_seed(2)
    again = enc_grad(False)
    for n in plain:
        assert torch.equal(plain[n], again[n]), n


@pytest.mark.parametrize('mix', ['per_op', 'mixed'])
def test_per_op_cross_attention_joins_the_encoder_gradient_chain(mix, monkeypatch):
    """round 5: on the per-op path the cross-attention k | v projections of a decoder pass sum their encoder-output gradients in
    their dX GEMMs too (PackedLinearFn + native_layer.DencChain), instead of leaving 23 (rows, d) additions per IWSLT step to
    autograd: same encoder gradients as with every layer returning its own (PASERO_NO_DENC_CHAIN), to bf16 round-off — one
    rounding per layer instead of two —, also when natively run and per-op layers share one chain, and no aten add of that
    size is left"""
    from pasero_amd import native_layer, rng
    V = 600
    model = _model(V, encoder_layers=1, decoder_layers=4, dropout=0.0)
    batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(5, 24, 20, 20, V).items()}
    if mix == 'mixed':   # layers 1 and 3 stay on the per-op path: a subclass that overrides a hook is never run natively
        for i in (1, 3):
            lay = model.decoder.layers[i]

            class PerOp(type(lay)):
                def ffn(self, *a, **k):
                    return super().ffn(*a, **k)
            lay.__class__ = PerOp

    def enc_grads(native):
        rng.manual_seed(3)
        model.zero_grad(set_to_none=True)
        monkeypatch.setattr(native_layer, '_OFF', not native)
        loss, _ = model(**batch)
        loss.backward()
        return {n: p.grad.float().clone() for n, p in model.encoder.named_parameters()}

    joined = enc_grads(native=(mix == 'mixed'))
    monkeypatch.setattr(native_layer, '_NO_DENC_CHAIN', True)
    alone = enc_grads(native=(mix == 'mixed'))
    for n in alone:
        # (a key bias shifts every score of a row alike: its gradient is zero in exact arithmetic and pure round-off here —
        # held against its weight's gradient, as everywhere in these tests)
        ref = (alone[n.replace('bias', 'weight')] if n.endswith('k_proj.bias') else alone[n]).norm().item()
        assert (joined[n] - alone[n]).norm().item() <= 2.5e-2 * ref + 1e-7, n
