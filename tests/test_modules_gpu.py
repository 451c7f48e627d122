"""Module-level parity on the GPU against the golden vectors of the real reference (tests/golden/*.npz, made by
oracle/make_golden.py): pasero_amd.modules.MultiheadAttention (three masking variants), the label-smoothed
cross-entropy (CrossEntropyFn = Transformer.compute_loss on materialised logits) and the sinusoidal table.
fp32 kernels; tolerances follow north_star (1e-4 relative for floating point) unless stated."""
import math

import numpy as np
import pytest
import torch

import paramgen
from conftest import load_golden
from model_utils import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


@pytest.mark.parametrize('variant', ['self_pad', 'self_causal', 'cross'])
def test_multihead_attention_vs_reference(variant):
    """modules.py:579-739: key-padding mask, 'a 2-D mask on a causal layer is dropped' (:602-605), ragged cross"""
    from pasero_amd.modules import MultiheadAttention
    g = load_golden('mha')
    d, H, B, T, S = (int(g[k]) for k in 'dHBTS')
    names = [str(n) for n in g[variant + ':param_names']]
    shapes = [tuple(int(x) for x in str(s).split(',')) for s in g[variant + ':param_shapes']]
    mha = MultiheadAttention(d, H, causal=(variant == 'self_causal'))
    assert [(k, tuple(v.shape)) for k, v in mha.state_dict().items()] == list(zip(names, shapes))
    mha.load_state_dict({k: torch.from_numpy(v) for k, v in paramgen.make_state_dict(31, list(zip(names, shapes))).items()})
    mha = mha.cuda()
    q = torch.from_numpy(paramgen.make_array(31, variant + '.q', (B, T, d))).cuda().requires_grad_()
    if variant == 'cross':
        kv = torch.from_numpy(paramgen.make_array(31, variant + '.kv', (B, S, d))).cuda().requires_grad_()
        src = S
    else:
        kv, src = q, T
    lens = torch.from_numpy(g[variant + ':lens']).cuda()
    mask = torch.arange(src, device='cuda')[None] >= lens[:, None]
    y, w = mha(query=q, key=kv, value=kv, attn_mask=mask)
    assert w is None
    y.backward(torch.from_numpy(paramgen.make_array(31, variant + '.dy', (B, T, d))).cuda())
    assert rel(y, g[variant + ':y']) < 1e-4
    assert rel(q.grad, g[variant + ':dq']) < 1e-4
    if variant == 'cross':
        assert rel(kv.grad, g[variant + ':dkv']) < 1e-4
    with torch.no_grad():  # return_attn: the (B,T,H,S) weights of the reference's explicit path (modules.py:742-771)
        y2, w2 = mha(query=q, key=kv, value=kv, attn_mask=mask, return_attn=True)
    assert w2.shape == g[variant + ':attn_weights'].shape
    assert np.abs(w2.cpu().numpy() - g[variant + ':attn_weights']).max() < 1e-5
    assert rel(y2, g[variant + ':y_return_attn']) < 1e-4
    for n, p in mha.named_parameters():
        ref = g[variant + ':grad:' + n]
        # the key bias shifts every score of a row by the same amount: its gradient is zero up to rounding
        assert rel(p.grad, ref) < 1e-4 or np.abs(ref).max() < 1e-5, n


@pytest.mark.parametrize('eps', [0.0, 0.1, 0.2])
def test_label_smoothed_cross_entropy_vs_reference(eps):
    """transformer.py:324-380: sum reduction, pad ignored, logs in bits"""
    from pasero_amd.autograd import CrossEntropyFn
    g = load_golden('ce_ls')
    B, T, V = int(g['B']), int(g['T']), int(g['V'])
    logits = torch.from_numpy(paramgen.make_array(41, 'ce.logits', (B, T, V), scale=2.0)).cuda().requires_grad_()
    target = torch.from_numpy(g['target']).cuda()
    sums = CrossEntropyFn.apply(logits, target, 1, eps)
    sums[0].backward()
    tag = f'eps{eps}'
    loss, nll, ntok = sums.tolist()
    assert abs(loss - float(g[tag + ':loss'])) <= 1e-5 * abs(float(g[tag + ':loss']))
    assert abs(loss / math.log(2) - float(g[tag + ':logs_loss'])) <= 1e-5 * abs(float(g[tag + ':logs_loss']))
    assert abs(nll / math.log(2) - float(g[tag + ':logs_nll_loss'])) <= 1e-5 * abs(float(g[tag + ':logs_nll_loss']))
    assert int(ntok) == int(g[tag + ':num_tokens'])
    dl = logits.grad
    assert rel(dl[:, :2], g[tag + ':dlogits_rows']) < 1e-4
    np.testing.assert_allclose(dl.sum(-1).cpu().numpy(), g[tag + ':dlogits_rowsum'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dl.abs().sum(-1).cpu().numpy(), g[tag + ':dlogits_abs_rowsum'], rtol=1e-4)


@pytest.mark.parametrize('d', [128, 512, 1024])
def test_sinusoidal_table_vs_reference(d):
    """modules.py:415-457, shift = 2; the table the embedding kernel adds (device copy)"""
    from pasero_amd.modules import SinusoidalPositionalEmbedding
    g = load_golden('sinpos')
    emb = SinusoidalPositionalEmbedding(300, d, shift=2).cuda()
    table = emb.table().cpu().numpy()
    # the table is built on the host in fp32 like the reference's: sin/cos/exp of another CPU's libm differ in the last
    # ulp of the angle (up to 301 rad -> 2e-5 absolute); the container that made the fixture agrees to 1e-6
    # (tests/test_oracle_golden.py::test_sinusoidal_positions)
    np.testing.assert_allclose(table[2:42], g[f'd{d}'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(table[19:22], g[f'd{d}_off'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(table[[0, 1, 2, 150, 301]], g[f'd{d}_rows'], rtol=0, atol=5e-5)


def test_merged_embedding_table_follows_the_fused_optimizer_step():
    """`Embedding.effective_weight` caches where(freeze_mask, frozen, weight) keyed on the tables'
    `_version`; `optim.Adam.fused_step` writes the parameters through raw pointers (pk_mt_adam), so it has to bump the
    versions itself — otherwise every training forward after the first step reads the step-0 table
    (pasero/models/modules.py:929-946 gathers from the live tables each call)."""
    from pasero_amd.modules import Embedding
    from pasero_amd.optim import Adam
    torch.manual_seed(3)
    V, d = 40, 64
    mask = torch.zeros(V, dtype=torch.bool)
    mask[::3] = True
    e = Embedding(V, d, 1, freeze_mask=mask).cuda()
    mask = mask.cuda()
    opt = Adam(list(e.parameters()), lr=0.1, weight_decay=0.0)
    ids = torch.randint(2, V, (4, 9), device='cuda')
    for step in range(3):
        E = e.effective_weight()
        assert torch.equal(E, torch.where(mask[:, None], e.frozen_embedding.weight, e.weight)), step
        before = e.weight.detach().clone()
        e(ids).square().sum().backward()
        v0 = e.weight._version
        opt.fused_step(1.0, 0.0)
        torch.cuda.synchronize()
        assert e.weight._version > v0
        assert not torch.equal(before, e.weight)  # the step moved the trainable table ...
        E2 = e.effective_weight()
        assert E2 is not E  # ... and the merged table was rebuilt, inside grad mode as well as outside
        assert torch.equal(E2, torch.where(mask[:, None], e.frozen_embedding.weight, e.weight))
        with torch.no_grad():
            assert torch.equal(e.effective_weight(), E2.detach())
        opt.zero_grad(set_to_none=True)


def test_lookup_through_an_alias_of_a_tied_table_keeps_its_own_gradient():
    """rows are deferred to the tied table's hook only by lookups of the tensor `tie_table` marked; a second
    autograd tensor over the same storage has its own gradient, and the table's gradient is then the dense dW alone"""
    from pasero_amd import autograd
    from pasero_amd.autograd import EmbeddingFn, VocabCrossEntropyFn, tie_table
    torch.manual_seed(5)
    V, d, B, T = 48, 64, 3, 7
    w = torch.nn.Parameter(torch.randn(V, d, device='cuda') * 0.1)
    tie_table(w)
    alias = w.detach().requires_grad_()
    assert alias.data_ptr() == w.data_ptr()
    x = torch.randn(B, T, d, device='cuda', requires_grad=True)
    tgt = torch.randint(2, V, (B, T), device='cuda')
    ids = torch.randint(2, V, (B, T), device='cuda')

    def run(table):
        for t in (w, alias, x):
            t.grad = None
        sums = VocabCrossEntropyFn.apply(x, w, tgt, 1, 0.1)
        (sums[0] + EmbeddingFn.apply(ids, table, None, 1.0, 0, 0.0, 1).square().sum()).backward()
        assert not autograd._table_sessions
        return w.grad.clone(), (None if alias.grad is None else alias.grad.clone())
    both, none = run(w)  # the marked tensor itself: one gradient holding the dense dW and the lookup's rows
    assert none is None
    dense, rows = run(alias)
    assert rows is not None and rel(dense + rows, both) < 1e-5 and rows.abs().sum() > 0
