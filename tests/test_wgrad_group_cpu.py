"""Host logic of the grouped weight-gradient launch (pasero_amd.autograd.WGradGroup / WGradSinkFn) without a GPU: which
gradient lands in which sink slot, row slices of a packed q|k|v gradient, accumulation when a slot is hit twice, the
refusal after the launch, and the autograd ordering the mechanism rests on (the sink's backward runs after the backward
of every op of its layer, including ops whose INPUT does not descend from the sink — the cross-attention key / value
projection).  The GEMMs themselves are stood in for by torch.matmul here; the kernels are checked in
tests/test_wgrad_group_gpu.py."""
import pytest
import torch

from pasero_amd import autograd as A
from pasero_amd import functional as F


@pytest.fixture()
def cpu_gemms(monkeypatch):
    monkeypatch.setattr(F, 'wgrad_group_eligible', lambda dy, x: dy.size(1) >= 8)  # "small" problems: the fallback path
    monkeypatch.setattr(F, 'wgrad_group', lambda entries: [(dy.t() @ x, dy.sum(0) if b else None) for dy, x, b in entries])
    monkeypatch.setattr(A, 'weight_grad', lambda dy, x, want_bias=False: ((dy.t() @ x, dy.sum(0)) if want_bias
                                                                           else dy.t() @ x))


def test_slots_slices_accumulation_and_refusal(cpu_gemms):
    torch.manual_seed(0)
    w1, b1 = torch.nn.Parameter(torch.randn(8, 5)), torch.nn.Parameter(torch.randn(8))
    wq, wk = torch.nn.Parameter(torch.randn(4, 5)), torch.nn.Parameter(torch.randn(4, 5))
    small = torch.nn.Parameter(torch.randn(3, 5))
    g = A.WGradGroup()
    g.bind([w1, b1, wq, wk, small])
    assert g.slot(w1) == 0 and g.slot(wk) == 3 and g.slot(torch.nn.Parameter(w1.detach().clone())) is None
    dy, x = torch.randn(7, 8), torch.randn(7, 5)
    g.add(dy, x, [(0, 0, 8)], [(1, 0, 8)])                       # a Linear with bias
    g.add(dy, x, [(0, 0, 8)], None)                              # the same weight used twice: gradients add up
    dyp = torch.randn(7, 8)
    g.add(dyp, x, [(2, 0, 4), (3, 4, 8)], None)                  # packed projection: rows 0..3 -> wq, 4..7 -> wk
    dys = torch.randn(7, 3)
    g.add(dys, x, [(4, 0, 3)], None)                             # not eligible: computed on the spot
    grads = g.flush()
    assert torch.allclose(grads[0], 2 * dy.t() @ x) and torch.allclose(grads[1], dy.sum(0))
    assert torch.allclose(grads[2], (dyp.t() @ x)[:4]) and torch.allclose(grads[3], (dyp.t() @ x)[4:])
    assert torch.allclose(grads[4], dys.t() @ x)
    with pytest.raises(RuntimeError, match='after the group had been launched'):
        g.add(dy, x, [(0, 0, 8)], None)


class _Lin(torch.autograd.Function):
    """y = x Wᵀ whose weight gradient is handed to the group (what LinearFn does on the GPU)"""

    @staticmethod
    def forward(ctx, x, w, group, log, tag):
        ctx.save_for_backward(x, w)
        ctx.group, ctx.log, ctx.tag = group, log, tag
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        ctx.log.append(ctx.tag)
        ctx.group.add(dy, x, [(ctx.group.slot(w), 0, w.size(0))], None)
        return dy @ w, None, None, None, None


def test_sink_runs_last_in_its_layer_and_returns_the_gradients(cpu_gemms):
    """a layer shaped like the decoder's: q from the layer input, k|v from ANOTHER tensor (the encoder output), both
    feeding one op; two such layers in sequence.  Each sink must run after both projections of its own layer."""
    torch.manual_seed(1)
    enc = torch.randn(6, 8, requires_grad=True)
    x = torch.randn(6, 8, requires_grad=True)
    ws = [torch.nn.Parameter(torch.randn(8, 8) * 0.3) for _ in range(4)]
    log = []

    def layer(x, wq, wkv, name):
        class Logged(A.WGradGroup):
            def flush(self):
                log.append(name + ':sink')
                return super().flush()

        g = Logged()
        x = A.WGradSinkFn.apply(x, g, wq, wkv)
        q = _Lin.apply(x, wq, g, log, name + ':q')
        kv = _Lin.apply(enc, wkv, g, log, name + ':kv')
        return torch.tanh(q * kv) + x

    y = layer(layer(x, ws[0], ws[1], 'L0'), ws[2], ws[3], 'L1')
    y.sum().backward()
    assert log.index('L1:sink') > max(log.index('L1:q'), log.index('L1:kv'))
    assert log.index('L0:sink') > max(log.index('L0:q'), log.index('L0:kv'))
    assert log.index('L1:sink') < log.index('L0:sink')
    # the same graph with ordinary autograd
    ref = [w.detach().clone().requires_grad_() for w in ws]
    x2, enc2 = x.detach().clone().requires_grad_(), enc.detach().clone().requires_grad_()

    def plain(x, wq, wkv):
        return torch.tanh((x @ wq.t()) * (enc2 @ wkv.t())) + x

    plain(plain(x2, ref[0], ref[1]), ref[2], ref[3]).sum().backward()
    for w, r in zip(ws, ref):
        assert torch.allclose(w.grad, r.grad, atol=1e-5)
    assert torch.allclose(x.grad, x2.grad, atol=1e-5) and torch.allclose(enc.grad, enc2.grad, atol=1e-5)


def test_a_node_that_hands_over_twice_replaces_its_entry(cpu_gemms):
    """a partial backward (autograd.grad towards an inner activation over a retained graph) runs deferred ops but never
    reaches the sink; the full backward that follows must not count their gradients twice"""
    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(8, 5))
    g = A.WGradGroup()
    g.bind([w])
    dy, x = torch.randn(7, 8), torch.randn(7, 5)
    g.add(dy, x, [(0, 0, 8)], None, owner=123)       # stale: from the partial backward
    dy2 = torch.randn(7, 8)
    g.add(dy2, x, [(0, 0, 8)], None, owner=123)      # the same node again, in the full backward
    g.add(dy, x, [(0, 0, 8)], None, owner=456)       # ANOTHER node sharing the weight still adds up
    grads = g.flush()
    assert torch.allclose(grads[0], (dy2 + dy).t() @ x, atol=1e-5)


# ---- the launch map of the grouped kernel (host-only entry point of the C ABI: no GPU work) ----
def _map_of(problems):
    import ctypes
    from pasero_amd import lib
    L = lib.load()
    arr = (lib.PkWgradProblem * len(problems))(*[lib.PkWgradProblem(16, 16, 16, None, M, N, K, M, N, N)
                                                 for M, N, K in problems])
    out = (ctypes.c_int * (2 * 65536))()
    grid = L.pk_gemm_wgrad_group_map(arr, len(problems), out, 65536)
    assert grid > 0 and grid % 8 == 0
    return [(out[2 * b], out[2 * b + 1]) for b in range(grid)]


def _layers(d, f, rows):
    enc = [(3 * d, d, rows), (d, d, rows), (f, d, rows), (d, f, rows)]
    dec = [(3 * d, d, rows), (d, d, rows), (d, d, rows), (2 * d, d, rows), (d, d, rows), (f, d, rows), (d, f, rows)]
    return enc, dec


@pytest.mark.parametrize('d,f,rows', [(512, 2048, 32768), (1024, 4096, 32768), (1024, 8192, 8192), (512, 2048, 24000),
                                      (768, 3072, 4096), (512, 2048, 1024)])
def test_launch_map_covers_every_tile_once(d, f, rows):
    """every (problem, K-slab, tile) of the plan is worked on by exactly one workgroup, whatever the packing decided"""
    import math
    for probs in _layers(d, f, rows):
        m = _map_of(probs)
        live = [e for e in m if e[0] >= 0]
        assert len(set(live)) == len(live)
        per_prob = {}
        for p, lin in live:
            per_prob.setdefault(p, []).append(lin)
        assert sorted(per_prob) == list(range(len(probs)))
        for p, lins in per_prob.items():
            M, N, K = probs[p]
            tiles = math.ceil(M / 256) * math.ceil(N / 256)
            assert sorted(lins) == list(range(len(lins))) and len(lins) % tiles == 0   # whole slabs, each position once
        # one round of the chip where the work fits one round: no XCD gets more than its 32 CUs
        if len(live) <= 256:
            for x in range(8):
                assert sum(1 for b, e in enumerate(m) if b % 8 == x and e[0] >= 0) <= 32


def test_launch_map_keeps_a_slab_unit_on_one_xcd_at_base_width():
    """C2 shapes: the tiles of one (problem, K-slab) — the workgroups that share rows of dY and X — never straddle two XCDs
    (each XCD has an L2 of its own: a straddling unit is read from memory twice)"""
    import math
    for probs in _layers(512, 2048, 32768):
        m = _map_of(probs)
        where = {}
        for b, (p, lin) in enumerate(m):
            if p < 0:
                continue
            M, N, K = probs[p]
            tiles = math.ceil(M / 256) * math.ceil(N / 256)
            where.setdefault((p, lin // tiles), set()).add(b % 8)
        assert all(len(x) == 1 for x in where.values()), where


def test_two_slab_lengths_where_one_leaves_the_last_round_half_empty():
    """NLLB-1.3B's layers at 8192 rows: one K-chunk length for all problems gives 640 (encoder) / 768 (decoder) workgroups of
    half the contraction — 2.5 / 3 rounds of the chip; the planner instead leaves fc1 / fc2 whole (256 workgroups: one round)
    and cuts only the d x d problems in two (list scheduling on 256 CUs says 16 % less).  At base width (one round either
    way) nothing changes.  And the map deals the LONG workgroups first: every XCD's list starts with its share of them."""
    import math
    enc, dec = _layers(1024, 8192, 8192)
    # (round 6: the encoder layer's small problems at a QUARTER of the length — 256 workgroups of 128 K-tiles + 256 of 32: every CU
    # one long and one short workgroup, 160 K-tiles each, where halves gave half the CUs 192; the decoder layer's 256 + 256 x 64
    # K-tiles were balanced as they were: measured 321-327 -> 310-313 us for the encoder layer, the IWSLT recipe's 16 000-row
    # encoder layer 592 -> 564)
    for probs, want in ((enc, [4, 4, 1, 1]), (dec, [2, 2, 2, 2, 2, 1, 1])):
        m = _map_of(probs)
        per = {}
        for pr, lin in m:
            if pr >= 0:
                per[pr] = per.get(pr, 0) + 1
        slabs = [per[i] // (math.ceil(M / 256) * math.ceil(N / 256)) for i, (M, N, K) in enumerate(probs)]
        assert slabs == want, slabs
        long_probs = {i for i, s in enumerate(slabs) if s == 1}
        for x in range(8):
            mine = [e[0] for b, e in enumerate(m) if b % 8 == x and e[0] >= 0]
            n_long = sum(1 for pr in mine if pr in long_probs)
            assert n_long == 32 and all(pr in long_probs for pr in mine[:n_long]), (x, mine)
    for probs in _layers(512, 2048, 32768):
        m = _map_of(probs)
        per = {}
        for pr, lin in m:
            if pr >= 0:
                per[pr] = per.get(pr, 0) + 1
        slabs = {per[i] // (math.ceil(M / 256) * math.ceil(N / 256)) for i, (M, N, K) in enumerate(probs)}
        assert len(slabs) == 1, slabs
