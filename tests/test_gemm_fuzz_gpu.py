"""Randomised shapes through pk_gemm's dispatch (skinny / 128-tile / 256-tile kernels, split-K, col-form operands, padded
leading dimensions, sizes that are not multiples of anything) against an fp64 product.  The dispatch rules grew out of
measurements; this guards the corners between them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pasero_amd import functional
    return functional


def _operand(rs, rows, cols, pad, dtype):
    """(rows, cols) view of a (rows, cols + pad) buffer whose pad columns are NaN"""
    buf = torch.full((rows, cols + pad), float('nan'), dtype=dtype)
    buf[:, :cols] = torch.from_numpy(rs.standard_normal((rows, cols)).astype(np.float32)).to(dtype)
    return buf.cuda()[:, :cols], buf[:, :cols].double()


@pytest.mark.parametrize('seed', range(6))
def test_gemm_dispatch_fuzz(F, seed):
    rs = np.random.RandomState(1000 + seed)
    sizes_mn = [1, 3, 8, 24, 64, 72, 130, 256, 264, 500, 512, 776, 1024, 1030, 2048, 2056]
    sizes_k = [8, 40, 64, 96, 128, 200, 512, 576, 1000, 1024, 2048, 4096, 4104]
    for case in range(28):
        dtype = [torch.bfloat16, torch.float16, torch.float32][rs.randint(3)]
        M, N, K = (int(rs.choice(sizes_mn)), int(rs.choice(sizes_mn)), int(rs.choice(sizes_k)))
        if dtype == torch.float32 and M * N * K > 2 ** 30:
            K = 512
        a_col, b_col = bool(rs.randint(2)), bool(rs.randint(2))
        epv = 4 if dtype == torch.float32 else 8
        pad_a, pad_b = int(rs.choice([0, epv, 3])), int(rs.choice([0, epv, 5]))
        a, a64 = _operand(rs, *((K, M) if a_col else (M, K)), pad_a, dtype)
        b, b64 = _operand(rs, *((K, N) if b_col else (N, K)), pad_b, dtype)
        splitk = int(rs.choice([1, 1, 2, 5, 16])) if K >= 512 else 1
        mode = int(rs.choice([0, 0, 1]))
        bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(dtype) if rs.randint(2) else None
        aux = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).to(dtype) if mode == 1 else None
        act = ['none', 'relu'][rs.randint(2)]
        ref = (a64.t() if a_col else a64) @ (b64 if b_col else b64.t())
        if bias is not None:
            ref = ref + bias.double()
        if act == 'relu':
            ref = ref.clamp(min=0)
        if aux is not None:
            ref = ref + aux.double()
        got = F.gemm(a, b, a_col=a_col, b_col=b_col, bias=None if bias is None else bias.cuda(), act=act,
                     aux=None if aux is None else aux.cuda(), mode=mode, splitk=splitk)
        what = (seed, case, str(dtype), M, N, K, a_col, b_col, pad_a, pad_b, splitk, mode, act, bias is not None)
        assert torch.isfinite(got.float()).all(), what
        tol = 3e-5 if dtype == torch.float32 else 8e-3
        scale = max(1.0, float(np.sqrt(K)))  # entries are sums of K unit-variance products
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= tol * scale * 4, (what, err)


@pytest.mark.parametrize('seed', range(4))
def test_gemm8p_fuzz(F, seed):
    """the phase-interleaved 256-tile kernel (csrc/gemm8p.hip) on every path it has: all four operand layouts, one /
    odd / even numbers of K-tiles, a partial last K-tile (K % 8 == 0: zero-filled through the buffer range check for
    col-form operands, masked per lane for row-form ones), split-K slabs, the fused bias gradient (col-form A), edge
    tiles in M and N, padded leading dimensions with NaN in the pad, every fused epilogue — against fp64"""
    from pasero_amd import lib
    L = lib.load()
    old = L.pk_gemm_use_8p(2)  # every eligible GEMM with M, N >= 256
    try:
        rs = np.random.RandomState(5000 + seed)
        sizes_m = [256, 264, 500, 512, 776, 1024, 1288]
        sizes_k = [64, 72, 128, 136, 192, 200, 320, 448, 512, 584, 1000, 1024, 2056]
        for case in range(24):
            dtype = [torch.bfloat16, torch.float16][rs.randint(2)]
            M, N, K = int(rs.choice(sizes_m)), int(rs.choice(sizes_m)), int(rs.choice(sizes_k))
            a_col, b_col = bool(rs.randint(2)), bool(rs.randint(2))
            pad_a, pad_b = int(rs.choice([0, 8, 16])), int(rs.choice([0, 8]))
            if a_col and M % 8:
                pad_a = 8  # (col form needs 16-byte addressable rows)
            a, a64 = _operand(rs, *((K, M) if a_col else (M, K)), pad_a, dtype)
            b, b64 = _operand(rs, *((K, N) if b_col else (N, K)), pad_b, dtype)
            splitk = int(rs.choice([1, 1, 2, 3])) if K >= 512 else 1
            mode = int(rs.choice([0, 0, 1, 2])) if splitk == 1 else int(rs.choice([0, 1]))
            act = ['none', 'relu'][rs.randint(2)]
            bias = (torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(dtype)
                    if (rs.randint(2) and mode != 2) else None)
            aux = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).to(dtype) if mode else None
            want_asum = a_col and bool(rs.randint(2))
            asum = torch.full((M,), float('nan'), dtype=dtype, device='cuda') if want_asum else None
            ref = (a64.t() if a_col else a64) @ (b64 if b_col else b64.t())
            if mode == 2:
                ref = ref * (aux.double() > 0) if act == 'relu' else ref
            else:
                if bias is not None:
                    ref = ref + bias.double()
                if act == 'relu':
                    ref = ref.clamp(min=0)
                if aux is not None:
                    ref = ref + aux.double()
            got = F.gemm(a, b, a_col=a_col, b_col=b_col, bias=None if bias is None else bias.cuda(), act=act,
                         aux=None if aux is None else aux.cuda(), mode=mode, splitk=splitk, asum_out=asum)
            what = (seed, case, str(dtype), M, N, K, a_col, b_col, pad_a, pad_b, splitk, mode, act, bias is not None,
                    want_asum)
            assert torch.isfinite(got.float()).all(), what
            scale = max(1.0, float(np.sqrt(K)))
            err = (got.double().cpu() - ref).abs().max().item()
            assert err <= 8e-3 * scale * 4, (what, err)
            if want_asum:
                ref_sum = a64.sum(0)
                err = (asum.double().cpu() - ref_sum).abs().max().item()
                assert err <= 8e-3 * scale * 4, (what, 'asum', err)
    finally:
        L.pk_gemm_use_8p(old)


@pytest.mark.parametrize('act', ['gelu', 'gelu_tanh', 'swiglu'])
def test_gemm8p_activation_preact_and_gate_epilogues(F, act):
    """the feed-forward epilogues beyond ReLU on the phase-interleaved kernel: GELU (erf: whisper_base, config.py:2550),
    tanh-GELU, SiLU — with the pre-activation as a second output (what the backward's act' reads), the act' product of
    the dH GEMM (mode 2) and the gate product act(x W1) * (x W3) of SwiGLU / GEGLU (mode 3) — against fp64"""
    from pasero_amd import lib
    L = lib.load()
    old = L.pk_gemm_use_8p(2)
    try:
        fn = {'gelu': lambda t: torch.nn.functional.gelu(t), 'gelu_tanh': lambda t: torch.nn.functional.gelu(t, approximate='tanh'),
              'swiglu': lambda t: torch.nn.functional.silu(t)}[act]
        rs = np.random.RandomState(77)
        M, N, K = 520, 776, 320
        a, a64 = _operand(rs, M, K, 8, torch.bfloat16)
        b, b64 = _operand(rs, N, K, 0, torch.bfloat16)
        bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).bfloat16()
        aux = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).bfloat16()
        z = (a64 @ b64.t()) * 0.05 + bias.double()
        pre = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        y = F.gemm(a, b, bias=bias.cuda(), act=act, preact=pre, alpha=0.05)
        assert (pre.double().cpu() - z).abs().max().item() < 4e-2 and (y.double().cpu() - fn(z)).abs().max().item() < 4e-2
        y3 = F.gemm(a, b, bias=bias.cuda(), act=act, aux=aux.cuda(), mode=3, alpha=0.05)
        assert (y3.double().cpu() - fn(z) * aux.double()).abs().max().item() < 8e-2
        zz = aux.double().clone().requires_grad_()
        fn(zz).sum().backward()
        y2 = F.gemm(a, b, act=act, aux=aux.cuda(), mode=2, alpha=0.05)
        assert (y2.double().cpu() - (a64 @ b64.t()) * 0.05 * zz.grad).abs().max().item() < 8e-2
    finally:
        L.pk_gemm_use_8p(old)


def test_gemm8p_takes_the_vocabulary_dx_shapes(F):
    """dX = dlogits . E with K = V = 8032 (not a multiple of 64) and V = 70376, row-form A, col-form B: on the
    phase-interleaved kernel, exact against fp32 on a row sample"""
    for M, N, K in [(8192, 512, 8032), (2048, 1024, 70376)]:
        g = torch.Generator(device='cuda').manual_seed(K)
        a = (torch.randn(M, K, device='cuda', generator=g) * 0.05).bfloat16()
        e = torch.randn(K, N, device='cuda', generator=g).bfloat16()
        got = F.gemm(a, e, b_col=True)
        rows = torch.arange(0, M, 97, device='cuda')
        ref = a[rows].float() @ e.float()
        assert ((got[rows].float() - ref).abs().max() / ref.abs().max()).item() < 6e-3


@pytest.mark.parametrize('a_col,b_col', [(False, False), (False, True), (True, True)])
def test_gemm8p_is_bitwise_reproducible_under_load(F, a_col, b_col):
    """race screen of the phase-interleaved K loop: its LDS images are refilled by DMA 5-6 phases ahead and guarded only by
    counted waits and barriers, so a misplaced read shows as rare wrong tiles that come and go with timing.  The same
    GEMM (short K = one iteration + tail paths, K = 512, an odd number of K-tiles, a partial last K-tile, split-K)
    runs 60 times while a second stream keeps the memory system busy; every result must equal the first bit for bit
    and the fp64 product within bf16 rounding."""
    from pasero_amd import lib
    L = lib.load()
    old = L.pk_gemm_use_8p(2)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
    try:
        for M, N, K, splitk in [(2048, 1024, 128, 1), (4096, 2048, 512, 1), (1024, 1024, 1088, 1), (2048, 512, 968, 1),
                                (1024, 512, 4096, 4)]:
            g = torch.Generator(device='cuda').manual_seed(M + K)
            A = torch.randn(M, K, device='cuda', generator=g).bfloat16()
            B = torch.randn(N, K, device='cuda', generator=g).bfloat16()
            a = A.t().contiguous() if a_col else A
            b = B.t().contiguous() if b_col else B
            ref = None
            for it in range(60):
                if it % 3 == 0:
                    with torch.cuda.stream(side):
                        junk.fill_(it & 255)  # uneven background traffic
                out = F.gemm(a, b, a_col=a_col, b_col=b_col, splitk=splitk)
                if ref is None:
                    ref = out.clone()
                    exact = A.double() @ B.double().t()
                    assert (ref.double() - exact).abs().max().item() <= 8e-3 * 4 * float(np.sqrt(K))
                else:
                    assert torch.equal(out, ref), (M, N, K, splitk, it)
            torch.cuda.synchronize()
    finally:
        L.pk_gemm_use_8p(old)


@pytest.mark.parametrize('rows,V,d', [(4096, 5006, 1024), (2048, 20486, 768), (2304, 41702, 768)])
def test_vocabulary_that_is_no_multiple_of_8_runs_on_the_256_tile_kernel(rows, V, d):
    """pk_gemm_ex: logits = x Eᵀ with V % 8 != 0 into rows padded to a multiple of 8 (PK_GEMM_PAD_N) and dX = dlogits E
    contracting over that V with zeros in the pad columns (PK_GEMM_PAD_K) — NLLB's V = 256 206.  Both must run on the
    phase-interleaved 256-tile kernel (launch sampling) and agree with fp64; the valid columns of the logits are bit for bit
    those of the same GEMM through the 128-tile kernel's arithmetic only up to summation order, so: fp64 tolerance."""
    import ctypes
    from pasero_amd import functional as F, lib
    torch.manual_seed(V)
    x = (torch.randn(rows, d, device='cuda') * 0.5).bfloat16()
    E = (torch.randn(V, d, device='cuda') * 0.05).bfloat16()
    ldp = (V + 15) // 16 * 16
    buf = torch.full((rows, ldp), float('nan'), dtype=torch.bfloat16, device='cuda')
    lg = buf[:, :V]
    L = lib.load()

    def tags(fn):
        lib.check(L.pk_gemm_timing_start(16, 1), 'start')
        fn()
        n = L.pk_gemm_timing_stop()
        out = []
        for i in range(n):
            ints = [ctypes.c_int() for _ in range(5)]
            fl, ms = ctypes.c_double(), ctypes.c_float()
            lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(t) for t in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
            out.append(ints[0].value)
        return out
    t = tags(lambda: F.gemm(x, E, out=lg, pad_n=True))
    # gemm8p, or its 128 x 256 tile (0x400: outputs whose half-tiles make whole rounds where the 256-tiles leave the last one half
    # empty), or — round 6 — the persistent walk of its 256 x 256 tiles (0x4000: outputs of two rounds and more)
    assert t and all((k & 0xF) == 8 and (k & ~0x4400) < 256 for k in t), t
    ref = x.double() @ E.double().t()
    assert torch.isfinite(buf[:, : (V + 7) // 8 * 8]).all()              # (pad columns: unspecified but finite)
    assert torch.isnan(buf[:, (V + 7) // 8 * 8:]).all()                  # nothing beyond the promised room is touched
    err = (lg.double() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), err
    # the gradient side: dlogits with ZERO pad columns, contraction over V
    g = torch.zeros(rows, ldp, dtype=torch.bfloat16, device='cuda')
    g[:, :V] = (torch.randn(rows, V, device='cuda') * 0.1).bfloat16()
    dx = torch.empty(rows, d, dtype=torch.bfloat16, device='cuda')
    t = tags(lambda: F.gemm(g[:, :V], E, b_col=True, out=dx, splitk=F.choose_splitk(rows, d, V), pad_k=True))
    assert t and all((k & 0xF) == 8 and (k & ~0x400) < 256 for k in t), t
    ref = g[:, :V].double() @ E.double()
    err = (dx.double() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), err
    # and the loss kernel really leaves zeros in the pad columns of its gradient
    tgt = torch.randint(0, V, (rows,), device='cuda')
    buf.fill_(float('nan'))
    F.gemm(x, E, out=lg, pad_n=True)
    rl, rn = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    F.ce_rows(lg, tgt, 1, 0.1, rl, rn, dlogits=lg)
    assert (buf[:, V: (V + 7) // 8 * 8] == 0).all() and torch.isfinite(buf[:, :V]).all()


@pytest.mark.parametrize('M,N,K,b_col,mode,act,bias', [
    (8192, 1024, 1024, False, 0, 'none', True), (8192, 1024, 8192, False, 0, 'relu', True),
    (8192, 1024, 1024, True, 1, 'none', False), (8192, 1024, 3072, True, 0, 'none', False),
    (8000, 1024, 1088, True, 2, 'relu', False), (4000, 2048, 1024, False, 1, 'none', True),
    (8192, 1024, 1000, True, 0, 'none', False),
    # round 5: a few thousand rows x d with a d-long contraction (the IWSLT recipe's 2048-row decoder projections and their dX)
    (2048, 1024, 1024, False, 0, 'none', True), (2048, 1024, 1024, True, 1, 'none', False), (1900, 1024, 1024, True, 0, 'none', False),
    (4096, 768, 1536, False, 0, 'relu', True),
    # round 5: 384 256-tiles = 1.5 rounds of the chip against 768 half-tiles = 3 (NLLB-1.3B's q|k|v projection at 8192 rows)
    (8192, 3072, 1024, False, 0, 'none', True)])
def test_half_m_tiles_for_outputs_that_fill_half_the_chip(M, N, K, b_col, mode, act, bias):
    """Outputs of 80..159 256-tiles (NLLB-1.3B's 8192 x 1024 projections at C5) run on gemm8p's 128 x 256 tile form (launch
    sampling: tag bit 0x400): forward with bias / ReLU, dX with the residual-branch gradient as aux, the ReLU-masked dH form,
    ragged M (a partial last tile), K with a partial last K-tile — against fp64.  Since round 5 also outputs of 48..159 such
    tiles at <= 4096 rows with a contraction of 1024..1536 (the 128-tile kernel took 19 us for 2048 x 1024 x 1024, this one 16)."""
    import ctypes
    from pasero_amd import functional as F, lib
    torch.manual_seed(M + N + K)
    dt = torch.float16 if K == 3072 else torch.bfloat16   # (one of the shapes in the reference's default 16-bit type)
    a = (torch.randn(M, K, device='cuda') * 0.5).to(dt)
    b = (torch.randn(K, N, device='cuda') * 0.05).to(dt) if b_col else (torch.randn(N, K, device='cuda') * 0.05).to(dt)
    bv = torch.randn(N, device='cuda').to(dt) if bias else None
    aux = torch.randn(M, N, device='cuda').to(dt) if mode else None
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(4, 1), 'start')
    out = F.gemm(a, b, b_col=b_col, bias=bv, aux=aux, act=act, mode=mode)
    n = L.pk_gemm_timing_stop()
    ints = [ctypes.c_int() for _ in range(5)]
    fl, ms = ctypes.c_double(), ctypes.c_float()
    lib.check(L.pk_gemm_timing_read(0, *[ctypes.byref(t) for t in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
    assert n == 1 and (ints[0].value & 0x40F) == 0x408, hex(ints[0].value)      # gemm8p, half-M tile
    ref = a.double() @ (b.double() if b_col else b.double().t())
    if mode == 2:
        ref = ref * (aux.double() > 0)
    else:
        if bv is not None:
            ref = ref + bv.double()
        if act == 'relu':
            ref = ref.clamp_min(0)
        if mode == 1:
            ref = ref + aux.double()
    err = (out.double() - ref).abs().max().item()
    assert err <= 1e-2 * ref.abs().max().item(), err
