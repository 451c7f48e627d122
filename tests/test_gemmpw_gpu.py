"""The persistent 128 x 256-tile GEMM (csrc/gemmpw.hip, round 6; include/pasero_hip.h: pk_gemm_use_pw) — the d = 1024 projections,
fc1 forward, masked dH GEMM and vocabulary logits (pasero/models/modules.py:92-96, transformer.py:999-1019) whose output is
many rounds of tiles behind a short contraction.  Through the C ABI (pk_gemm, pk_gemm_relu_bits) with the kernel switched on
and off in one process: same summation order as the tiled kernels, so every result must be BIT FOR BIT theirs (alpha = 1); and
against an fp64 product.  Shapes around every edge of the walk: tile counts that are no multiple of the workgroup count (the
workgroups run different numbers of tiles), ragged M (rows past M are read as zeros and never stored), N with a partial last
tile, padded vocabulary rows, K-tile counts 10..32 of every residue mod 3 (the ring rotates by K-tiles % 3 per tile), both
operand forms of B, bias / no bias, fp16."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def _pw(on):
    from pasero_amd import lib
    return lib.load().pk_gemm_use_pw(int(on))


def _took_pw(fn):
    """run fn() with the library's GEMM sampling on -> (result, kernel tags of the sampled launches)"""
    from pasero_amd import lib
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(64, 1), 'pk_gemm_timing_start')
    out = fn()
    torch.cuda.synchronize()
    n = L.pk_gemm_timing_stop()
    ints = [ctypes.c_int() for _ in range(5)]
    flops, ms = ctypes.c_double(), ctypes.c_float()
    tags = []
    for i in range(n):
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(flops), ctypes.byref(ms)), 'read')
        tags.append(ints[0].value)
    return out, tags


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(*shape, device='cuda', generator=g) * scale).to(dtype)


CASES = [
    # M, N, K, b_col, bias
    (8192, 3072, 1024, False, True),    # C5 q|k|v forward: 768 tiles = 3 per workgroup, nk = 16 (rotation 1)
    (8192, 3072, 1024, True, False),    # the same contraction with col-form B (a dX GEMM's operand form)
    (8192, 2048, 1024, False, True),    # C5 cross k|v forward: 512 tiles = 2 per workgroup
    (4096, 8192, 640, False, True),     # nk = 10 (the minimum; rotation 1), 1024 tiles
    (4096, 8192, 704, True, True),      # nk = 11 (rotation 2)
    (4096, 8192, 768, False, False),    # nk = 12 (rotation 0)
    (2048, 8192, 2048, False, True),    # nk = 32 (the maximum taken)
    (8200, 2560, 1024, False, True),    # ragged M (8200 = 64 tiles + 8 rows), 650 tiles: workgroups with 2 and with 3 tiles
    (8192, 2184, 1024, False, True),    # N = 8.53 tiles: a partial last column tile (2184 % 256 = 136: chunks past N are dead)
    (8192, 2184, 1024, True, True),
    (3000, 12000, 1024, False, False),  # both ragged; 24 x 47 = 1128 tiles
]


@pytest.mark.parametrize('M,N,K,b_col,bias', CASES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_lean_epilogue_bitwise_the_tiled_kernels(M, N, K, b_col, bias, dtype):
    from pasero_amd import functional as F
    if dtype == torch.float16 and (M, N, K) != (8192, 3072, 1024) and (M, N) != (8200, 2560):
        pytest.skip('fp16: two shapes are enough (the type only changes the MFMA and the conversions)')
    a = _rand((M, K), dtype, 1, 0.5)
    b = _rand((K, N) if b_col else (N, K), dtype, 2, 0.5)
    bv = _rand((N,), dtype, 3) if bias else None
    prev = _pw(1)
    try:
        got, tags = _took_pw(lambda: F.gemm(a, b, b_col=b_col, bias=bv))
        assert tags and all(t & 0x800 for t in tags), [hex(t) for t in tags]  # the persistent kernel took it
        _pw(0)
        ref, tags0 = _took_pw(lambda: F.gemm(a, b, b_col=b_col, bias=bv))
        assert not any(t & 0x4800 for t in tags0)
    finally:
        _pw(prev)
    assert torch.equal(got, ref)
    r64 = a.double() @ (b.double() if b_col else b.double().t())
    if bias:
        r64 += bv.double()
    tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -10
    assert (got.double() - r64).abs().max().item() <= tol * r64.abs().max().item()


def test_output_with_a_pitch_and_alpha():
    """C as a column block of a wider buffer (ldc > N); alpha != 1 (results equal to rounding: the multiply-add may contract
    differently from the tiled epilogue)"""
    from pasero_amd import functional as F
    M, N, K = 8192, 2048, 1024
    a, b = _rand((M, K), torch.bfloat16, 5, 0.5), _rand((N, K), torch.bfloat16, 6, 0.5)
    bv = _rand((N,), torch.bfloat16, 7)
    wide = torch.full((M, N + 512), 7.0, dtype=torch.bfloat16, device='cuda')
    prev = _pw(1)
    try:
        out, tags = _took_pw(lambda: F.gemm(a, b, bias=bv, out=wide[:, 256:256 + N], alpha=0.375))
        assert all(t & 0x800 for t in tags)
        _pw(0)
        ref = F.gemm(a, b, bias=bv, alpha=0.375)
    finally:
        _pw(prev)
    assert (wide[:, :256] == 7).all() and (wide[:, 256 + N:] == 7).all()  # nothing written beside the block
    assert (wide[:, 256:256 + N].float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()


def test_padded_vocabulary_rows():
    """pk_gemm_ex's PAD_N promise (a vocabulary that is no multiple of 8: NLLB's 256 206 — here 10 006): the rows of the
    output have room for N rounded up to 8, the last chunk is stored whole; the columns < N bit for bit the tiled kernel's"""
    from pasero_amd import functional as F
    M, V, K = 4096, 10006, 1024
    x, w = _rand((M, K), torch.bfloat16, 8, 0.5), _rand((V, K), torch.bfloat16, 9, 0.5)
    ldp = (V + 15) // 16 * 16
    bufs = [torch.zeros(M, ldp, dtype=torch.bfloat16, device='cuda') for _ in range(2)]
    prev = _pw(1)
    try:
        _, tags = _took_pw(lambda: F.gemm(x, w, out=bufs[0][:, :V], pad_n=True))
        assert all(t & 0x800 for t in tags), [hex(t) for t in tags]
        _pw(0)
        F.gemm(x, w, out=bufs[1][:, :V], pad_n=True)
    finally:
        _pw(prev)
    assert torch.equal(bufs[0][:, :V], bufs[1][:, :V])
    r64 = x.double() @ w.double().t()
    assert (bufs[0][:, :V].double() - r64).abs().max().item() <= 2 ** -8 * r64.abs().max().item()


@pytest.mark.parametrize('M,f,d', [(8192, 8192, 1024), (8200, 4096, 1024), (4096, 8192, 640)])
def test_relu_feed_forward_with_the_mask_as_bits(M, f, d):
    """pk_gemm_relu_bits on the persistent kernel: fc1 forward (h = relu(x W1^T + b1) + the mask bits) and the masked dH GEMM
    (dh = bit ? dZ W2 : 0, col-form W2) — h, bits and dh bit for bit the tiled kernels', the bits = (h > 0)"""
    from pasero_amd import functional as F
    x, w1, b1 = _rand((M, d), torch.bfloat16, 11, 0.5), _rand((f, d), torch.bfloat16, 12, 0.1), _rand((f,), torch.bfloat16, 13, 0.2)
    dz, w2 = _rand((M, d), torch.bfloat16, 14, 0.5), _rand((d, f), torch.bfloat16, 15, 0.1)
    assert F.relu_bits_eligible(x, w1)
    prev = _pw(1)
    try:
        (h, bits), t1 = _took_pw(lambda: F.gemm_relu_bits(x, w1, b1))
        dh, t2 = _took_pw(lambda: F.gemm_mask_bits(dz, w2, bits))
        assert all(t & 0x800 for t in t1 + t2), [hex(t) for t in t1 + t2]
        _pw(0)
        h0, bits0 = F.gemm_relu_bits(x, w1, b1)
        dh0 = F.gemm_mask_bits(dz, w2, bits0)
    finally:
        _pw(prev)
    assert torch.equal(h, h0) and torch.equal(bits, bits0) and torch.equal(dh, dh0)
    want = (h > 0).view(M, f // 8, 8).to(torch.uint8)
    packed = (want << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, packed)
    r64 = (dz.double() @ w2.double()) * (h > 0)
    assert (dh.double() - r64).abs().max().item() <= 2 ** -8 * r64.abs().max().item()


def test_repeated_launches_are_reproducible_under_a_competing_stream():
    """race screen of the hand-counted waits (LDS-DMA ring continued across tiles, side loads and stores in the same in-order
    queue): 60 launches over two data sets in turn beside a copy stream, every one bit for bit the first"""
    from pasero_amd import functional as F
    sets = [(_rand((8192, 1024), torch.bfloat16, 20 + k, 0.5), _rand((3072, 1024), torch.bfloat16, 30 + k, 0.5),
             _rand((3072,), torch.bfloat16, 40 + k)) for k in range(2)]
    prev = _pw(1)
    try:
        first = [F.gemm(a, b, bias=bv).clone() for a, b, bv in sets]
        side = torch.cuda.Stream()
        src = torch.randn(32 << 20, device='cuda')
        dst = torch.empty_like(src)
        bad = torch.zeros((), dtype=torch.int64, device='cuda')
        for it in range(60):
            if it % 3 != 2:
                with torch.cuda.stream(side):
                    dst.copy_(src)
            a, b, bv = sets[it & 1]
            bad += (F.gemm(a, b, bias=bv) != first[it & 1]).any()
        torch.cuda.synchronize()
        assert int(bad) == 0
    finally:
        _pw(prev)


def test_shapes_it_does_not_take_stay_on_the_tiled_kernels():
    from pasero_amd import functional as F
    prev = _pw(1)
    try:
        for M, N, K, kw in [(8192, 1024, 1024, {}),                      # one round of tiles
                            (8192, 3072, 512, {}),                       # eight K-tiles: too few for a tile to leave in
                            (8192, 3072, 4096, {}),                      # long contraction: the tiled kernels' ground
                            (8192, 3072, 1024, {'act': 'relu'})]:        # ReLU without the mask bits
            a, b = _rand((M, K), torch.bfloat16, 50, 0.5), _rand((N, K), torch.bfloat16, 51, 0.5)
            _, tags = _took_pw(lambda: F.gemm(a, b, **kw))
            assert tags and not any(t & 0x800 for t in tags), (M, N, K, [hex(t) for t in tags])
    finally:
        _pw(prev)


# ---- the persistent walk of 256 x 256 tiles (gemm8p_tile's PW form, gemm8p_pt_kernel): bit 1 of pk_gemm_use_pw, ON by default ----
PT_CASES = [
    # M, N, K, b_col, bias, act
    (8192, 8192, 1024, False, True, 'none'),     # C5 fc1-sized: 1024 tiles = 4 per workgroup
    (8192, 8192, 1024, True, False, 'none'),     # col-form B (a dX GEMM's operand form)
    (8192, 8192, 256, False, True, 'relu'),      # four K-tiles (the minimum), ReLU without the mask bits
    (16000, 8192, 1024, False, True, 'none'),    # ragged M (62.5 tile rows): the IWSLT recipe's encoder rows; 2016 tiles
    (6000, 24000, 384, True, True, 'none'),      # ragged M and N (93.75 tile columns), six K-tiles, col-form B; 2256 tiles
    (32768, 4096, 4096, False, False, 'none'),   # a long contraction (64 K-tiles)
    (8200, 16384, 128 * 5, False, True, 'none'),  # ten K-tiles; workgroups with 8 and with 9 tiles
]


@pytest.mark.parametrize('M,N,K,b_col,bias,act', PT_CASES)
def test_persistent_walk_bitwise_the_one_tile_kernel(M, N, K, b_col, bias, act):
    """Same tile, same K loop, same epilogue code: every output bit for bit the one-tile-per-workgroup kernel's; the walk only
    changes which workgroup computes a tile and when its operands are requested (the next tile's K-tile 0 behind this tile's
    last K-tile; per-lane offsets of tile (0, 0) + an SGPR offset; rows past M / N as zeros through the range check)."""
    from pasero_amd import functional as F
    a = _rand((M, K), torch.bfloat16, 1, 0.5)
    b = _rand((K, N) if b_col else (N, K), torch.bfloat16, 2, 0.5)
    bv = _rand((N,), torch.bfloat16, 3) if bias else None
    prev = _pw(2)
    try:
        got, tags = _took_pw(lambda: F.gemm(a, b, b_col=b_col, bias=bv, act=act))
        assert tags and all(t & 0x4000 for t in tags), [hex(t) for t in tags]
        _pw(0)
        ref, tags0 = _took_pw(lambda: F.gemm(a, b, b_col=b_col, bias=bv, act=act))
        assert not any(t & 0x4800 for t in tags0)
    finally:
        _pw(prev)
    assert torch.equal(got, ref)
    if M * N * K <= 8192 * 8192 * 1024:  # (the fp64 product of the largest cases takes seconds: the bitwise check carries them)
        r64 = a.double() @ (b.double() if b_col else b.double().t())
        if bias:
            r64 += bv.double()
        if act == 'relu':
            r64.clamp_(min=0)
        assert (got.double() - r64).abs().max().item() <= 2 ** -8 * r64.abs().max().item()


def test_persistent_walk_mask_bits_padded_rows_and_what_it_leaves_alone():
    from pasero_amd import functional as F
    prev = _pw(2)
    try:
        # the ReLU feed-forward with the mask as bits, both directions (pk_gemm_relu_bits)
        M, f, d = 8192, 8192, 1024
        x, w1, b1 = _rand((M, d), torch.bfloat16, 11, 0.5), _rand((f, d), torch.bfloat16, 12, 0.1), _rand((f,), torch.bfloat16, 13, 0.2)
        dz, w2 = _rand((M, d), torch.bfloat16, 14, 0.5), _rand((d, f), torch.bfloat16, 15, 0.1)
        (h, bits), t1 = _took_pw(lambda: F.gemm_relu_bits(x, w1, b1))
        dh, t2 = _took_pw(lambda: F.gemm_mask_bits(dz, w2, bits))
        assert all(t & 0x4000 for t in t1 + t2), [hex(t) for t in t1 + t2]
        # a padded vocabulary (pk_gemm_ex PAD_N): 2048 x 256 206, the C5 logits chunk
        V = 256206
        xl, wl = _rand((2048, 1024), torch.bfloat16, 16, 0.5), _rand((V, 1024), torch.bfloat16, 17, 0.05)
        ldp = (V + 15) // 16 * 16
        lg = [torch.zeros(2048, ldp, dtype=torch.bfloat16, device='cuda') for _ in range(2)]
        _, t3 = _took_pw(lambda: F.gemm(xl, wl, out=lg[0][:, :V], pad_n=True))
        assert all(t & 0x4000 for t in t3), [hex(t) for t in t3]
        # not its shapes: fewer than two rounds of tiles, an odd number of K-tiles, an aux operand
        for Ms, Ns, Ks, kw in [(8192, 3072, 1024, {}), (8192, 8192, 192, {}),
                               (8192, 8192, 1024, {'aux': _rand((8192, 8192), torch.bfloat16, 18), 'mode': 1})]:
            a, b = _rand((Ms, Ks), torch.bfloat16, 19, 0.5), _rand((Ns, Ks), torch.bfloat16, 20, 0.5)
            _, tg = _took_pw(lambda: F.gemm(a, b, **kw))
            assert tg and not any(t & 0x4000 for t in tg), (Ms, Ns, Ks, [hex(t) for t in tg])
        _pw(0)
        h0, bits0 = F.gemm_relu_bits(x, w1, b1)
        dh0 = F.gemm_mask_bits(dz, w2, bits0)
        F.gemm(xl, wl, out=lg[1][:, :V], pad_n=True)
    finally:
        _pw(prev)
    assert torch.equal(h, h0) and torch.equal(bits, bits0) and torch.equal(dh, dh0)
    assert torch.equal(lg[0][:, :V], lg[1][:, :V])


def test_persistent_walk_race_screen():
    """the next tile's K-tile 0 lands in stage 0 while the epilogue stages through stage 1, and a later tile's first two waits
    are relaxed past the epilogue's stores: 60 launches over two data sets beside a copy stream, every one bit for bit the first"""
    from pasero_amd import functional as F
    sets = [(_rand((8192, 1024), torch.bfloat16, 60 + k, 0.5), _rand((8192, 1024), torch.bfloat16, 70 + k, 0.5),
             _rand((8192,), torch.bfloat16, 80 + k)) for k in range(2)]
    prev = _pw(2)
    try:
        first = [F.gemm(a, b, bias=bv).clone() for a, b, bv in sets]
        side = torch.cuda.Stream()
        src = torch.randn(32 << 20, device='cuda')
        dst = torch.empty_like(src)
        bad = torch.zeros((), dtype=torch.int64, device='cuda')
        for it in range(60):
            if it % 3 != 2:
                with torch.cuda.stream(side):
                    dst.copy_(src)
            a, b, bv = sets[it & 1]
            bad += (F.gemm(a, b, bias=bv) != first[it & 1]).any()
        torch.cuda.synchronize()
        assert int(bad) == 0
    finally:
        _pw(prev)
