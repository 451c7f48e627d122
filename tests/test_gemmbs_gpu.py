"""The B-stationary GEMM (csrc/gemmbs.hip: K = 512, row-form A, thousands of rows, bias / ReLU epilogue) behind pk_gemm:
random problems around its tile and step boundaries against an fp64 product AND bitwise against the tiled kernels (the
MFMA shape, the k order and the epilogue arithmetic are the same, so every bit must agree); padded leading dimensions
with NaN in the padding, rows / columns that are not multiples of the step or the strip, an output that is a column
slice of a wider tensor, and the untouched bytes around the output."""
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
    buf = torch.full((rows, cols + pad), float('nan'), dtype=dtype)
    buf[:, :cols] = torch.from_numpy(rs.standard_normal((rows, cols)).astype(np.float32)).to(dtype)
    return buf.cuda()[:, :cols], buf[:, :cols].double()


def _timed_kernels(L, fn):
    """kernel tags of the GEMM launches `fn` makes (the library's own launch sampling, stride 1)"""
    import ctypes
    from pasero_amd import lib
    lib.check(L.pk_gemm_timing_start(16, 1), 'start')
    fn()
    n = L.pk_gemm_timing_stop()
    tags = []
    for i in range(n):
        ints = [ctypes.c_int() for _ in range(5)]
        fl, ms = ctypes.c_double(), ctypes.c_float()
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(x) for x in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
        tags.append(ints[0].value)
    return tags


@pytest.mark.parametrize('seed', range(4))
def test_gemmbs_fuzz_against_fp64_and_the_tiled_kernel(F, seed):
    from pasero_amd import lib
    L = lib.load()
    rs = np.random.RandomState(7000 + seed)
    rows = [8192, 12296, 16384, 24000, 30008, 32768]
    cols = [512, 520, 1024, 1032, 1536, 2048]

    def takes(M, N):  # pk_gemmbs_eligible's size rule: every workgroup walks at least eight 32-row steps
        nt_n, steps = -(-N // 256), -(-M // 32)
        g = min(steps, max(1, 256 // nt_n))
        return -(-steps // g) >= 8

    taken = 0
    for case in range(12):
        dtype = [torch.bfloat16, torch.float16][rs.randint(2)]
        M, N, K = int(rs.choice(rows)), int(rs.choice(cols)), 512
        b_col = bool(rs.randint(2))
        pad_a, pad_b, pad_c = int(rs.choice([0, 8, 64])), int(rs.choice([0, 8])), int(rs.choice([0, 8, 24]))
        a, a64 = _operand(rs, M, K, pad_a, dtype)
        b, b64 = _operand(rs, *((K, N) if b_col else (N, K)), pad_b, dtype)
        bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(dtype) if rs.randint(2) else None
        act = ['none', 'relu', 'mask'][rs.randint(3)]
        alpha = float(rs.choice([1.0, 1.0, 0.5]))
        ref = a64 @ (b64 if b_col else b64.t())
        ref = ref * alpha
        aux = None
        if act == 'mask':  # mode 2: v * relu'(aux), no bias (the dH GEMM of a ReLU feed-forward)
            bias = None
            pad_x = int(rs.choice([0, 8]))
            aux, aux64 = _operand(rs, M, N, pad_x, dtype)
            ref = ref * (aux64 > 0)
        if bias is not None:
            ref = ref + bias.double()
        if act == 'relu':
            ref = ref.clamp(min=0)
        outs = {}
        for mode in (1, 0):
            L.pk_gemm_use_bs(mode)
            buf = torch.full((M + 1, N + pad_c), 7.0, dtype=dtype, device='cuda')
            out = buf[:M, :N]
            if act == 'mask':
                tags = _timed_kernels(L, lambda: F.gemm(a, b, b_col=b_col, aux=aux, act='relu', mode=2, alpha=alpha, out=out))
            else:
                tags = _timed_kernels(L, lambda: F.gemm(a, b, b_col=b_col, bias=None if bias is None else bias.cuda(),
                                                        act=act, alpha=alpha, out=out))
            outs[mode] = (buf, tags)
        L.pk_gemm_use_bs(1)
        what = (seed, case, str(dtype), M, N, b_col, pad_a, pad_b, pad_c, act, alpha, bias is not None)
        (buf1, tags1), (buf0, tags0) = outs[1], outs[0]
        assert any(t & 0x200 for t in tags1) == takes(M, N) and not any(t & 0x200 for t in tags0), (what, tags1, tags0)
        taken += takes(M, N)
        got = buf1[:M, :N]
        assert torch.isfinite(got.float()).all(), what
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= 8e-3 * np.sqrt(K) * 4, (what, err)
        assert torch.equal(buf1.view(torch.int16), buf0.view(torch.int16)), what  # output AND the bytes around it
        assert bool((buf1[M] == 7.0).all()) and (pad_c == 0 or bool((buf1[:, N:] == 7.0).all())), what
    assert taken >= 4, taken


def test_gemmbs_declines_what_it_does_not_take(F):
    """other contractions, few rows, col-form A, residual / general epilogues stay on the tiled kernels"""
    from pasero_amd import lib
    L = lib.load()
    L.pk_gemm_use_bs(1)
    a = torch.randn(16384, 512, device='cuda').bfloat16()
    w = torch.randn(1024, 512, device='cuda').bfloat16()
    aux = torch.randn(16384, 1024, device='cuda').bfloat16()
    assert any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a, w)))
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a, w, aux=aux, mode=1)))
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a, w, act='gelu')))  # (GELU only with its preact output)
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a, w, act='silu', preact=torch.empty_like(aux))))
    wide = torch.empty(16384, 1032, device='cuda').bfloat16()
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a, w, act='gelu', preact=wide[:, :1024])))  # pitch != C's
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a[:512], w)))
    a2 = torch.randn(16384, 1024, device='cuda').bfloat16()
    w2 = torch.randn(1024, 1024, device='cuda').bfloat16()
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(a2, w2)))
    at = torch.randn(512, 16384, device='cuda').bfloat16()
    assert not any(t & 0x200 for t in _timed_kernels(L, lambda: F.gemm(at, w.t().contiguous()[:, :1024], a_col=True, b_col=True)))


def test_gemmbs_is_bitwise_reproducible(F):
    a = torch.randn(32768, 512, device='cuda').bfloat16()
    w = torch.randn(2048, 512, device='cuda').bfloat16()
    bias = torch.randn(2048, device='cuda').bfloat16()
    first = F.gemm(a, w, bias=bias, act='relu')
    for _ in range(20):
        assert torch.equal(F.gemm(a, w, bias=bias, act='relu').view(torch.int16), first.view(torch.int16))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_relu_mask_as_bits(F, dtype):
    """pk_gemm_relu_bits: fc1 forward writes h and, next to it, one bit per element (h > 0); the dH GEMM reads the bits
    instead of h.  h must equal pk_gemm's ReLU output bit for bit, the bits must be numpy's packbits of (h > 0) in little
    bit order, and the masked product must equal pk_gemm mode 2 with h as the mask operand bit for bit — ragged rows too."""
    rs = np.random.RandomState(5)
    for M, N in ((32768, 2048), (12296 + 8, 1536)):
        K = 512
        x = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(dtype).cuda()
        w1 = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32)).to(dtype).cuda()
        b1 = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(dtype).cuda()
        w2 = torch.from_numpy(rs.standard_normal((K, N)).astype(np.float32)).to(dtype).cuda()
        dy = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(dtype).cuda()
        assert F.relu_bits_eligible(x, w1)
        h, bits = F.gemm_relu_bits(x, w1, b1)
        h_ref = F.gemm(x, w1, bias=b1, act='relu')
        assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16))
        want = np.packbits((h_ref.float() > 0).cpu().numpy(), axis=1, bitorder='little')
        assert np.array_equal(bits.cpu().numpy(), want)
        dh = F.gemm_mask_bits(dy, w2, bits)
        dh_ref = F.gemm(dy, w2, b_col=True, act='relu', aux=h_ref, mode=2)
        assert torch.equal(dh.view(torch.int16), dh_ref.view(torch.int16))
    assert not F.relu_bits_eligible(x[:512], w1)  # too few rows: the tiled kernels, the activations as the mask


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,N,K', [(8192, 8192, 1024), (4096, 4096, 1024), (8192, 1024, 2048), (8200, 4096, 1024), (32768, 4096, 1024)])
def test_relu_mask_as_bits_on_the_phase_interleaved_kernel(F, dtype, M, N, K):
    """round 5: the same mask bits for the d = 1024 feed-forward (NLLB-1.3B at C5: 8192 x 8192 x 1024; transformer_big at C3:
    32768 x 4096 x 1024), through gemm8p's lean epilogue on 256 x 256 and 128 x 256 tiles (`epilogue_pass_bits`): h bit for bit
    pk_gemm's ReLU output, the bits numpy's packbits of (h > 0), the masked dH bit for bit pk_gemm mode 2 with h as the mask
    operand; ragged rows; the launch sampling says which kernel ran (tag 0x1000)."""
    import ctypes
    from pasero_amd import lib
    rs = np.random.RandomState(M + N)
    x = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(dtype).cuda()
    w1 = (torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32)) * 0.05).to(dtype).cuda()
    b1 = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(dtype).cuda()
    w2 = (torch.from_numpy(rs.standard_normal((K, N)).astype(np.float32)) * 0.05).to(dtype).cuda()
    dy = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(dtype).cuda()
    assert F.relu_bits_eligible(x, w1)
    L = lib.load()
    lib.check(L.pk_gemm_timing_start(8, 1), 'start')
    h, bits = F.gemm_relu_bits(x, w1, b1)
    dh = F.gemm_mask_bits(dy, w2, bits)
    n = L.pk_gemm_timing_stop()
    tags = []
    for i in range(n):
        ints = [ctypes.c_int() for _ in range(5)]
        fl, ms = ctypes.c_double(), ctypes.c_float()
        lib.check(L.pk_gemm_timing_read(i, *[ctypes.byref(t) for t in ints], ctypes.byref(fl), ctypes.byref(ms)), 'read')
        tags.append(ints[0].value)
    assert n == 2 and all(t & 0x1000 for t in tags) and (tags[1] & 0x2000), [hex(t) for t in tags]
    h_ref = F.gemm(x, w1, bias=b1, act='relu')
    assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16))
    want = np.packbits((h_ref.float() > 0).cpu().numpy(), axis=1, bitorder='little')
    assert np.array_equal(bits.cpu().numpy(), want)
    dh_ref = F.gemm(dy, w2, b_col=True, act='relu', aux=h_ref, mode=2)
    assert torch.equal(dh.view(torch.int16), dh_ref.view(torch.int16))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_gelu_epilogues_between_the_mfmas(F, dtype):
    """the GELU feed-forward of the speech / BERT-style configurations (pasero/models/modules.py:220-228): fc1 forward
    writes gelu(pre) AND pre, the dH GEMM multiplies by gelu'(pre).  Both must equal the tiled kernel bit for bit (the
    same act_fwd_fast / act_bwd_fast on the same fp32 sums) and the fp64 erf formulas to storage precision — ragged
    rows, padded pitches, the bytes around the outputs."""
    from pasero_amd import lib
    L = lib.load()
    rs = np.random.RandomState(11)
    for M, N, pad in ((24000, 2048, 0), (12296, 1536, 8), (32768, 520, 24)):
        K = 512
        x, x64 = _operand(rs, M, K, 0, dtype)
        w1, w164 = _operand(rs, N, K, 8, dtype)
        w1, w164 = w1 * 0.05, w164 * 0.05
        w164 = w1.double().cpu()
        b1 = (torch.from_numpy(rs.standard_normal(N).astype(np.float32)) * 0.5).to(dtype)
        res = {}
        for mode in (1, 0):
            L.pk_gemm_use_bs(mode)
            hbuf = torch.full((M + 1, N + pad), 7.0, dtype=dtype, device='cuda')
            pbuf = torch.full((M + 1, N + pad), 5.0, dtype=dtype, device='cuda')
            tags = _timed_kernels(L, lambda: F.gemm(x, w1, bias=b1.cuda(), act='gelu', out=hbuf[:M, :N], preact=pbuf[:M, :N]))
            assert any(t & 0x200 for t in tags) == bool(mode), (M, N, mode, tags)
            res[mode] = (hbuf, pbuf)
        L.pk_gemm_use_bs(1)
        (h1, p1), (h0, p0) = res[1], res[0]
        assert torch.equal(h1.view(torch.int16), h0.view(torch.int16)) and torch.equal(p1.view(torch.int16), p0.view(torch.int16)), (M, N)
        assert bool((h1[M] == 7.0).all()) and bool((p1[M] == 5.0).all()), (M, N)
        pre64 = x64 @ w164.t() + b1.double()
        h64 = 0.5 * pre64 * (1 + torch.erf(pre64 / np.sqrt(2.0)))
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        assert (p1[:M, :N].double().cpu() - pre64).abs().max().item() <= eps * pre64.abs().max().item() + 1e-3
        assert (h1[:M, :N].double().cpu() - h64).abs().max().item() <= eps * h64.abs().max().item() + 1e-3
        # backward: dH = (dZ W2) * gelu'(pre), W2 [K, N] in col form
        dz, dz64 = _operand(rs, M, K, 8, dtype)
        w2, w264 = _operand(rs, K, N, 0, dtype)
        w2 = w2 * 0.05
        w264 = w2.double().cpu()
        pre = p1[:M, :N]
        out = {}
        for mode in (1, 0):
            L.pk_gemm_use_bs(mode)
            buf = torch.full((M + 1, N + pad), 3.0, dtype=dtype, device='cuda')
            tags = _timed_kernels(L, lambda: F.gemm(dz, w2, b_col=True, aux=pre, act='gelu', mode=2, out=buf[:M, :N]))
            assert any(t & 0x200 for t in tags) == bool(mode), (M, N, mode, tags)
            out[mode] = buf
        L.pk_gemm_use_bs(1)
        assert torch.equal(out[1].view(torch.int16), out[0].view(torch.int16)), (M, N)
        p64 = pre.double().cpu()
        dgelu = 0.5 * (1 + torch.erf(p64 / np.sqrt(2.0))) + p64 * torch.exp(-0.5 * p64 * p64) / np.sqrt(2 * np.pi)
        ref = (dz64 @ w264) * dgelu
        assert (out[1][:M, :N].double().cpu() - ref).abs().max().item() <= eps * ref.abs().max().item() + 2e-3, (M, N)
