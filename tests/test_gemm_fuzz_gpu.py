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
