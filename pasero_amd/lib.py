"""ctypes binding of libpasero_hip.so (the C ABI declared in include/pasero_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, this raises.  The product
path never routes through PyTorch eager ops or the CPU oracle for the work these kernels do.
"""
import ctypes
import os
from ctypes import c_int, c_float, c_longlong, c_size_t, c_ulonglong, c_void_p, c_char_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PASERO_HIP_LIB') or os.path.join(_HERE, 'libpasero_hip.so')  # env: diagnostic builds

PK_F32, PK_BF16, PK_F16 = 0, 1, 2
ACT = {'none': 0, None: 0, 'relu': 1, 'gelu': 2, 'gelu_tanh': 3, 'geglu': 3, 'swiglu': 4, 'silu': 4}

P, I, F, LL, SZ, ULL = c_void_p, c_int, c_float, c_longlong, c_size_t, c_ulonglong

# name -> (restype, argtypes); must stay in sync with include/pasero_hip.h (tests/test_boundary_cpu.py checks the symbols)
SIGNATURES = {
    'pk_version': (I, []),
    'pk_last_error': (c_char_p, []),
    'pk_gemm': (I, [P, P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, LL, I, I, I, I, F, I, I, P, SZ, P, P]),
    'pk_gemm_ex': (I, [P, P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, LL, I, I, I, I, F, I, I, P, SZ, P, I, P]),
    'pk_gemm_use_8p': (I, [I]),
    'pk_gemm_use_bs': (I, [I]),
    'pk_gemm_use_pw': (I, [I]),
    'pk_gemm_relu_bits_eligible': (I, [P, P, P, P, LL, LL, LL, LL, LL, LL, LL, I, I, I]),
    'pk_gemm_relu_bits': (I, [P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, I, I, F, I, P]),
    'pk_gemm_timing_start': (I, [I, I]),
    'pk_gemm_timing_stop': (I, []),
    'pk_gemm_timing_read': (I, [I, P, P, P, P, P, P, P]),
    'pk_gemm_timing_shape': (I, [I, P, P, P]),
    'pk_gemm_wgrad_group_eligible': (I, [P, I]),
    'pk_gemm_wgrad_group_workspace': (SZ, [P, I]),
    'pk_gemm_wgrad_group': (I, [P, I, I, P, SZ, P]),
    'pk_gemm_wgrad_group_map': (I, [P, I, P, I]),
    'pk_gemm_wgrad_pair': (I, [I]),
    'pk_gemm_ln_eligible': (I, [LL, LL, LL, LL, LL, I]),
    'pk_gemm_ln_fwd': (I, [P] * 10 + [LL] * 6 + [F, F, ULL, ULL, I, P]),
    'pk_decoder_step_scratch': (SZ, [P, I]),
    'pk_decoder_step': (I, [P, P, I, I, I, P, P, LL, P, P, I, P, SZ, P, LL, P]),
    'pk_argmax_rows': (I, [P, LL, LL, LL, P, LL, I, P]),
    'pk_pad_rows': (I, [P, I, P, P, I, I, LL, I, P]),
    'pk_residual_ln_fwd': (I, [P, P, P, P, P, P, P, P, LL, I, F, F, ULL, ULL, I, P]),
    'pk_residual_ln_bwd_workspace': (SZ, [LL, I]),
    'pk_residual_ln_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, SZ, LL, I, F, ULL, ULL, I, P]),
    'pk_residual_ln_bwd_partials': (I, [P, P, P, P, P, P, P, P, P, SZ, LL, I, F, ULL, ULL, I, P]),
    'pk_ln_param_grads': (I, [P, I, LL, I, I, P]),
    'pk_attn_fwd': (I, [P, P, P, P, P, P, I, I, I, I, I] + [LL] * 8 + [I, F, F, ULL, ULL, P, I, P]),
    'pk_attn_bwd': (I, [P] * 11 + [I, I, I, I, I] + [LL] * 16 + [I, F, F, P, I, P]),
    'pk_attn_fwd_rope': (I, [P, P, P, P, P, P, I, I, I, I, I] + [LL] * 8 + [I, F, F, ULL, ULL, P, P, P, I, I, I, I, P]),
    'pk_attn_bwd_rope': (I, [P] * 11 + [I, I, I, I, I] + [LL] * 16 + [I, F, F, P, P, P, I, I, I, I, P]),
    'pk_attn_probs': (I, [P, P, P, P, I, I, I, I, I, LL, LL, LL, LL, I, F, I, P]),
    'pk_embed_fwd': (I, [P, P, P, P, LL, I, I, LL, F, I, F, ULL, ULL, I, P]),
    'pk_embed_bwd_workspace': (SZ, [LL, LL, I]),
    'pk_embed_bwd': (I, [P, P, P, P, SZ, LL, I, LL, LL, F, F, ULL, ULL, I, P]),
    'pk_embed_bwd_acc': (I, [P, P, P, P, SZ, LL, I, LL, LL, F, F, ULL, ULL, I, P]),
    'pk_ce_rows': (I, [P, LL, P, P, LL, P, P, P, LL, LL, LL, F, I, P]),
    'pk_ce_finalize': (I, [P, P, P, LL, LL, P, P]),
    'pk_colsum_workspace': (SZ, [LL, LL]),
    'pk_colsum': (I, [P, LL, P, LL, LL, P, SZ, I, P]),
    'pk_dropout': (I, [P, P, LL, F, ULL, ULL, I, P]),
    'pk_scale': (I, [P, P, LL, P, F, I, P]),
    'pk_act_fwd': (I, [P, P, LL, I, I, P]),
    'pk_act_bwd': (I, [P, P, P, LL, I, I, P]),
    'pk_glu_fwd': (I, [P, P, LL, I, I, P]),
    'pk_glu_bwd': (I, [P, P, P, LL, I, I, P]),
    'pk_col2im1d': (I, [P, P, I, I, I, I, I, I, I, I, I, P]),
    'pk_gated_act_bwd': (I, [P, P, P, P, P, LL, I, I, P]),
    'pk_rope': (I, [P, P, LL, I, LL, I, I, P, P, I, I, I, I, I, P]),
    'pk_mt_chunk_size': (I, []),
    'pk_mt_sqnorm': (I, [P, I, P, P, I, F, P, P, I, P, I, P]),
    'pk_mt_adam': (I, [P, I, P, P, I, P, F, F, F, F, F, F, F, I, P, I, P]),
    'pk_mt_copy': (I, [P, I, P, P, I, I, P]),
    'pk_comm_open': (I, [c_char_p]),
    'pk_comm_unique_id': (I, [P, I]),
    'pk_comm_init': (I, [P, I, I]),
    'pk_comm_destroy': (I, []),
    'pk_comm_size': (I, []),
    'pk_comm_all_reduce_mean': (I, [P, LL, I, I, P, P]),
    'pk_comm_direct_plan': (I, [LL, I, I, P, P, P, P]),
    'pk_layer_fwd': (I, [P]),
    'pk_layer_fwd_ws': (SZ, [P]),
    'pk_layer_bwd_sizes': (I, [P, P, P]),
    'pk_layer_bwd': (I, [P]),
    'pk_logmel_workspace': (SZ, [I]),
    'pk_logmel': (I, [P, P, LL, P, P, SZ, I, P]),
}

PK_WGRAD_MAX = 8
PK_GEMM_PAD_N, PK_GEMM_PAD_K = 1, 2  # pk_gemm_ex promises (include/pasero_hip.h)


class PkWgradProblem(ctypes.Structure):
    """include/pasero_hip.h: one weight-gradient problem of a grouped launch"""
    _fields_ = [('A', P), ('B', P), ('C', P), ('asum_out', P),
                ('M', LL), ('N', LL), ('K', LL), ('lda', LL), ('ldb', LL), ('ldc', LL)]


class PkAttnBlock(ctypes.Structure):
    """include/pasero_hip.h: an attention sub-block of pk_layer_fwd / pk_layer_bwd"""
    _fields_ = [(n, P) for n in ('w_in', 'b_in', 'w_o', 'b_o', 'ln_g', 'ln_b', 'proj', 'kv', 'attn', 'z', 'y', 'ln_out', 'lse',
                                 'mean', 'rstd', 'dw_in', 'db_in', 'dw_o', 'db_o', 'dln_g', 'dln_b')] + [('drop_offset', ULL)]


class PkFfnBlock(ctypes.Structure):
    _fields_ = [(n, P) for n in ('w1', 'b1', 'w2', 'b2', 'ln_g', 'ln_b', 'h', 'pre', 'z', 'y', 'ln_out', 'bits', 'mean', 'rstd',
                                 'dw1', 'db1', 'dw2', 'db2', 'dln_g', 'dln_b')] + [('drop_offset', ULL)]


class PkLayer(ctypes.Structure):
    _fields_ = ([(n, I) for n in ('dtype', 'is_decoder', 'fused_tail', 'act', 'B', 'T', 'S', 'd', 'f', 'heads', 'prenorm')] +
                [('eps', F), ('drop_p', F), ('attn_scale', F), ('seed', ULL), ('x', P), ('enc', P), ('self_pad', P),
                 ('cross_pad', P), ('self_', PkAttnBlock), ('cross', PkAttnBlock), ('ffn', PkFfnBlock), ('dy', P), ('dx', P),
                 ('denc', P), ('denc_prev', P), ('dy_masked', P), ('dx_masked', P), ('dx_mask_offset', ULL), ('scratch', P),
                 ('ws', P), ('scratch_bytes', SZ), ('ws_bytes', SZ), ('stream', P)])


_lib = None


def load():
    """Load the library (once).  Raises ImportError with the build command if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, "csrc")}` '
            '(or `python -c "import __graft_entry__ as g; g.build()"`). pasero_amd has no fallback path.'
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pk_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed (code {rc}): {msg}')


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return PK_F32
    if t.dtype == torch.bfloat16:
        return PK_BF16
    if t.dtype == torch.float16:
        return PK_F16
    raise TypeError(f'pasero_amd kernels support float32, bfloat16 and float16, got {t.dtype}')


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  (Every kernel call asks: torch.cuda.current_stream()
    builds a Stream object behind three Python layers, ~10 us — 2 ms per C2 step; the raw handle is one C call.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('pasero_amd kernels need CUDA/HIP tensors (no CPU fallback); got a CPU tensor')


_workspaces = {}


def workspace(nbytes: int, device, tag: str = 'default') -> torch.Tensor:
    """Grow-only per-(device, stream, tag) scratch buffer; kernels on one stream use it in order, so sharing is safe
    (the weight-gradient GEMMs that run on a second stream get their own)."""
    key = (device.index if isinstance(device, torch.device) else str(device), tag,
           stream_ptr() if torch.cuda.is_available() else 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws
