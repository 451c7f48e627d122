/* pasero_hip.h — C ABI of libpasero_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the Transformer
 * encoder-decoder training hot path of naver/pasero (pasero/models/transformer.py, pasero/models/modules.py).
 *
 * The reference has NO native / FFI layer (it is pure Python on stock torch ops), so there is no
 * reference header to mirror: each entry point below replaces the stock-torch op sequence cited next to it, and is
 * what a binding of this path (ctypes in pasero_amd/lib.py; cgo / JNI / N-API equally) links against.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless stated; the caller allocates all outputs
 *     and workspaces; tensors are row-major and contiguous unless a leading dimension (ld*) is given.
 *   - `dtype`: PK_F32 or PK_BF16 selects the storage type of all T* tensors of the call (accumulation is always fp32).
 *   - `stream` is a hipStream_t; all work is enqueued on it, nothing synchronises, nothing allocates: every call is
 *     hipGraph-capturable.
 *   - returns 0 on success, -1 on an argument error, or a hipError_t; pk_last_error() gives the message
 *     (thread-local).
 *   - dropout masks are a pure function of (seed, offset, element index) [Philox4x32-10]: the backward call takes
 *     the same (seed, offset) as the forward call and regenerates the mask, nothing is stored.
 */
#ifndef PASERO_HIP_H
#define PASERO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PK_F32 = 0, PK_BF16 = 1 };
/* pasero/models/modules.py:220-228 get_activation_fn */
enum { PK_ACT_NONE = 0, PK_ACT_RELU = 1, PK_ACT_GELU = 2, PK_ACT_GELU_TANH = 3, PK_ACT_SILU = 4 };

int pk_version(void);
const char* pk_last_error(void);

/* ---- Linear layers (K2/K4/K5/K6): replaces nn.Linear fwd/bwd, pasero/models/modules.py:92-96, and the tied
 * output projection, modules.py:935-947.
 *   C[m,n] = epi( alpha * sum_k A(m,k) * B(n,k) )
 *   a_col = 0: A stored [M][lda] (k contiguous)      a_col = 1: A stored [K][lda] (m contiguous)
 *   b_col = 0: B stored [N][ldb] (k contiguous)      b_col = 1: B stored [K][ldb] (n contiguous)
 *   forward  y = x Wᵀ + b : A = x, B = W, (0,0)   dX = dY W : A = dY, B = W, (0,1)   dW = dYᵀ X : A = dY, B = X, (1,1)
 *   mode 0: C = act(v + bias)        (preact, if given, receives v + bias)
 *   mode 1: C = act(v + bias) + aux  (residual add, or gradient accumulation with aux == C)
 *   mode 2: C = v * act'(aux)        (backward through the activation; aux = pre-activation, or post- for ReLU)
 *   splitk > 1: K is cut into `splitk` slices reduced through `workspace` (>= splitk*M*N*4 bytes), deterministic. */
int pk_gemm(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact, long long M,
            long long N, long long K, long long lda, long long ldb, long long ldc, long long ldaux, long long ldpre,
            int a_col, int b_col, int act, int mode, float alpha, int dtype, int splitk, void* workspace,
            size_t ws_bytes, void* stream);

/* ---- Residual + dropout + LayerNorm (K4): replaces `residual + dropout(x)` followed by nn.LayerNorm,
 * pasero/models/transformer.py:1043-1054,1073-1086 (encoder), :1322-1339,1389-1407 (decoder), :941-947 (Norm).
 *   z = (residual ? residual : 0) + dropout(x)          -> z_out (optional)
 *   y = (z - mean) * rstd * gamma + beta                -> y_out, mean[rows], rstd[rows]   (only if gamma != NULL) */
int pk_residual_ln_fwd(const void* x, const void* residual, const void* gamma, const void* beta, void* z_out,
                       void* y_out, float* mean, float* rstd, long long rows, int d, float eps, float drop_p,
                       unsigned long long seed, unsigned long long offset, int dtype, void* stream);
/*   dz = LN_bwd(dy; z, gamma, mean, rstd) (if gamma) + dz_extra (if given)
 *   dres_out = dz (optional)      dx_out = dz * keep_mask / (1-p) (optional)      dgamma, dbeta (optional) */
size_t pk_residual_ln_bwd_workspace(long long rows, int d);
int pk_residual_ln_bwd(const void* dy, const void* dz_extra, const void* z, const void* gamma, const float* mean,
                       const float* rstd, void* dres_out, void* dx_out, void* dgamma, void* dbeta, void* workspace,
                       size_t ws_bytes, long long rows, int d, float drop_p, unsigned long long seed,
                       unsigned long long offset, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PASERO_HIP_H */
