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
 *   - `dtype`: PK_F32, PK_BF16 or PK_F16 (the reference's default, config.py:518-523) selects the storage type of all
 *     T* tensors of the call; accumulation is always fp32.
 *   - `stream` is a hipStream_t; all work is enqueued on it, nothing synchronises, nothing allocates: every call is
 *     hipGraph-capturable.
 *   - returns 0 on success, -1 on an argument error, or a hipError_t; pk_last_error() gives the message
 *     (thread-local).
 *   - dropout masks are a pure function of (seed, offset, element index) [Philox4x32-10]: the backward call takes
 *     the same (seed, offset) as the forward call and regenerates the mask, nothing is stored.  Element i is kept iff
 *     the 16-bit draw number (i & 7) of philox(seed, offset, i >> 3) is >= floor(p * 65536): one evaluation serves the
 *     eight elements of a 16-byte chunk (p is quantised to 1/65536; the kept values are scaled by 1 / (1 - p)).
 */
#ifndef PASERO_HIP_H
#define PASERO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PK_F32 = 0, PK_BF16 = 1, PK_F16 = 2 };
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
 *   mode 3: C = act(v + bias) * aux  (gated FFN: SwiGLU / GEGLU, pasero/models/transformer.py:1013-1016)
 *   splitk > 1: K is cut into `splitk` slices reduced through `workspace` (>= splitk*M*(N+1)*4 bytes), deterministic.
 *   aux may alias C (accumulate in place: every element of aux is read by the thread that stores it).
 *   asum_out (optional, needs a_col = 1): asum_out[m] = sum_k A(m,k) — the bias gradient db = colsum(dY) comes out of
 *   the weight-gradient GEMM that already streams dY, instead of a separate pass over it. */
int pk_gemm(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact, long long M,
            long long N, long long K, long long lda, long long ldb, long long ldc, long long ldaux, long long ldpre,
            int a_col, int b_col, int act, int mode, float alpha, int dtype, int splitk, void* workspace,
            size_t ws_bytes, void* asum_out, void* stream);
/* pk_gemm with promises about padding (a vocabulary that is no multiple of 8 — NLLB's 256 206 — inside buffers whose rows
 * are padded to one; without them such a GEMM only fits the 128-tile kernel: 715-770 instead of 1 150-1 300 TFLOP/s at C5):
 *   PK_GEMM_PAD_N: C's rows have room for N rounded up to 8 (ldc >= that): whole 16-byte chunks may be stored, the pad
 *     columns receive unspecified finite values.  Mode 0 without bias / activation / preact, no split-K.
 *   PK_GEMM_PAD_K: row-form A, col-form B: A's rows extend to K rounded up to 8 (lda >= that) with ZEROS in the pad columns
 *     (pk_ce_rows leaves them so in its gradient); B has exactly K rows.
 * A promise that does not apply to the call (other layouts, fp32, N or K already a multiple of 8) is ignored. */
#define PK_GEMM_PAD_N 1
#define PK_GEMM_PAD_K 2
int pk_gemm_ex(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact, long long M,
               long long N, long long K, long long lda, long long ldb, long long ldc, long long ldaux, long long ldpre,
               int a_col, int b_col, int act, int mode, float alpha, int dtype, int splitk, void* workspace,
               size_t ws_bytes, void* asum_out, int pad_flags, void* stream);
/* Launch timing for the roofline measurement (no reference counterpart; the analogue of wrapping the reference's
 * nn.Linear calls in torch.cuda.Event pairs).  After pk_gemm_timing_start(max_samples, stride) every `stride`-th pk_gemm
 * call records a HIP event pair around exactly its main GEMM kernel (not the split-K reduce), on the launching stream.
 * pk_gemm_timing_stop() ends sampling and returns the number of samples; pk_gemm_timing_read(i, ...) synchronises on
 * sample i and returns the kernel that ran (128 = gemm_kernel 128x128 tiles, 256 = gemm256_kernel, 8 | flags = a
 * gemm8p instantiation: 0x10 general epilogue, 0x20 partial last K-tile, 0x400 the 128 x 256 tile, 0x40 the grouped weight-gradient launch of
 * pk_gemm_wgrad_group [flops = the sum over the group, splitk = the number of problems], 0x80 pk_gemm_ln_fwd), its
 * operand layouts, split-K factor, dtype, 2*M*N*K and the elapsed milliseconds. */
int pk_gemm_timing_start(int max_samples, int stride);
/* diagnostic: 1 / 0 routes the 256-tile GEMMs to the phase-interleaved kernel (gemm8p.hip, default) / to gemm256.hip;
 * 2 additionally sends every eligible GEMM with M, N >= 256 there, whether or not its tiles fill the chip (tests);
 * a negative argument only queries.  Returns the previous setting (env PK_GEMM_8P sets the initial one). */
int pk_gemm_use_8p(int on);
/* diagnostic: 1 / 0 lets pk_gemm send short-contraction GEMMs (K = 512, row-form A, thousands of rows; mode 0 with bias and
 * no activation / ReLU / GELU-with-preact, mode 2 with ReLU' or GELU') to the B-stationary kernel (gemmbs.hip, default) /
 * keeps them on the tiled kernels; negative: query only.  Returns the previous setting (env PK_GEMM_BS sets the initial
 * one).  Sample tag in pk_gemm_timing_read: 0x200 | number of K-tiles | activation << 4 | 0x40 the act'-mask epilogue
 * (mode 2) | 0x80 the mask as bits | 0x100 the pre-activation as second output. */
int pk_gemm_use_bs(int on);
/* diagnostic: a MASK of two persistent GEMM kernels for row-form A (round 6; negative: query only; returns the previous mask;
 * env PK_GEMM_PW sets the initial one, default 2).  Both keep the summation order of the tiled kernels: results bit for bit equal.
 *   bit 1 (ON by default): the persistent WALK of gemm8p.hip's 256 x 256 tiles (gemm8p_pt_kernel) — one workgroup per CU runs
 *     its tiles one after the other and requests the next tile's first K-tile behind the current tile's last, so a tile's
 *     epilogue runs with its successor's operands on their way; an even number >= 4 of whole K-tiles, at least two rounds of
 *     tiles over the chip, mode 0 (bias, none / ReLU) or the mask-bit epilogues of pk_gemm_relu_bits.  Worth 0.4 % of the C3 /
 *     C5 steps (the vocabulary logits + 4 %): the one-tile launch already overlaps a tile's store drain with the next
 *     workgroup's prologue.  Sample tag in pk_gemm_timing_read: 8 | 0x4000 (| 0x1000 / 0x2000 the mask bits written / read).
 *   bit 0 (off): the persistent 128 x 256-tile kernel with a second accumulator set (gemmpw.hip: a finished tile leaves from
 *     registers during the next tile's first K-tiles: no prologue, no epilogue phase; 10..32 whole K-tiles) — measured 20-30 %
 *     SLOWER than the tiled kernels, kept as an experiment (docs/experiments.md).  Sample tag 8 | 0x800. */
int pk_gemm_use_pw(int on);
/* The ReLU feed-forward's mask as ONE BIT per element (fc1 forward / fc2 dX of pasero/models/transformer.py:999-1019 at
 * K = 512): `bits` [M][ldbits] bytes, bit (n & 7) of byte n >> 3 of row m = (C[m][n] > 0) of the forward call.
 *   mode 0:  C = relu(alpha * A B^T + bias), bits written next to it          (fc1 forward)
 *   mode 2:  C = bit ? alpha * A B : 0                                        (dH = (dZ W2) * relu'(h): reads 1/16 of what
 *            the activations as the mask operand of pk_gemm mode 2 would — that GEMM is memory-bound at C2)
 * Only shapes of the B-stationary kernel (pk_gemm_relu_bits_eligible: row-form A, K = 512, N >= 256, thousands of rows, 16-bit);
 * for everything else pk_gemm with act = PK_ACT_RELU and, in backward, aux = the activations. */
int pk_gemm_relu_bits_eligible(const void* A, const void* B, const void* C, const void* bias, long long M, long long N,
                               long long K, long long lda, long long ldb, long long ldc, long long ldbits, int b_col,
                               int mode, int dtype);
int pk_gemm_relu_bits(const void* A, const void* B, void* C, const void* bias, unsigned char* bits, long long M, long long N,
                      long long K, long long lda, long long ldb, long long ldc, long long ldbits, int b_col, int mode,
                      float alpha, int dtype, void* stream);
int pk_gemm_timing_stop(void);
int pk_gemm_timing_read(int i, int* kernel, int* a_col, int* b_col, int* splitk, int* dtype, double* flops, float* ms);
/* the problem size of sample i (M, N, K as pk_gemm was called; 1, 1, 1 for a grouped launch, whose flops are a sum).  Kernel
 * tag 64 = the few-rows kernel (gemm_skinny.hip).  No reference counterpart (diagnostics: tools/gemm_in_model.py). */
int pk_gemm_timing_shape(int i, long long* M, long long* N, long long* K);

/* ---- The weight gradients of one layer in ONE launch: replaces the per-nn.Linear `grad_weight = grad_output^T @ input`
 * (+ `grad_bias = grad_output.sum(0)`) that autograd issues one by one during the backward of a layer,
 * pasero/models/transformer.py:1056-1099 (encoder layer), :1341-1417 (decoder layer), modules.py:92-96.
 *   problem p:  C_p[M,N] = A_p^T B_p   with A_p = dY [K][lda] and B_p = X [K][ldb] (both as pk_gemm's col form),
 *               asum_out_p[m] = sum_k A_p(k,m)  (optional: the bias gradient), C_p in the operands' 16-bit type.
 * Each of those GEMMs has a small output (4..16 tiles of 256 x 256 at d = 512) and a contraction over all B*T rows: alone
 * it needs 16..64 K-slices to fill the chip, i.e. that many fp32 partial outputs and a reduction launch per GEMM.  Up to
 * PK_WGRAD_MAX problems launched together share the 256 CUs with 4-5 slices each: one GEMM launch (every workgroup a
 * K-chunk of the same length) + one reduction launch (fixed slab order: deterministic).
 *   pk_gemm_wgrad_group_eligible: 1 if a problem can ride (16-bit dtype, M, N >= 256, 16-byte addressable operands and
 *     output, operands below 2 GiB); anything else goes through pk_gemm.  pk_gemm_wgrad_group refuses ineligible problems.
 *   pk_gemm_wgrad_group_workspace: bytes of fp32 scratch the group needs (0 if no problem is split). */
#define PK_WGRAD_MAX 8
typedef struct PkWgradProblem {
    const void* A;
    const void* B;
    void* C;
    void* asum_out; /* [M] or NULL */
    long long M, N, K, lda, ldb, ldc;
} PkWgradProblem;
int pk_gemm_wgrad_group_eligible(const PkWgradProblem* problem, int dtype);
size_t pk_gemm_wgrad_group_workspace(const PkWgradProblem* problems, int n);
int pk_gemm_wgrad_group(const PkWgradProblem* problems, int n, int dtype, void* workspace, size_t ws_bytes, void* stream);
/*   pk_gemm_wgrad_group_map (host only, no GPU work): which workgroup of the launch works on what.  The eight XCDs of the
 *     chip have an L2 each and the hardware deals workgroups to them round-robin (workgroup b runs on XCD b % 8), so the
 *     tiles of one (problem, K-slab) unit — which re-read the same rows of dY and X — are kept on ONE XCD wherever they fit.
 *     out[2 b] = problem index, out[2 b + 1] = position in that problem's slab-major (K-slab, tile) walk, both -1 for a
 *     workgroup that exits at once; at most `cap` workgroups are written.  Returns the grid size, -1 on bad arguments. */
int pk_gemm_wgrad_group_map(const PkWgradProblem* problems, int n, int* out, int cap);
/*   Round 5: a problem the plan cuts into exactly TWO K-slabs is finished inside the GEMM launch — of the two workgroups of a
 *     tile the second to arrive adds the first one's fp32 partial (published write-through behind an agent-scope flag) to its
 *     own and stores the 16-bit tile; fp32 addition commutes, so the result is bit for bit the reduction launch's whatever the
 *     arrival order.  A group of only such problems (NLLB-1.3B's layers at 8192 rows) has no reduction launch at all.
 *     Round 6: the hand-off fails LOUDLY and is stream-safe.  The ticket / flag words are per (device, stream) — grouped
 *     launches running at once on two streams never see each other's tickets; a stream beyond the library's 32 buffers takes
 *     the reduction launch.  A second workgroup whose bounded wait ends without its partner's flag stores NaN for its tile
 *     and raises a sticky error word in host-visible memory: the NEXT pk_gemm_wgrad_group call on any stream returns an error
 *     (pk_last_error names the tile) after re-zeroing every ticket buffer — a wrong weight gradient is never silent.
 *     pk_gemm_wgrad_pair: diagnostic switch, 1 / 0 = on (default; env PK_WGRAD_PAIR sets the initial value) / every split
 *     problem through the reduction launch; 2 = on AND the first workgroup's publish dropped, the second's wait shortened
 *     (tests of the error path only); negative: query only.  Returns the previous setting. */
int pk_gemm_wgrad_pair(int on);

/* ---- Linear + residual + dropout + LayerNorm in one kernel (K4 fused into K2/K5): replaces the tail of a post-norm
 * sub-block, `x = self.out_proj(x)` / `x = self.fc2(x)` followed by `x = residual + dropout(x); x = LayerNorm(x)`,
 * pasero/models/modules.py:739, pasero/models/transformer.py:1018,1043-1048,1076-1086 and :1322-1339,1389-1407.
 *   v = A Wᵀ + bias;  z = (residual ? residual : 0) + dropout(v) -> z_out (optional);  y = LN(z) * gamma + beta -> y_out
 *   A [M][lda], W [N][ldb] (nn.Linear layout), residual [M][ldr], z_out / y_out [M][N] contiguous, mean / rstd [M] fp32
 *   (mean == NULL: RMSNorm, beta must be NULL).  Dropout mask and statistics exactly as pk_residual_ln_fwd takes them
 *   (same Philox function of (seed, offset, element): pk_residual_ln_bwd regenerates the mask), so the stand-alone
 *   backward serves both.  LayerNorm needs whole rows in one tile: N must be 512 (pk_gemm_ln_eligible; K % 64 == 0,
 *   16-bit operands, 16-byte addressable); other shapes take pk_gemm + pk_residual_ln_fwd. */
int pk_gemm_ln_eligible(long long M, long long N, long long K, long long lda, long long ldb, int dtype);
int pk_gemm_ln_fwd(const void* A, const void* W, const void* bias, const void* residual, const void* gamma,
                   const void* beta, void* z_out, void* y_out, float* mean, float* rstd, long long M, long long N,
                   long long K, long long lda, long long ldb, long long ldr, float eps, float drop_p,
                   unsigned long long seed, unsigned long long offset, int dtype, void* stream);

/* ---- Residual + dropout + LayerNorm (K4): replaces `residual + dropout(x)` followed by nn.LayerNorm,
 * pasero/models/transformer.py:1043-1054,1073-1086 (encoder), :1322-1339,1389-1407 (decoder), :941-947 (Norm).
 *   z = (residual ? residual : 0) + dropout(x)          -> z_out (optional)
 *   y = (z - mean) * rstd * gamma + beta                -> y_out, mean[rows], rstd[rows]   (only if gamma != NULL)
 *   mean == NULL selects RMSNorm (pasero/models/modules.py:192-202): y = z * rsqrt(mean(z^2) + eps) * gamma in fp32,
 *   beta must be NULL; pk_residual_ln_bwd with mean == NULL is its backward (dbeta must be NULL). */
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
/*   The same with the parameter gradients DEFERRED: pk_residual_ln_bwd_partials leaves the per-workgroup partial sums of
 *   dgamma / dbeta in `workspace` (pk_residual_ln_bwd_workspace bytes, one workspace per LayerNorm), pk_ln_param_grads
 *   finishes up to PK_LN_GROUP_MAX LayerNorms of the same (rows, d) in ONE launch — a layer's two or three LayerNorms
 *   (pk_layer_bwd); each result equals pk_residual_ln_bwd's bit for bit (the same reduction in the same order). */
#define PK_LN_GROUP_MAX 4
typedef struct PkLnParamGrad {
    const void* workspace; /* written by pk_residual_ln_bwd_partials */
    void* dgamma;          /* [d] or NULL */
    void* dbeta;           /* [d] or NULL */
} PkLnParamGrad;
int pk_residual_ln_bwd_partials(const void* dy, const void* dz_extra, const void* z, const void* gamma, const float* mean,
                                const float* rstd, void* dres_out, void* dx_out, void* workspace, size_t ws_bytes,
                                long long rows, int d, float drop_p, unsigned long long seed, unsigned long long offset,
                                int dtype, void* stream);
int pk_ln_param_grads(const PkLnParamGrad* items, int n, long long rows, int d, int dtype, void* stream);

/* ---- Scaled-dot-product attention, head_dim 64 or 128 (K3): replaces F.scaled_dot_product_attention and the mask
 * assembly around it, pasero/models/modules.py:654-677,707-720 (fallback :742-771).
 *   q (B,T,H,hd), k/v (B,S,H,hd), o (B,T,H,hd): element strides *_bs (batch) and *_rs (row), head stride hd.
 *   key_pad: the reference's bool (B,S) key-padding mask, 1 byte per key, or NULL.  causal: query t attends keys
 *   s <= t + (S - T).  A query with every key masked outputs 0.  lse [B,H,T] fp32 (natural log, scaled scores). */
int pk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const unsigned char* key_pad,
                int B, int H, int T, int S, int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs,
                long long v_bs, long long v_rs, long long o_bs, long long o_rs, int causal, float scale, float drop_p,
                unsigned long long seed, unsigned long long offset, unsigned char* drop_mask, int dtype,
                void* stream);
/*   Attention-probability dropout (`dropout_p` of F.scaled_dot_product_attention, modules.py:707-720; the IWSLT2023
 *   recipes train with attention_dropout 0.1): o = (dropout(softmax(..)) v), softmax denominator unaffected.  With
 *   drop_p > 0 the forward call draws one keep bit per (query, key) from Philox(seed, offset) and stores them in
 *   drop_mask [B][H][T][8*ceil(S/64)] bytes (bit s%8 of byte s/8 of a row); pk_attn_bwd takes the same buffer.
 *   dq, dk, dv from d_o; `delta` [B,H,T] fp32 is scratch (rowsum(dO*O), produced and consumed inside the call) */
int pk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                float* delta, void* dq, void* dk, void* dv, const unsigned char* key_pad, int B, int H, int T, int S,
                int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs, long long v_bs,
                long long v_rs, long long o_bs, long long o_rs, long long do_bs, long long do_rs, long long dq_bs,
                long long dq_rs, long long dk_bs, long long dk_rs, long long dv_bs, long long dv_rs, int causal,
                float scale, float drop_p, const unsigned char* drop_mask, int dtype, void* stream);
/*   The same with ROTARY POSITIONS applied inside the kernels (RotaryEmbedding.forward on q and k, pasero/models/modules.py:
 *   617-623, 982-1025 — the "fused QKV-projection + RoPE" of the north_star, folded into the consumer instead of a pass of its
 *   own): q and k are the UNROTATED projection outputs; every kernel rotates the rows it loads (query t by the angle of position
 *   q_pos0 + t, key s by k_pos0 + s; fp32 arithmetic, rounded once, as pk_rope does) and the backward kernels rotate dQ / dK
 *   back as they leave, so dq / dk are the gradients of the unrotated projection and no pk_rope pass runs in either direction.
 *   cos_t / sin_t: fp32 [max_pos][hd / 2], 16-byte aligned (the tables pk_rope takes). */
int pk_attn_fwd_rope(const void* q, const void* k, const void* v, void* o, float* lse, const unsigned char* key_pad,
                     int B, int H, int T, int S, int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs,
                     long long v_bs, long long v_rs, long long o_bs, long long o_rs, int causal, float scale, float drop_p,
                     unsigned long long seed, unsigned long long offset, unsigned char* drop_mask, const float* cos_t,
                     const float* sin_t, int max_pos, int q_pos0, int k_pos0, int dtype, void* stream);
int pk_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                     float* delta, void* dq, void* dk, void* dv, const unsigned char* key_pad, int B, int H, int T, int S,
                     int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                     long long o_bs, long long o_rs, long long do_bs, long long do_rs, long long dq_bs, long long dq_rs,
                     long long dk_bs, long long dk_rs, long long dv_bs, long long dv_rs, int causal, float scale,
                     float drop_p, const unsigned char* drop_mask, const float* cos_t, const float* sin_t, int max_pos,
                     int q_pos0, int k_pos0, int dtype, void* stream);

/*   Attention weights softmax(q k^T * scale + masks) as a (B, T, H, S) tensor, fully masked rows = 0: what the reference's
 *   explicit path returns for `return_attn` / return_layers (modules.py:742-771).  Not used in training. */
int pk_attn_probs(const void* q, const void* k, void* probs, const unsigned char* key_pad, int B, int H, int T, int S,
                  int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs, int causal, float scale,
                  int dtype, void* stream);

/* ---- Token + positional embedding (K1): replaces Embedding.forward, `*= embed_scale`, `+= positions`, Dropout,
 * pasero/models/modules.py:916-933,435-457,467-484; pasero/models/transformer.py:727-744,866-878.
 *   out[tok] = dropout( E[clip(ids[tok])] * scale + pos[pos_start + tok % Tlen] )      (pos may be NULL)
 *   ids == NULL: E is a dense (ntok, d) input, out[tok] = dropout(E[tok] * scale + pos[...]) — the speech path,
 *   transformer.py:739-744 */
int pk_embed_fwd(const long long* ids, const void* E, const void* pos, void* out, long long ntok, int Tlen, int d,
                 long long V, float scale, int pos_start, float drop_p, unsigned long long seed,
                 unsigned long long offset, int dtype, void* stream);
/*   dE[V,d] = sum over tokens of dout * keep/(1-p) * scale into row ids[tok]; row pad_idx (nn.Embedding padding_idx) and
 *   rows without tokens are zero.  Deterministic: the token positions are radix-sorted by id (stable) and every row is
 *   summed in that fixed order in fp32, rounded once — no float atomics.  `workspace`: pk_embed_bwd_workspace(ntok, V, d)
 *   bytes.  d % 8 == 0, d <= 4096. */
size_t pk_embed_bwd_workspace(long long ntok, long long V, int d);
int pk_embed_bwd(const long long* ids, const void* dout, void* dE, void* workspace, size_t ws_bytes, long long ntok,
                 int d, long long V, long long pad_idx, float scale, float drop_p, unsigned long long seed,
                 unsigned long long offset, int dtype, void* stream);
/*   the same sums ADDED INTO a gradient that is already in dE — what autograd does with a separate (V, d) `add` when the
 *   embedding matrix is also the tied output projection (pasero/models/transformer.py:151-153, modules.py:935-947: the
 *   table receives the projection's dense dW and the lookups' sparse rows): every row with tokens = fp32(existing row) + its
 *   fixed-order sum, rounded once; rows without tokens and row pad_idx are not touched (no memset, no dense pass). */
int pk_embed_bwd_acc(const long long* ids, const void* dout, void* dE, void* workspace, size_t ws_bytes, long long ntok,
                     int d, long long V, long long pad_idx, float scale, float drop_p, unsigned long long seed,
                     unsigned long long offset, int dtype, void* stream);

/* ---- Label-smoothed cross-entropy (K6): replaces logits.float() + 2x F.cross_entropy(ignore_index=pad,
 * reduction='sum', label_smoothing) + 3 .item() syncs, pasero/models/transformer.py:354-380.
 *   per row i with target != pad:  nll_i = lse_i - x_i[y_i];  loss_i = (1-eps) nll_i + eps (lse_i - mean_c x_i[c])
 *   dlogits (optional, may alias logits) = softmax - eps/V - (1-eps) onehot, zero on pad rows. */
int pk_ce_rows(const void* logits, long long ld, const long long* target, void* dlogits, long long ldd,
               float* row_loss, float* row_nll, float* row_lse, long long rows, long long V, long long pad_idx,
               float eps, int dtype, void* stream);
/*   sums3 = { sum loss_i, sum nll_i, #(target != pad) } (fp32, accumulated in fp64, deterministic) */
int pk_ce_finalize(const float* row_loss, const float* row_nll, const long long* target, long long rows,
                   long long pad_idx, float* sums3, void* stream);

/* ---- small reductions / elementwise helpers of the autograd glue ---- */
/*   out[n] = sum_m x[m][n]   (bias gradients of nn.Linear, learned-position gradients) */
size_t pk_colsum_workspace(long long M, long long N);
int pk_colsum(const void* x, long long ld, void* out, long long M, long long N, void* workspace, size_t ws_bytes,
              int dtype, void* stream);
/*   out = x * keep/(1-p)  (nn.Dropout fwd; the same call with the same seed/offset is its backward) */
int pk_dropout(const void* x, void* out, long long n, float drop_p, unsigned long long seed,
               unsigned long long offset, int dtype, void* stream);
/*   out = x * (*dev_scalar) * host_scalar   (dev_scalar: device fp32 scalar or NULL) */
int pk_scale(const void* x, void* out, long long n, const float* dev_scalar, float host_scalar, int dtype,
             void* stream);

/* ---- Speech frontend (K7) helpers: the Conv1d stack of ConvolutionSubsampler (pasero/models/modules.py:774-834) is
 * run channels-last as pk_gemm calls on a strided window view of the zero-padded input; these are the pieces around.
 *   pk_act_fwd: out = act(x)        pk_act_bwd: out = dy * act'(x)          (modules.py:220-228)
 *   pk_glu_*:   x [rows][2C] -> a * sigmoid(b)  (nn.GLU over channels, modules.py:802) and its backward
 *   pk_col2im1d: dx[b][l][c] = sum_j dA[b*R + r][j*C + c] over r*stride + j - pad == l, r < Lout  (conv input grad) */
int pk_act_fwd(const void* x, void* out, long long n, int act, int dtype, void* stream);
int pk_act_bwd(const void* dy, const void* x, void* out, long long n, int act, int dtype, void* stream);
int pk_glu_fwd(const void* x, void* out, long long rows, int C, int dtype, void* stream);
int pk_glu_bwd(const void* dy, const void* x, void* dx, long long rows, int C, int dtype, void* stream);
int pk_col2im1d(const void* dA, void* dx, int B, int L, int C, int R, int Lout, int ksize, int stride, int pad,
                int dtype, void* stream);

/* ---- Whisper log-mel features (K8).  The reference computes them offline with the third-party
 * transformers.WhisperFeatureExtractor (examples/Whisper/extract-features.py:107-117, no mel code in the repo); this
 * restates that algorithm: wav (B clips, fp32, 16 kHz, `wav_stride` floats apart, wav_len[b] valid samples or NULL = all
 * 480000) -> out (B, 3000, 80) fp32 = ((max(log10 mel, max - 8)) + 4) / 4, frames x mel as extract-features.py:116. */
size_t pk_logmel_workspace(int B);
int pk_logmel(const float* wav, const long long* wav_len, long long wav_stride, float* out, void* workspace,
              size_t ws_bytes, int B, void* stream);

/* ---- Rotary position embedding (RoPE) on a packed projection output: replaces RotaryEmbedding.forward,
 * pasero/models/modules.py:982-1025 (GPT-J style halves: rotate(x) = cat(-x2, x1), head_dim 64 or 128).
 *   x, y: (rows, ld) with `total_cols` used columns; the first `ncols` columns (q|k heads of head_dim) are rotated by
 *   the angle of position pos_offset + row % Tlen, the rest (v) is copied.  cos_t / sin_t: fp32
 *   [max_pos][head_dim / 2] tables.  inverse = 1 applies the transposed rotation (backward pass).  y must not alias x. */
int pk_rope(const void* x, void* y, long long rows, int Tlen, long long ld, int ncols, int total_cols,
            const float* cos_t, const float* sin_t, int max_pos, int pos_offset, int inverse, int head_dim, int dtype,
            void* stream);

/* ---- "Next" row: fused gradient normalisation + global-norm clipping + Adam over all parameters.
 * Replaces the per-parameter Python loops of Trainer.train_step (pasero/training.py:455-482), clip_grad_norm_
 * (pasero/optimization.py:390-427) and Adam.step (pasero/optimization.py:56-149; fp32 moments, bf16 params updated
 * through fp32).  `table` (device int64): [param ptrs | grad ptrs | exp_avg ptrs | exp_avg_sq ptrs | numel], each
 * `ntensors` long; the chunk list maps workgroup -> (tensor, start), pk_mt_chunk_size() elements per chunk.
 *   pk_mt_sqnorm: partial[0..nchunks) = chunk sums of squares of this table's gradients; with `gnorm_out`:
 *                 gnorm_out[0] = scale * sqrt(sum partial_all[0..n_all)) — ONE norm over every table (parameter group,
 *                 dtype) whose partials earlier calls left in `partial_all`, like the reference's clip_grad_norm_
 *   pk_mt_adam  : g' = g * scale * min(1, max_norm / (gnorm + 1e-6)) (max_norm <= 0: no clipping); Adam update of m, v, p.
 *                 `bias_corr` (device fp32 [2 * ntensors]: 1 - beta1^step_t | sqrt(1 - beta2^step_t)) carries the
 *                 reference's per-parameter `state['step']` (optimization.py:120-125); NULL: every tensor at `step`.
 *   pk_mt_copy  : dst_t = src_t for every tensor; `table` = [src ptrs | dst ptrs | numel].  The gradient pack of a
 *                 data-parallel bucket (the reducer that replaces torch DDP's, pasero/training.py:243-250). */
int pk_mt_chunk_size(void);
int pk_mt_sqnorm(const long long* table, int ntensors, const int* chunk_tensor, const long long* chunk_start, int nchunks,
                 float scale, float* partial, const float* partial_all, int n_all, float* gnorm_out, int dtype,
                 void* stream);
int pk_mt_adam(const long long* table, int ntensors, const int* chunk_tensor, const long long* chunk_start, int nchunks,
               const float* gnorm, float scale, float max_norm, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, const float* bias_corr, int dtype, void* stream);
int pk_mt_copy(const long long* table, int ntensors, const int* chunk_tensor, const long long* chunk_start, int nchunks,
               int dtype, void* stream);

/* ---- Data-parallel gradient all-reduce over RCCL's C API (replaces the collective inside torch DDP's
 * reducer, pasero/training.py:243-250).  The RCCL symbols come from the library the process already loaded
 * (pk_comm_open: dlopen of PyTorch-ROCm's librccl), one communicator per process:
 *   pk_comm_unique_id   rank 0 draws the id (128 bytes) and the host side broadcasts it;
 *   pk_comm_init        every rank, collectively, on its current device;
 *   pk_comm_all_reduce_mean  buf <- mean over ranks, in place, on `stream`; schedule 0 = ncclAllReduce(avg),
 *                       1 = ncclReduceScatter(avg) + ncclAllGather, 2 = direct: grouped send/recv of the shards over all
 *                       xGMI links at once, a fixed-order fp32 sum of the n copies, grouped send/recv of the result
 *                       (needs `scratch` of `count` elements).  Schedules 1, 2: count % (8 * nranks) == 0.
 *   pk_comm_direct_plan the element offsets schedule 2 uses (shard length, own shard, per-peer send / receive offsets):
 *                       host arithmetic only — no GPU, no communicator — so the layout can be replayed on any machine */
int pk_comm_open(const char* librccl_path);
int pk_comm_unique_id(void* out, int nbytes);
int pk_comm_init(const void* id_bytes, int nranks, int rank);
int pk_comm_destroy(void);
int pk_comm_size(void);
int pk_comm_all_reduce_mean(void* buf, long long count, int dtype, int schedule, void* scratch, void* stream);
int pk_comm_direct_plan(long long count, int nranks, int rank, long long* shard, long long* own_off, long long* send_off,
                        long long* recv_off);

/* gated activation backward (SwiGLU / GEGLU FFN, transformer.py:1013-1016): h = act(z) * u
 *   dz = dh * u * act'(z)      du = dh * act(z) */
int pk_gated_act_bwd(const void* dh, const void* z, const void* u, void* dz, void* du, long long n, int act, int dtype,
                     void* stream);

/* ---- "Next" row: one incremental decoding step (T = 1) of the whole decoder stack in one call.
 * Replaces TransformerDecoder.forward with a non-empty `state` (pasero/models/transformer.py:831-898), the decoder
 * layer's self_attention / cross_attention / ffn (:1246-1320, :1224-1244, :1341-1417) and the KV-cache branch of
 * MultiheadAttention.forward (pasero/models/modules.py:621-641) for the stock layer (ReLU / GELU / SiLU feed-forward,
 * LayerNorm, sinusoidal / learned / no positions, post- or pre-norm).  All pointers are device memory except `plan`,
 * `plan->layers` and the three pointer tables, which are host memory.
 *   ids[B]            the tokens decoded at the previous step (one per sentence)
 *   t                 rows already in the self-attention caches; this step's K/V rows are written at index t
 *   pos_start         row of the positional table for this step (= positional shift + offset)
 *   self_k/self_v[l]  per-layer caches [B][cap][d]  (the reference concatenates a new tensor every step, :636-637)
 *   cross_kv[l]       per-layer [B*S][2d] = [k_proj(encoder_out) | v_proj(encoder_out)], computed once per sentence
 *                     (the reference recomputes both projections every step, :612-615)
 *   enc_mask          (B,S) bool key-padding mask of the encoder output, or NULL
 *   logits            [B][ld_logits] output of the (tied) projection
 * pk_argmax_rows: out[row * out_stride] = index of the first maximum of x[row][0..n) — the greedy choice
 * (pasero/decoding.py:1196-1205) without leaving the device. */
typedef struct {
    const void *qkv_w, *qkv_b, *out_w, *out_b, *ln1_g, *ln1_b;   /* self-attention [3d][d], [d][d]; self_attn_layer_norm */
    const void *cq_w, *cq_b, *cout_w, *cout_b, *ln2_g, *ln2_b;   /* cross-attention q and output projections; its norm */
    const void *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ln3_g, *ln3_b;   /* feed-forward [f][d], [d][f]; final_layer_norm */
} PkDecoderLayerWeights;
typedef struct {
    int n_layers, d, heads, ffn, act, prenorm, dtype, scaled_attn;
    long long vocab;
    float eps, embed_scale;
    const void *embed, *pos;                 /* token embedding [V][d]; positional table rows or NULL */
    const void *embed_ln_g, *embed_ln_b;     /* layernorm_embedding or NULL */
    const void *final_ln_g, *final_ln_b;     /* decoder.layer_norm (pre-norm) or NULL */
    const void* out_w;                       /* output projection [V][d] (= embed when tied) */
    const PkDecoderLayerWeights* layers;
} PkDecoderPlan;
size_t pk_decoder_step_scratch(const PkDecoderPlan* plan, int B);
int pk_decoder_step(const PkDecoderPlan* plan, const long long* ids, int B, int t, int pos_start, void* const* self_k,
                    void* const* self_v, long long cap, const void* const* cross_kv, const unsigned char* enc_mask,
                    int S, void* scratch, size_t scratch_bytes, void* logits, long long ld_logits, void* stream);
int pk_argmax_rows(const void* x, long long rows, long long n, long long ld, long long* out, long long out_stride,
                   int dtype, void* stream);

/* ---- "Next" row: feature collate.  Replaces the CPU pad_sequence + dtype cast of
 * utils.tokens_as_tensor for floating-point feature sequences (pasero/utils.py:709-736) as used on speech batches read
 * from NumpyFile rows (pasero/files.py:103-175): the ragged rows are uploaded once, concatenated, in the file's dtype
 * (src_dtype: 0 = f32, 1 = bf16, 2 = f16 — the offline Whisper / wav2vec features are fp16), and scattered on the device
 * into the zero-padded (B, Tmax, D) batch in the model dtype (out_dtype: PK_F32 / PK_BF16).
 *   src [total_rows][D]; offsets[B+1] (device int64): rows offsets[b] .. offsets[b+1] belong to sequence b. */
int pk_pad_rows(const void* src, int src_dtype, const long long* offsets, void* out, int out_dtype, int B,
                long long Tmax, int D, void* stream);

/* ---- One Transformer layer per call (host side of the hot path): forward and backward launch sequences of the STOCK
 * post-norm layer — `TransformerEncoderLayer.forward` / `TransformerDecoderLayer.forward`, pasero/models/transformer.py:
 * 1056-1099 and 1341-1417 (self-attention [-> cross-attention] -> feed-forward, each `x = LayerNorm(residual +
 * dropout(f(x)))`, :1043-1054) and what autograd makes of them — issued by ONE C call each instead of ~13 / ~27 Python
 * dispatches: the same entry points of this header in the same order with the same arguments (pk_gemm, pk_attn_fwd/bwd,
 * pk_gemm_ln_fwd or pk_gemm + pk_residual_ln_fwd, pk_residual_ln_bwd, pk_gemm_wgrad_group), so the results are bit-for-bit
 * those of the per-op path.  The caller allocates every buffer (activations kept for backward, outputs, gradients,
 * scratch); pointers are device memory, the structs themselves host memory.
 *   sub-block `self`: proj = x W_in^T + b_in ([q|k|v], W_in [3d][d]);  attn = softmax(q k^T scale [+ causal / key_pad]) v;
 *                     y = LN(x + dropout(attn W_o^T + b_o))          (z = the LayerNorm input, mean / rstd its statistics)
 *   sub-block `cross` (decoder): proj = y_self W_in[0:d]^T + b_in[0:d];  kv = enc W_in[d:3d]^T + b_in[d:3d];  as above
 *   sub-block `ffn`:  h = act(y W_1^T + b_1)  (pre = the pre-activation when act is not none / ReLU);  y = LN(y + dropout(h W_2^T + b_2))
 *   dropout of sub-block i draws Philox(seed, drop_offset_i) (conventions above); attention-probability dropout: not here.
 *   fused_tail: the block ends run as pk_gemm_ln_fwd (needs d = 512), else pk_gemm + pk_residual_ln_fwd.
 *   prenorm (`--encoder-prenorm` / `--decoder-prenorm`, transformer.py:1049-1054): every sub-block is
 *     z = x + dropout(f(LN(x))) instead — `ln_out` keeps LN(x), `z` the sub-block's output (the layer's output is ffn.z),
 *     mean / rstd are the statistics of the sub-block's INPUT, `y` is scratch; in backward the gradient of the residual
 *     branch enters the LayerNorm backward kernel as its `dz_extra` (the per-op path adds it in a separate pass).
 * pk_layer_bwd_sizes: bytes of `scratch` (gradient temporaries) and `ws` (split-K / grouped weight-gradient / LayerNorm
 * parameter-gradient workspaces) the backward call needs.  Weight gradients: dw_in [3d][d], db_in [3d], dw_o [d][d], ...
 * in the operands' type, all of one layer in ONE grouped launch. */
typedef struct {
    const void *w_in, *b_in, *w_o, *b_o, *ln_g, *ln_b; /* parameters (biases may be NULL) */
    void *proj, *kv, *attn, *z, *y;                     /* kept by forward: [rows][3d] (cross: [rows][d]), cross [B*S][2d], [rows][d] x 3 */
    void* ln_out;                                       /* pre-norm only: LayerNorm(block input) [rows][d] */
    float *lse, *mean, *rstd;                           /* [B][H][T], [rows], [rows] */
    void *dw_in, *db_in, *dw_o, *db_o, *dln_g, *dln_b;  /* written by backward */
    unsigned long long drop_offset;
} PkAttnBlock;
typedef struct {
    const void *w1, *b1, *w2, *b2, *ln_g, *ln_b;
    void *h, *pre, *z, *y;                              /* [rows][f], [rows][f] or NULL, [rows][d], [rows][d] */
    void* ln_out;                                       /* pre-norm only */
    unsigned char* bits;                                /* ReLU mask as bits [rows][f / 8] (pk_gemm_relu_bits) or NULL */
    float *mean, *rstd;
    void *dw1, *db1, *dw2, *db2, *dln_g, *dln_b;
    unsigned long long drop_offset;
} PkFfnBlock;
typedef struct {
    int dtype, is_decoder, fused_tail, act, B, T, S, d, f, heads, prenorm;
    float eps, drop_p, attn_scale;
    unsigned long long seed;
    const void *x, *enc;                                /* layer input [B*T][d]; encoder output [B*S][d] (decoder) */
    const unsigned char *self_pad, *cross_pad;          /* (B,T) / (B,S) bool key-padding masks or NULL */
    PkAttnBlock self, cross;
    PkFfnBlock ffn;
    const void* dy;                                     /* backward: gradient of the layer output (ffn.y; pre-norm: ffn.z) */
    void *dx, *denc;                                    /* backward: gradients of x and (decoder) enc */
    const void* denc_prev;                              /* decoder: NULL, or the enc gradient accumulated so far by the layers
                                                          * above (may be denc itself): denc = this layer's + denc_prev */
    /* pre-norm layers with dropout, backward — the hand-over of a masked gradient between stacked layers (autograd.DropLink):
     * dy_masked: NULL, or dy already pushed through THIS layer's feed-forward dropout mask (written by the consumer of the
     *   layer's output: the layer above): the layer does not draw that mask again;
     * dx_masked: NULL, or room for [B*T][d]: the layer's last LayerNorm backward also writes dx through the dropout mask
     *   (drop_p, seed, dx_mask_offset) — the feed-forward block end of the layer BELOW, whose dy_masked it becomes. */
    const void* dy_masked;
    void* dx_masked;
    unsigned long long dx_mask_offset;
    void *scratch, *ws;
    size_t scratch_bytes, ws_bytes;
    void* stream;
} PkLayer;
int pk_layer_fwd(const PkLayer* layer);
/* bytes of workspace pk_layer_fwd can use through layer->ws / ws_bytes (0: none): a pre-norm layer whose fc2 has a long
 * contraction at a few thousand rows (NLLB-1.3B's 8192 -> 1024 at the IWSLT recipe's 2048-row decoder batch) runs that GEMM
 * as K-slabs + reduction, like its dX GEMM in pk_layer_bwd; without the workspace every forward GEMM runs unsplit.
 * Replaces: nothing in the reference (nn.Linear, pasero/models/modules.py:92-96, leaves the split to the BLAS library). */
size_t pk_layer_fwd_ws(const PkLayer* layer);
int pk_layer_bwd_sizes(const PkLayer* layer, size_t* scratch_bytes, size_t* ws_bytes);
int pk_layer_bwd(const PkLayer* layer);

#ifdef __cplusplus
}
#endif
#endif /* PASERO_HIP_H */
