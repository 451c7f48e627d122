// One incremental decoding step (one new token per sentence) of the Transformer decoder stack, driven from native code:
// the ~75 kernel launches of a step are issued by one C call instead of ~90 Python-level op dispatches (measured: the
// Python path costs 2.05 ms per step at any batch size — pure launch overhead).
//
// Replaces, for T = 1 and a non-empty incremental state: TransformerDecoder.forward (pasero/models/transformer.py:831-898),
// TransformerDecoderLayer.forward/.self_attention/.cross_attention/.ffn (:1341-1417, :1246-1320, :1224-1244) and the
// KV-cache branch of MultiheadAttention.forward (pasero/models/modules.py:621-641).  Differences by design:
//   * the self-attention K/V cache is a preallocated [B][cap][D] buffer per layer that the new row is appended to
//     (the reference re-concatenates the whole cache every step, modules.py:636-637);
//   * the cross-attention K/V projections of the encoder output are computed once per sentence and passed in
//     (the reference recomputes k_proj/v_proj(encoder_out) every step, modules.py:612-615).
// Every matrix product, LayerNorm and attention call is the same pk_* kernel the training path uses.
#include <utility>
#include "common.h"

extern "C" {
int pk_gemm(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact, long long M,
            long long N, long long K, long long lda, long long ldb, long long ldc, long long ldaux, long long ldpre,
            int a_col, int b_col, int act, int mode, float alpha, int dtype, int splitk, void* workspace,
            size_t ws_bytes, void* asum_out, void* stream);
int pk_residual_ln_fwd(const void* x, const void* residual, const void* gamma, const void* beta, void* z_out,
                       void* y_out, float* mean, float* rstd, long long rows, int d, float eps, float drop_p,
                       unsigned long long seed, unsigned long long offset, int dtype, void* stream);
int pk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const unsigned char* key_pad,
                int B, int H, int T, int S, int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs,
                long long v_bs, long long v_rs, long long o_bs, long long o_rs, int causal, float scale, float drop_p,
                unsigned long long seed, unsigned long long offset, unsigned char* drop_mask, int dtype, void* stream);
int pk_embed_fwd(const long long* ids, const void* E, const void* pos, void* out, long long ntok, int Tlen, int d,
                 long long V, float scale, int pos_start, float drop_p, unsigned long long seed,
                 unsigned long long offset, int dtype, void* stream);
}

// ---- plan: mirrors include/pasero_hip.h (PkDecoderLayerWeights / PkDecoderPlan) ----
struct PkDecoderLayerWeights {
    const void *qkv_w, *qkv_b;    // self-attention packed projection [3D][D], [3D]
    const void *out_w, *out_b;    // self-attention output projection [D][D]
    const void *ln1_g, *ln1_b;    // self_attn_layer_norm
    const void *cq_w, *cq_b;      // cross-attention query projection [D][D]
    const void *cout_w, *cout_b;  // cross-attention output projection
    const void *ln2_g, *ln2_b;    // encoder_attn_layer_norm
    const void *fc1_w, *fc1_b;    // [F][D]
    const void *fc2_w, *fc2_b;    // [D][F]
    const void *ln3_g, *ln3_b;    // final_layer_norm
};
struct PkDecoderPlan {
    int n_layers, d, heads, ffn, act, prenorm, dtype, scaled_attn;
    long long vocab;
    float eps, embed_scale;
    const void* embed;        // [V][D] token embedding
    const void* pos;          // positional table rows [.][D] or NULL
    const void* embed_ln_g;   // layernorm_embedding (NULL = Identity)
    const void* embed_ln_b;
    const void* final_ln_g;   // decoder.layer_norm (pre-norm stacks; NULL = Identity)
    const void* final_ln_b;
    const void* out_w;        // output projection [V][D] (the embedding itself when tied)
    const PkDecoderLayerWeights* layers;
};

namespace {

// append this step's K and V rows (columns [D, 3D) of the packed projection) to the per-layer caches at position t
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ ck,
                                                        T* __restrict__ cv, int d, int t, long long cap) {
    constexpr int EPV = 16 / sizeof(T);
    const int b = blockIdx.x;
    const T* src = qkv + (long long)b * 3 * d + d;
    T* dk = ck + ((long long)b * cap + t) * d;
    T* dv = cv + ((long long)b * cap + t) * d;
    for (int c = threadIdx.x * EPV; c < d; c += blockDim.x * EPV) {
        store16<T>(dk + c, load16<T>(src + c));
        store16<T>(dv + c, load16<T>(src + d + c));
    }
}

// greedy choice: first index of the row maximum (torch.argmax semantics on ties: lowest index)
template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ x, long long ld, long long n,
                                                          long long* __restrict__ out, long long out_stride) {
    const long long row = blockIdx.x;
    const T* p = x + row * ld;
    float best = -INFINITY;
    long long bi = n;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        float v = to_f32(p[i]);
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    __shared__ float sv[256];
    __shared__ long long si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            float v = sv[threadIdx.x + s];
            long long i = si[threadIdx.x + s];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row * out_stride] = si[0] < n ? si[0] : 0;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct Scratch {
    char *x0, *x1, *h, *qkv, *q, *o, *ff;
    float *lse, *mean, *rstd;
    size_t total;
};
Scratch carve(const PkDecoderPlan* p, int B, char* base) {
    const size_t es = p->dtype == PK_F32 ? 4 : 2;
    Scratch s;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = base ? base + off : nullptr; off += align256(bytes); return r; };
    s.x0 = take((size_t)B * p->d * es);
    s.x1 = take((size_t)B * p->d * es);
    s.h = take((size_t)B * p->d * es);
    s.qkv = take((size_t)B * 3 * p->d * es);
    s.q = take((size_t)B * p->d * es);
    s.o = take((size_t)B * p->d * es);
    s.ff = take((size_t)B * p->ffn * es);
    s.lse = (float*)take((size_t)B * p->heads * 4);
    s.mean = (float*)take((size_t)B * 4);
    s.rstd = (float*)take((size_t)B * 4);
    s.total = off;
    return s;
}

}  // namespace

extern "C" size_t pk_decoder_step_scratch(const PkDecoderPlan* plan, int B) {
    return carve(plan, B > 0 ? B : 1, nullptr).total;
}

#define RC(call)                 \
    do {                         \
        int rc_ = (call);        \
        if (rc_ != 0) return rc_; \
    } while (0)

extern "C" int pk_decoder_step(const PkDecoderPlan* plan, const long long* ids, int B, int t, int pos_start,
                               void* const* self_k, void* const* self_v, long long cap, const void* const* cross_kv,
                               const unsigned char* enc_mask, int S, void* scratch, size_t scratch_bytes, void* logits,
                               long long ld_logits, void* stream) {
    PK_CHECK_ARG(plan && ids && self_k && self_v && cross_kv && scratch && logits, "pk_decoder_step: null argument");
    PK_CHECK_ARG(B > 0 && t >= 0 && t < cap && S > 0, "pk_decoder_step: bad sizes (B=%d t=%d cap=%lld S=%d)", B, t, cap, S);
    PK_CHECK_ARG(plan->d % 64 == 0 && plan->heads * 64 == plan->d, "pk_decoder_step: head_dim must be 64");
    PK_CHECK_ARG(plan->dtype == PK_F32 || plan->dtype == PK_BF16 || plan->dtype == PK_F16, "pk_decoder_step: bad dtype");
    Scratch s = carve(plan, B, (char*)scratch);
    PK_CHECK_ARG(scratch_bytes >= s.total, "pk_decoder_step: scratch too small (%zu < %zu)", scratch_bytes, s.total);
    const int D = plan->d, H = plan->heads, F = plan->ffn, dt = plan->dtype;
    const float scale = plan->scaled_attn ? 0.125f : 1.f;
    hipStream_t st = (hipStream_t)stream;
    char *x = s.x0, *xn = s.x1;

    // residual + LayerNorm of a sub-block: post-norm  x <- LN(y + x);  pre-norm  x <- y + x (no LN here)
    auto block_end = [&](const void* y, const void* g, const void* b) -> int {
        if (plan->prenorm) RC(pk_residual_ln_fwd(y, x, nullptr, nullptr, xn, nullptr, nullptr, nullptr, B, D, plan->eps, 0.f, 0, 0, dt, stream));
        else RC(pk_residual_ln_fwd(y, x, g, b, nullptr, xn, s.mean, s.rstd, B, D, plan->eps, 0.f, 0, 0, dt, stream));
        std::swap(x, xn);
        return 0;
    };
    // input of a sub-block: pre-norm LN(x) -> h, post-norm x itself
    auto block_in = [&](const void* g, const void* b, const char** in) -> int {
        if (!plan->prenorm) { *in = x; return 0; }
        RC(pk_residual_ln_fwd(x, nullptr, g, b, nullptr, s.h, s.mean, s.rstd, B, D, plan->eps, 0.f, 0, 0, dt, stream));
        *in = s.h;
        return 0;
    };
    auto linear = [&](const void* in, const void* w, const void* b, void* out, int n, int k, int act) -> int {
        return pk_gemm(in, w, out, b, nullptr, nullptr, B, n, k, k, k, n, 0, 0, 0, 0, act, 0, 1.f, dt, 1, nullptr, 0,
                       nullptr, stream);
    };

    // token + position embedding (transformer.py:866-878): one row per sentence, position = pos_start
    RC(pk_embed_fwd(ids, plan->embed, plan->pos, x, B, 1, D, plan->vocab, plan->embed_scale, pos_start, 0.f, 0, 0, dt,
                    stream));
    if (plan->embed_ln_g) {
        RC(pk_residual_ln_fwd(x, nullptr, plan->embed_ln_g, plan->embed_ln_b, nullptr, xn, s.mean, s.rstd, B, D, plan->eps,
                              0.f, 0, 0, dt, stream));
        std::swap(x, xn);
    }
    const size_t es = dt == PK_F32 ? 4 : 2;
    for (int l = 0; l < plan->n_layers; ++l) {
        const PkDecoderLayerWeights& w = plan->layers[l];
        const char* in;
        // ---- self-attention over the cache (keys 0..t) ----
        RC(block_in(w.ln1_g, w.ln1_b, &in));
        RC(linear(in, w.qkv_w, w.qkv_b, s.qkv, 3 * D, D, PK_ACT_NONE));
        if (dt != PK_F32)  // a 16-bit row copy: bf16 and fp16 alike
            hipLaunchKernelGGL((kv_append_kernel<bf16>), dim3(B), dim3(64), 0, st, (const bf16*)s.qkv, (bf16*)self_k[l],
                               (bf16*)self_v[l], D, t, cap);
        else
            hipLaunchKernelGGL((kv_append_kernel<float>), dim3(B), dim3(128), 0, st, (const float*)s.qkv,
                               (float*)self_k[l], (float*)self_v[l], D, t, cap);
        PK_LAUNCH_CHECK();
        RC(pk_attn_fwd(s.qkv, self_k[l], self_v[l], s.o, s.lse, nullptr, B, H, 1, t + 1, 64, 3LL * D, 3LL * D, cap * D, D,
                       cap * D, D, D, D, 0, scale, 0.f, 0, 0, nullptr, dt, stream));
        RC(linear(s.o, w.out_w, w.out_b, s.q, D, D, PK_ACT_NONE));
        RC(block_end(s.q, w.ln1_g, w.ln1_b));
        // ---- cross-attention over the cached projections of the encoder output ----
        RC(block_in(w.ln2_g, w.ln2_b, &in));
        RC(linear(in, w.cq_w, w.cq_b, s.q, D, D, PK_ACT_NONE));
        const char* ckv = (const char*)cross_kv[l];
        RC(pk_attn_fwd(s.q, ckv, ckv + (size_t)D * es, s.o, s.lse, enc_mask, B, H, 1, S, 64, D, D, (long long)S * 2 * D,
                       2LL * D, (long long)S * 2 * D, 2LL * D, D, D, 0, scale, 0.f, 0, 0, nullptr, dt, stream));
        RC(linear(s.o, w.cout_w, w.cout_b, s.q, D, D, PK_ACT_NONE));
        RC(block_end(s.q, w.ln2_g, w.ln2_b));
        // ---- feed-forward ----
        RC(block_in(w.ln3_g, w.ln3_b, &in));
        RC(linear(in, w.fc1_w, w.fc1_b, s.ff, F, D, plan->act));
        RC(linear(s.ff, w.fc2_w, w.fc2_b, s.q, D, F, PK_ACT_NONE));
        RC(block_end(s.q, w.ln3_g, w.ln3_b));
    }
    if (plan->final_ln_g) {
        RC(pk_residual_ln_fwd(x, nullptr, plan->final_ln_g, plan->final_ln_b, nullptr, xn, s.mean, s.rstd, B, D, plan->eps,
                              0.f, 0, 0, dt, stream));
        std::swap(x, xn);
    }
    // output projection (tied: x E^T, modules.py:935-947)
    RC(pk_gemm(x, plan->out_w, logits, nullptr, nullptr, nullptr, B, plan->vocab, D, D, D, ld_logits, 0, 0, 0, 0, PK_ACT_NONE,
               0, 1.f, dt, 1, nullptr, 0, nullptr, stream));
    return 0;
}

extern "C" int pk_argmax_rows(const void* x, long long rows, long long n, long long ld, long long* out,
                              long long out_stride, int dtype, void* stream) {
    PK_CHECK_ARG(x && out && n > 0 && ld >= n, "pk_argmax_rows: bad arguments");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PK_BF16)
        hipLaunchKernelGGL((argmax_rows_kernel<bf16>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16*)x, ld, n, out, out_stride);
    else if (dtype == PK_F16)
        hipLaunchKernelGGL((argmax_rows_kernel<f16>), dim3((unsigned)rows), dim3(256), 0, st, (const f16*)x, ld, n, out, out_stride);
    else if (dtype == PK_F32)
        hipLaunchKernelGGL((argmax_rows_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)x, ld, n, out, out_stride);
    else PK_CHECK_ARG(false, "pk_argmax_rows: dtype %d not supported", dtype);
    PK_LAUNCH_CHECK();
    return 0;
}
