// One Transformer layer per call: the launch sequences of the stock post-norm encoder / decoder layer, forward and
// backward, issued from C (include/pasero_hip.h: pk_layer_fwd / pk_layer_bwd).  Nothing is computed here: every step is
// a call of another entry point of this library, in the order and with the arguments the per-op path
// (pasero_amd/autograd.py: PackedLinearFn, AttentionFn, LinearResidualLnFn / LinearFn + ResidualLayerNormFn,
// FFNResidualLnFn / FFNFn, WGradSinkFn) uses — which is what makes the two paths agree bit for bit — minus ~25 Python
// dispatches, tensor allocations and ctypes marshallings per layer and direction.
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/pasero_hip.h"

extern "C" void pk_set_error(const char* fmt, ...);

namespace {

#define PK_TRY(call)                 \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != 0) return rc_;    \
    } while (0)
#define PK_REQ(cond, ...)            \
    do {                             \
        if (!(cond)) {               \
            pk_set_error(__VA_ARGS__); \
            return -1;               \
        }                            \
    } while (0)

inline size_t esz(int dtype) { return dtype == PK_F32 ? 4 : 2; }
inline char* at(const void* p, long long elems, int dtype) { return (char*)p + (size_t)elems * esz(dtype); }

// pasero_amd/functional.py: choose_splitk — split the contraction of a GEMM whose output has too few tiles
int choose_splitk(long long M, long long N, long long K) {
    const long long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles > 256 || K < 1024) {
        const long long t256 = ((M + 255) / 256) * ((N + 255) / 256);
        return (80 <= t256 && t256 < 160 && K >= 2048 && M % 8 == 0 && N % 8 == 0 && K % 8 == 0) ? 2 : 1;
    }
    return (int)std::max(1LL, std::min(512 / tiles, K / 512));
}
inline size_t splitk_ws(int sk, long long M, long long N) { return sk > 1 ? (size_t)2 * sk * M * (N + 1) * 4 : 0; }
// Forward GEMMs are not split as a rule (their outputs fill the chip).  The exception: a projection back to d from a long
// contraction at a few thousand rows — NLLB-1.3B's fc2 (8192 -> 1024) at the IWSLT recipe's 2048-row decoder batch is 32 tiles
// of 256 x 256: 123 us on the 128-tile kernel, ~40 as 8 K-slabs of the 256-tile kernel + reduction (what its dX GEMM already
// does).  4 here = 8 slabs there (pk_gemm re-derives the factor for its 256-tiles, capped at twice the caller's).
inline int fwd_split(long long M, long long N, long long K) {
    // (M <= 2048 only: from 4096 rows up the 128 x 256-tile form takes these GEMMs, and a batch and its halves must not fall on
    // two sides of a split rule — tests/test_configs_fullsize_gpu.py compares them)
    return (K >= 4096 && K % 512 == 0 && N <= 1024 && N % 256 == 0 && M >= 512 && M <= 2048 && M % 256 == 0) ? 4 : 1;
}

struct Bump {  // the backward's gradient temporaries: one region of `scratch` each (256-byte aligned)
    char* base;
    size_t off = 0;
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    }
};

struct Bwd {  // where everything of one backward call lives
    void *dres_f, *dsub_f, *dh, *dy_mid, *dres_c, *dsub_c, *dcattn, *dq, *dkv, *dy_self, *dres_s, *dsub_s, *dattn, *dproj;
    void* dln;  // pre-norm: gradient of a LayerNorm output (one at a time)
    float* delta;
    size_t scratch_bytes, ws_split, ws_group, ws_ln;
};

// the plan of the backward call: regions of `scratch`, sizes of the three workspaces (regions of `ws`)
int plan_bwd(const PkLayer& L, char* scratch, Bwd* b) {
    const long long rows = (long long)L.B * L.T, rows_kv = (long long)L.B * L.S, d = L.d, f = L.f;
    const size_t e = esz(L.dtype);
    const bool drop = L.drop_p > 0.f;
    Bump bump{scratch};
    b->dres_f = bump.take(rows * d * e);
    b->dsub_f = drop ? bump.take(rows * d * e) : b->dres_f;
    b->dh = bump.take(rows * f * e);
    b->dy_mid = bump.take(rows * d * e);  // gradient of the feed-forward block's input
    b->dres_c = b->dsub_c = b->dcattn = b->dq = b->dkv = b->dy_self = nullptr;
    if (L.is_decoder) {
        b->dres_c = bump.take(rows * d * e);
        b->dsub_c = drop ? bump.take(rows * d * e) : b->dres_c;
        b->dcattn = bump.take(rows * d * e);
        b->dq = bump.take(rows * d * e);
        b->dkv = bump.take(rows_kv * 2 * d * e);
        b->dy_self = bump.take(rows * d * e);
    }
    b->dres_s = bump.take(rows * d * e);
    b->dsub_s = drop ? bump.take(rows * d * e) : b->dres_s;
    b->dattn = bump.take(rows * d * e);
    b->dproj = bump.take(rows * 3 * d * e);
    b->dln = L.prenorm ? bump.take(rows * d * e) : nullptr;
    b->delta = (float*)bump.take((size_t)L.B * L.heads * std::max(L.T, 1) * sizeof(float));
    b->scratch_bytes = bump.off;
    // split-K workspaces of the dX GEMMs that go through `_dx_gemm` (autograd.py): the largest one
    size_t ws = 0;
    ws = std::max(ws, splitk_ws(choose_splitk(rows, d, d), rows, d));          // out-proj dX (self / cross), cross q dX
    ws = std::max(ws, splitk_ws(choose_splitk(rows, d, 3 * d), rows, d));      // q|k|v dX
    if (!L.fused_tail || L.prenorm) ws = std::max(ws, splitk_ws(choose_splitk(rows, d, f), rows, d));  // fc1 dX (FFNFn path)
    if (L.is_decoder) ws = std::max(ws, splitk_ws(choose_splitk(rows_kv, d, 2 * d), rows_kv, d));  // k|v dX
    b->ws_split = (ws + 255) & ~(size_t)255;
    b->ws_ln = (pk_residual_ln_bwd_workspace(rows, (int)d) + 255) & ~(size_t)255;
    return 0;
}

// the layer's weight-gradient problems in the order the per-op path hands them to its group
int wgrad_problems(const PkLayer& L, const Bwd& b, PkWgradProblem* p) {
    const long long rows = (long long)L.B * L.T, rows_kv = (long long)L.B * L.S, d = L.d, f = L.f;
    // what each projection read: post-norm — the previous sub-block's output; pre-norm — its own LayerNorm's output
    const void* ffn_in = L.prenorm ? L.ffn.ln_out : (L.is_decoder ? L.cross.y : L.self.y);
    const void* cross_in = L.prenorm ? L.cross.ln_out : L.self.y;
    const void* self_in = L.prenorm ? L.self.ln_out : L.x;
    int n = 0;
    p[n++] = PkWgradProblem{b.dsub_f, L.ffn.h, L.ffn.dw2, L.ffn.db2, d, f, rows, d, f, f};        // fc2: dZ^T H
    p[n++] = PkWgradProblem{b.dh, ffn_in, L.ffn.dw1, L.ffn.db1, f, d, rows, f, d, d};             // fc1: dH^T Y
    if (L.is_decoder) {
        p[n++] = PkWgradProblem{b.dsub_c, L.cross.attn, L.cross.dw_o, L.cross.db_o, d, d, rows, d, d, d};
        p[n++] = PkWgradProblem{b.dkv, L.enc, at(L.cross.dw_in, d * d, L.dtype), L.cross.db_in ? at(L.cross.db_in, d, L.dtype) : nullptr,
                                2 * d, d, rows_kv, 2 * d, d, d};
        p[n++] = PkWgradProblem{b.dq, cross_in, L.cross.dw_in, L.cross.db_in, d, d, rows, d, d, d};
    }
    p[n++] = PkWgradProblem{b.dsub_s, L.self.attn, L.self.dw_o, L.self.db_o, d, d, rows, d, d, d};
    p[n++] = PkWgradProblem{b.dproj, self_in, L.self.dw_in, L.self.db_in, 3 * d, d, rows, 3 * d, d, d};
    return n;
}

int check(const PkLayer& L) {
    PK_REQ(L.dtype == PK_BF16 || L.dtype == PK_F16, "pk_layer: 16-bit layers only (dtype %d)", L.dtype);
    PK_REQ(L.B > 0 && L.T > 0 && L.d > 0 && L.f > 0 && L.heads > 0 && L.d % L.heads == 0, "pk_layer: bad sizes");
    PK_REQ(L.x && L.self.w_in && L.self.w_o && L.self.ln_g && L.ffn.w1 && L.ffn.w2 && L.ffn.ln_g, "pk_layer: null parameter");
    PK_REQ(!L.is_decoder || (L.enc && L.S > 0 && L.cross.w_in && L.cross.w_o && L.cross.ln_g), "pk_layer: decoder layer without encoder output");
    PK_REQ(!L.ffn.bits || (L.act == PK_ACT_RELU && L.f % 8 == 0), "pk_layer: the bit mask is the ReLU feed-forward's");
    return 0;
}

// the end of a post-norm sub-block: y = LN(residual + dropout(a W^T + b)), z and the statistics kept
int block_end(const PkLayer& L, const void* a, long long K, const void* w, const void* bias, const void* residual,
              const void* g, const void* be, void* z, void* y, float* mean, float* rstd, unsigned long long offset) {
    const long long rows = (long long)L.B * L.T, d = L.d;
    if (L.fused_tail)
        return pk_gemm_ln_fwd(a, w, bias, residual, g, be, z, y, mean, rstd, rows, d, K, K, K, d, L.eps, L.drop_p, L.seed,
                              offset, L.dtype, L.stream);
    // the projection's output passes through the z buffer (the LayerNorm kernel holds a row in registers: in place is safe)
    PK_TRY(pk_gemm(a, w, z, bias, nullptr, nullptr, rows, d, K, K, K, d, 0, 0, 0, 0, PK_ACT_NONE, 0, 1.f, L.dtype, 1, nullptr, 0,
                   nullptr, L.stream));
    return pk_residual_ln_fwd(z, residual, g, be, z, y, mean, rstd, rows, (int)d, L.eps, L.drop_p, L.seed, offset, L.dtype,
                              L.stream);
}

// pre-norm sub-block pieces: ln_out = LN(in) (statistics kept), and z = in + dropout(o) at its end
int pre_norm(const PkLayer& L, const void* in, const void* g, const void* be, void* ln_out, float* mean, float* rstd) {
    return pk_residual_ln_fwd(in, nullptr, g, be, nullptr, ln_out, mean, rstd, (long long)L.B * L.T, L.d, L.eps, 0.f, 0, 0, L.dtype, L.stream);
}
int pre_end(const PkLayer& L, const void* o, const void* in, void* z, unsigned long long offset) {
    return pk_residual_ln_fwd(o, in, nullptr, nullptr, z, nullptr, nullptr, nullptr, (long long)L.B * L.T, L.d, 0.f, L.drop_p, L.seed,
                              offset, L.dtype, L.stream);
}

int layer_fwd_prenorm(const PkLayer& L) {
    const long long rows = (long long)L.B * L.T, rows_kv = (long long)L.B * L.S, d = L.d, f = L.f;
    const int hd = L.d / L.heads, dt = L.dtype;
    auto linear = [&](const void* a, const void* w, const void* bias, void* out, long long M, long long N, long long K, int act, void* pre) {
        // (forward split-K: fwd_split gates on M — inside its row range the slab count depends on N and K only; across it the
        // rows agree to fp32 round-off of the accumulation, functional.fwd_split)
        int sk = (pre || dt == PK_F32) ? 1 : fwd_split(M, N, K);
        const size_t need = splitk_ws(sk, M, N);
        if (sk > 1 && (!L.ws || L.ws_bytes < need)) sk = 1;
        return pk_gemm(a, w, out, bias, nullptr, pre, M, N, K, K, K, N, 0, pre ? N : 0, 0, 0, act, 0, 1.f, dt, sk, sk > 1 ? L.ws : nullptr,
                       sk > 1 ? need : 0, nullptr, L.stream);
    };
    PK_REQ(L.self.ln_out && L.ffn.ln_out && (!L.is_decoder || L.cross.ln_out), "pk_layer_fwd: pre-norm layer without ln_out buffers");
    // the end of a block (z = in + dropout(o)) and the LayerNorm that starts the NEXT block of the layer in one pass: the
    // same kernel writes z and LN(z) — with the next block's gamma / beta, its statistics taken on the rounded z as a
    // separate pass over z would take them (bit for bit the two launches it replaces)
    auto end_and_next_norm = [&](const void* o, const void* in, void* z, unsigned long long offset, const void* g, const void* be,
                                 void* ln_out, float* mean, float* rstd) {
        return pk_residual_ln_fwd(o, in, g, be, z, ln_out, mean, rstd, rows, (int)d, L.eps, L.drop_p, L.seed, offset, dt, L.stream);
    };
    // ---- self-attention ----
    PK_TRY(pre_norm(L, L.x, L.self.ln_g, L.self.ln_b, L.self.ln_out, L.self.mean, L.self.rstd));
    PK_TRY(linear(L.self.ln_out, L.self.w_in, L.self.b_in, L.self.proj, rows, 3 * d, d, PK_ACT_NONE, nullptr));
    PK_TRY(pk_attn_fwd(L.self.proj, at(L.self.proj, d, dt), at(L.self.proj, 2 * d, dt), L.self.attn, L.self.lse,
                       L.is_decoder ? nullptr : L.self_pad, L.B, L.heads, L.T, L.T, hd, (long long)L.T * 3 * d, 3 * d,
                       (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * d, d,
                       L.is_decoder && L.T > 1, L.attn_scale, 0.f, 0, 0, nullptr, dt, L.stream));
    PK_TRY(linear(L.self.attn, L.self.w_o, L.self.b_o, L.self.y, rows, d, d, PK_ACT_NONE, nullptr));
    if (L.is_decoder)
        PK_TRY(end_and_next_norm(L.self.y, L.x, L.self.z, L.self.drop_offset, L.cross.ln_g, L.cross.ln_b, L.cross.ln_out, L.cross.mean,
                                 L.cross.rstd));
    else
        PK_TRY(end_and_next_norm(L.self.y, L.x, L.self.z, L.self.drop_offset, L.ffn.ln_g, L.ffn.ln_b, L.ffn.ln_out, L.ffn.mean,
                                 L.ffn.rstd));
    const void* z = L.self.z;
    // ---- cross-attention (decoder) ----
    if (L.is_decoder) {
        PK_TRY(linear(L.cross.ln_out, L.cross.w_in, L.cross.b_in, L.cross.proj, rows, d, d, PK_ACT_NONE, nullptr));
        PK_TRY(linear(L.enc, at(L.cross.w_in, d * d, dt), L.cross.b_in ? at(L.cross.b_in, d, dt) : nullptr, L.cross.kv, rows_kv, 2 * d, d,
                      PK_ACT_NONE, nullptr));
        PK_TRY(pk_attn_fwd(L.cross.proj, L.cross.kv, at(L.cross.kv, d, dt), L.cross.attn, L.cross.lse, L.cross_pad, L.B, L.heads,
                           L.T, L.S, hd, (long long)L.T * d, d, (long long)L.S * 2 * d, 2 * d, (long long)L.S * 2 * d, 2 * d,
                           (long long)L.T * d, d, 0, L.attn_scale, 0.f, 0, 0, nullptr, dt, L.stream));
        PK_TRY(linear(L.cross.attn, L.cross.w_o, L.cross.b_o, L.cross.y, rows, d, d, PK_ACT_NONE, nullptr));
        PK_TRY(end_and_next_norm(L.cross.y, z, L.cross.z, L.cross.drop_offset, L.ffn.ln_g, L.ffn.ln_b, L.ffn.ln_out, L.ffn.mean,
                                 L.ffn.rstd));
        z = L.cross.z;
    }
    // ---- feed-forward ----
    if (L.ffn.bits)
        PK_TRY(pk_gemm_relu_bits(L.ffn.ln_out, L.ffn.w1, L.ffn.h, L.ffn.b1, L.ffn.bits, rows, f, d, d, d, f, f / 8, 0, 0, 1.f, dt, L.stream));
    else
        PK_TRY(linear(L.ffn.ln_out, L.ffn.w1, L.ffn.b1, L.ffn.h, rows, f, d, L.act, L.ffn.pre));
    PK_TRY(linear(L.ffn.h, L.ffn.w2, L.ffn.b2, L.ffn.y, rows, d, f, PK_ACT_NONE, nullptr));
    return pre_end(L, L.ffn.y, z, L.ffn.z, L.ffn.drop_offset);
}

}  // namespace

extern "C" int pk_layer_fwd(const PkLayer* lp) {
    PK_REQ(lp, "pk_layer_fwd: null layer");
    const PkLayer& L = *lp;
    PK_TRY(check(L));
    if (L.prenorm) return layer_fwd_prenorm(L);
    const long long rows = (long long)L.B * L.T, rows_kv = (long long)L.B * L.S, d = L.d, f = L.f;
    const int hd = L.d / L.heads, dt = L.dtype;
    // ---- self-attention: one packed projection, attention straight on its columns, block end ----
    PK_TRY(pk_gemm(L.x, L.self.w_in, L.self.proj, L.self.b_in, nullptr, nullptr, rows, 3 * d, d, d, d, 3 * d, 0, 0, 0, 0, PK_ACT_NONE, 0,
                   1.f, dt, 1, nullptr, 0, nullptr, L.stream));
    PK_TRY(pk_attn_fwd(L.self.proj, at(L.self.proj, d, dt), at(L.self.proj, 2 * d, dt), L.self.attn, L.self.lse,
                       L.is_decoder ? nullptr : L.self_pad, L.B, L.heads, L.T, L.T, hd, (long long)L.T * 3 * d, 3 * d,
                       (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * d, d,
                       L.is_decoder && L.T > 1, L.attn_scale, 0.f, 0, 0, nullptr, dt, L.stream));
    PK_TRY(block_end(L, L.self.attn, d, L.self.w_o, L.self.b_o, L.x, L.self.ln_g, L.self.ln_b, L.self.z, L.self.y, L.self.mean,
                     L.self.rstd, L.self.drop_offset));
    const void* y = L.self.y;
    // ---- cross-attention (decoder): q from the block input, k|v from the encoder output ----
    if (L.is_decoder) {
        PK_TRY(pk_gemm(y, L.cross.w_in, L.cross.proj, L.cross.b_in, nullptr, nullptr, rows, d, d, d, d, d, 0, 0, 0, 0, PK_ACT_NONE, 0, 1.f,
                       dt, 1, nullptr, 0, nullptr, L.stream));
        PK_TRY(pk_gemm(L.enc, at(L.cross.w_in, d * d, dt), L.cross.kv, L.cross.b_in ? at(L.cross.b_in, d, dt) : nullptr, nullptr,
                       nullptr, rows_kv, 2 * d, d, d, d, 2 * d, 0, 0, 0, 0, PK_ACT_NONE, 0, 1.f, dt, 1, nullptr, 0, nullptr, L.stream));
        PK_TRY(pk_attn_fwd(L.cross.proj, L.cross.kv, at(L.cross.kv, d, dt), L.cross.attn, L.cross.lse, L.cross_pad, L.B, L.heads,
                           L.T, L.S, hd, (long long)L.T * d, d, (long long)L.S * 2 * d, 2 * d, (long long)L.S * 2 * d, 2 * d,
                           (long long)L.T * d, d, 0, L.attn_scale, 0.f, 0, 0, nullptr, dt, L.stream));
        PK_TRY(block_end(L, L.cross.attn, d, L.cross.w_o, L.cross.b_o, y, L.cross.ln_g, L.cross.ln_b, L.cross.z, L.cross.y,
                         L.cross.mean, L.cross.rstd, L.cross.drop_offset));
        y = L.cross.y;
    }
    // ---- feed-forward (ReLU at base width: the mask also leaves as one bit per element, for the dH GEMM of backward) ----
    if (L.ffn.bits)
        PK_TRY(pk_gemm_relu_bits(y, L.ffn.w1, L.ffn.h, L.ffn.b1, L.ffn.bits, rows, f, d, d, d, f, f / 8, 0, 0, 1.f, dt, L.stream));
    else
        PK_TRY(pk_gemm(y, L.ffn.w1, L.ffn.h, L.ffn.b1, nullptr, L.ffn.pre, rows, f, d, d, d, f, 0, f, 0, 0, L.act, 0, 1.f, dt, 1, nullptr,
                       0, nullptr, L.stream));
    return block_end(L, L.ffn.h, f, L.ffn.w2, L.ffn.b2, y, L.ffn.ln_g, L.ffn.ln_b, L.ffn.z, L.ffn.y, L.ffn.mean, L.ffn.rstd,
                     L.ffn.drop_offset);
}

// bytes of workspace pk_layer_fwd can use (PkLayer::ws / ws_bytes; without it the forward GEMMs run unsplit): 0 = none
extern "C" size_t pk_layer_fwd_ws(const PkLayer* lp) {
    if (!lp || !lp->prenorm || lp->dtype == PK_F32) return 0;
    const long long rows = (long long)lp->B * lp->T;
    return splitk_ws(fwd_split(rows, lp->d, lp->f), rows, lp->d);
}

extern "C" int pk_layer_bwd_sizes(const PkLayer* lp, size_t* scratch_bytes, size_t* ws_bytes) {
    PK_REQ(lp && scratch_bytes && ws_bytes, "pk_layer_bwd_sizes: null argument");
    PK_TRY(check(*lp));
    Bwd b;
    PK_TRY(plan_bwd(*lp, nullptr, &b));
    // the grouped launch's workspace depends only on the problems' shapes: placeholder addresses do
    PkLayer L = *lp;
    PkWgradProblem probs[PK_WGRAD_MAX];
    const int n = wgrad_problems(L, b, probs);
    b.ws_group = (pk_gemm_wgrad_group_workspace(probs, n) + 255) & ~(size_t)255;
    *scratch_bytes = b.scratch_bytes;
    *ws_bytes = std::max(b.ws_split, b.ws_group) + (L.is_decoder ? 3 : 2) * b.ws_ln;  // (one partial-sum area per LayerNorm)
    return 0;
}

extern "C" int pk_layer_bwd(const PkLayer* lp) {
    PK_REQ(lp, "pk_layer_bwd: null layer");
    const PkLayer& L = *lp;
    PK_TRY(check(L));
    PK_REQ(L.dy && L.dx && (!L.is_decoder || L.denc), "pk_layer_bwd: null gradient buffer");
    const long long rows = (long long)L.B * L.T, rows_kv = (long long)L.B * L.S, d = L.d, f = L.f;
    const int hd = L.d / L.heads, dt = L.dtype;
    const bool drop = L.drop_p > 0.f;
    Bwd b;
    PK_TRY(plan_bwd(L, (char*)L.scratch, &b));
    PkWgradProblem probs[PK_WGRAD_MAX];
    const int nprob = wgrad_problems(L, b, probs);
    b.ws_group = (pk_gemm_wgrad_group_workspace(probs, nprob) + 255) & ~(size_t)255;
    const size_t ws_main = std::max(b.ws_split, b.ws_group);
    PK_REQ(L.scratch && L.scratch_bytes >= b.scratch_bytes, "pk_layer_bwd: scratch too small (%zu < %zu)", L.scratch_bytes, b.scratch_bytes);
    const int nln = L.is_decoder ? 3 : 2;
    PK_REQ(L.ws && L.ws_bytes >= ws_main + nln * b.ws_ln, "pk_layer_bwd: workspace too small (%zu < %zu)", L.ws_bytes,
           ws_main + nln * b.ws_ln);
    void* ws = L.ws;
    // the LayerNorms' parameter gradients: every backward pass leaves its partial sums in an area of its own, ONE launch
    // reduces the layer's two or three at the end (as separate launches: three 5 us kernels of 64 workgroups each)
    PkLnParamGrad ln_items[3];
    int n_ln = 0;
    auto ln_ws = [&](void* dg, void* db) -> void* {
        void* area = (char*)L.ws + ws_main + n_ln * b.ws_ln;
        ln_items[n_ln].workspace = area; ln_items[n_ln].dgamma = dg; ln_items[n_ln].dbeta = db;
        ++n_ln;
        return area;
    };
    auto ln_finish = [&]() { return pk_ln_param_grads(ln_items, n_ln, rows, (int)d, dt, L.stream); };
    // LayerNorm + dropout backward of a block end: gradient of the residual branch, (masked, scaled) gradient of the
    // projection's output, parameter gradients
    auto ln_bwd = [&](const void* dy, const void* z, const void* g, const float* mean, const float* rstd, void* dres, void* dsub,
                      void* dg, void* db, unsigned long long offset) {
        return pk_residual_ln_bwd_partials(dy, nullptr, z, g, mean, rstd, dres, drop ? dsub : nullptr, ln_ws(dg, db), b.ws_ln, rows,
                                           (int)d, L.drop_p, L.seed, offset, dt, L.stream);
    };
    // dX = dY W (+ aux) with the per-op path's split rule (`_dx_gemm`)
    auto dx_gemm = [&](const void* dy, const void* w, void* out, const void* aux, long long M, long long N, long long K, bool rule) {
        const int sk = rule ? choose_splitk(M, N, K) : 1;
        return pk_gemm(dy, w, out, nullptr, aux, nullptr, M, N, K, K, N, N, aux ? N : 0, 0, 0, 1, PK_ACT_NONE, aux ? 1 : 0, 1.f, dt, sk,
                       sk > 1 ? ws : nullptr, sk > 1 ? splitk_ws(sk, M, N) : 0, nullptr, L.stream);
    };
    if (L.prenorm) {
        // z = in + dropout(f(LN(in))): the incoming gradient dz goes (i) through the dropout mask into f's backward and
        // (ii), as the residual branch, into the LayerNorm backward kernel's `dz_extra`
        auto undrop = [&](const void* dz, void* out, unsigned long long offset) -> int {
            return drop ? pk_dropout(dz, out, rows * d, L.drop_p, L.seed, offset, dt, L.stream) : 0;
        };
        // d_in = LN_bwd(dln) + dz is the gradient of the block's input = the output z of the block BELOW, whose backward
        // starts by pushing it through that block's dropout mask: the kernel writes that masked copy too (`below`: its
        // buffer, `below_offset`: that block's dropout offset) — no pk_dropout pass between the blocks of a layer
        auto ln_in_bwd = [&](const void* dln, const void* dz, const void* in, const void* g, const float* mean, const float* rstd,
                             void* d_in, void* dg, void* db, void* below, unsigned long long below_offset) {
            const bool emit = drop && below;
            return pk_residual_ln_bwd_partials(dln, dz, in, g, mean, rstd, d_in, emit ? below : nullptr, ln_ws(dg, db), b.ws_ln, rows,
                                               (int)d, emit ? L.drop_p : 0.f, emit ? L.seed : 0, emit ? below_offset : 0, dt, L.stream);
        };
        const void* in_f = L.is_decoder ? L.cross.z : L.self.z;
        // feed-forward
        // (L.dy_masked: the layer above has written dy through this mask already, in its last LayerNorm backward)
        const bool handed = drop && L.dy_masked;
        if (!handed) PK_TRY(undrop(L.dy, b.dsub_f, L.ffn.drop_offset));
        const void* do_f = handed ? L.dy_masked : drop ? b.dsub_f : L.dy;
        if (L.ffn.bits)
            PK_TRY(pk_gemm_relu_bits(do_f, L.ffn.w2, b.dh, nullptr, L.ffn.bits, rows, f, d, d, f, f, f / 8, 1, 2, 1.f, dt, L.stream));
        else if (L.act == PK_ACT_NONE)
            PK_TRY(pk_gemm(do_f, L.ffn.w2, b.dh, nullptr, nullptr, nullptr, rows, f, d, d, f, f, 0, 0, 0, 1, PK_ACT_NONE, 0, 1.f, dt, 1, nullptr,
                           0, nullptr, L.stream));
        else
            PK_TRY(pk_gemm(do_f, L.ffn.w2, b.dh, nullptr, L.ffn.pre ? L.ffn.pre : L.ffn.h, nullptr, rows, f, d, d, f, f, f, 0, 0, 1, L.act, 2,
                           1.f, dt, 1, nullptr, 0, nullptr, L.stream));
        PK_TRY(dx_gemm(b.dh, L.ffn.w1, b.dln, nullptr, rows, d, f, true));
        PK_TRY(ln_in_bwd(b.dln, L.dy, in_f, L.ffn.ln_g, L.ffn.mean, L.ffn.rstd, b.dy_mid, L.ffn.dln_g, L.ffn.dln_b,
                         L.is_decoder ? b.dsub_c : b.dsub_s, L.is_decoder ? L.cross.drop_offset : L.self.drop_offset));
        const void* dz = b.dy_mid;
        PkWgradProblem pr[PK_WGRAD_MAX];
        int n = wgrad_problems(L, b, pr);
        pr[0].A = do_f;  // (fc2's dY is the un-dropped gradient, wherever it lives)
        if (L.is_decoder) {
            // (b.dsub_c = dz through the cross block's mask: written by the feed-forward block's LayerNorm backward above)
            const void* do_c = drop ? b.dsub_c : dz;
            pr[2].A = do_c;
            PK_TRY(dx_gemm(do_c, L.cross.w_o, b.dcattn, nullptr, rows, d, d, true));
            PK_TRY(pk_attn_bwd(L.cross.proj, L.cross.kv, at(L.cross.kv, d, dt), L.cross.attn, b.dcattn, L.cross.lse, b.delta, b.dq, b.dkv,
                               at(b.dkv, d, dt), L.cross_pad, L.B, L.heads, L.T, L.S, hd, (long long)L.T * d, d, (long long)L.S * 2 * d,
                               2 * d, (long long)L.S * 2 * d, 2 * d, (long long)L.T * d, d, (long long)L.T * d, d, (long long)L.T * d, d,
                               (long long)L.S * 2 * d, 2 * d, (long long)L.S * 2 * d, 2 * d, 0, L.attn_scale, 0.f, nullptr, dt, L.stream));
            PK_TRY(dx_gemm(b.dkv, at(L.cross.w_in, d * d, dt), L.denc, L.denc_prev, rows_kv, d, 2 * d, true));
            PK_TRY(dx_gemm(b.dq, L.cross.w_in, b.dln, nullptr, rows, d, d, true));
            PK_TRY(ln_in_bwd(b.dln, dz, L.self.z, L.cross.ln_g, L.cross.mean, L.cross.rstd, b.dy_self, L.cross.dln_g, L.cross.dln_b,
                             b.dsub_s, L.self.drop_offset));
            dz = b.dy_self;
        }
        // (b.dsub_s: written by the LayerNorm backward of the block above)
        const void* do_s = drop ? b.dsub_s : dz;
        pr[L.is_decoder ? 5 : 2].A = do_s;
        PK_TRY(dx_gemm(do_s, L.self.w_o, b.dattn, nullptr, rows, d, d, true));
        PK_TRY(pk_attn_bwd(L.self.proj, at(L.self.proj, d, dt), at(L.self.proj, 2 * d, dt), L.self.attn, b.dattn, L.self.lse, b.delta, b.dproj,
                           at(b.dproj, d, dt), at(b.dproj, 2 * d, dt), L.is_decoder ? nullptr : L.self_pad, L.B, L.heads, L.T, L.T, hd,
                           (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * d, d,
                           (long long)L.T * d, d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d,
                           L.is_decoder && L.T > 1, L.attn_scale, 0.f, nullptr, dt, L.stream));
        PK_TRY(dx_gemm(b.dproj, L.self.w_in, b.dln, nullptr, rows, d, 3 * d, true));
        // (L.dx_masked: the layer below is a pre-norm layer with the same dropout: its feed-forward mask applied here)
        PK_TRY(ln_in_bwd(b.dln, dz, L.x, L.self.ln_g, L.self.mean, L.self.rstd, L.dx, L.self.dln_g, L.self.dln_b, L.dx_masked,
                         L.dx_mask_offset));
        PK_TRY(ln_finish());
        return pk_gemm_wgrad_group(pr, n, dt, ws, b.ws_group, L.stream);
    }
    // ---- feed-forward block ----
    PK_TRY(ln_bwd(L.dy, L.ffn.z, L.ffn.ln_g, L.ffn.mean, L.ffn.rstd, b.dres_f, b.dsub_f, L.ffn.dln_g, L.ffn.dln_b, L.ffn.drop_offset));
    if (L.ffn.bits)
        PK_TRY(pk_gemm_relu_bits(b.dsub_f, L.ffn.w2, b.dh, nullptr, L.ffn.bits, rows, f, d, d, f, f, f / 8, 1, 2, 1.f, dt, L.stream));
    else if (L.act == PK_ACT_NONE)
        PK_TRY(pk_gemm(b.dsub_f, L.ffn.w2, b.dh, nullptr, nullptr, nullptr, rows, f, d, d, f, f, 0, 0, 0, 1, PK_ACT_NONE, 0, 1.f, dt, 1,
                       nullptr, 0, nullptr, L.stream));
    else  // dH = (dZ W2) * act'(.)
        PK_TRY(pk_gemm(b.dsub_f, L.ffn.w2, b.dh, nullptr, L.ffn.pre ? L.ffn.pre : L.ffn.h, nullptr, rows, f, d, d, f, f, f, 0, 0, 1, L.act, 2,
                       1.f, dt, 1, nullptr, 0, nullptr, L.stream));
    // gradient of the block input: dH W1 + the residual branch (fused block end: no split rule, as FFNResidualLnFn)
    PK_TRY(dx_gemm(b.dh, L.ffn.w1, b.dy_mid, b.dres_f, rows, d, f, !L.fused_tail));
    const void* dy_blk = b.dy_mid;
    // ---- cross-attention block (decoder) ----
    if (L.is_decoder) {
        PK_TRY(ln_bwd(dy_blk, L.cross.z, L.cross.ln_g, L.cross.mean, L.cross.rstd, b.dres_c, b.dsub_c, L.cross.dln_g, L.cross.dln_b,
                      L.cross.drop_offset));
        PK_TRY(dx_gemm(b.dsub_c, L.cross.w_o, b.dcattn, nullptr, rows, d, d, true));
        PK_TRY(pk_attn_bwd(L.cross.proj, L.cross.kv, at(L.cross.kv, d, dt), L.cross.attn, b.dcattn, L.cross.lse, b.delta, b.dq, b.dkv,
                           at(b.dkv, d, dt), L.cross_pad, L.B, L.heads, L.T, L.S, hd, (long long)L.T * d, d, (long long)L.S * 2 * d,
                           2 * d, (long long)L.S * 2 * d, 2 * d, (long long)L.T * d, d, (long long)L.T * d, d, (long long)L.T * d, d,
                           (long long)L.S * 2 * d, 2 * d, (long long)L.S * 2 * d, 2 * d, 0, L.attn_scale, 0.f, nullptr, dt, L.stream));
        PK_TRY(dx_gemm(b.dkv, at(L.cross.w_in, d * d, dt), L.denc, L.denc_prev, rows_kv, d, 2 * d, true));
        PK_TRY(dx_gemm(b.dq, L.cross.w_in, b.dy_self, b.dres_c, rows, d, d, true));
        dy_blk = b.dy_self;
    }
    // ---- self-attention block ----
    PK_TRY(ln_bwd(dy_blk, L.self.z, L.self.ln_g, L.self.mean, L.self.rstd, b.dres_s, b.dsub_s, L.self.dln_g, L.self.dln_b,
                  L.self.drop_offset));
    PK_TRY(dx_gemm(b.dsub_s, L.self.w_o, b.dattn, nullptr, rows, d, d, true));
    PK_TRY(pk_attn_bwd(L.self.proj, at(L.self.proj, d, dt), at(L.self.proj, 2 * d, dt), L.self.attn, b.dattn, L.self.lse, b.delta, b.dproj,
                       at(b.dproj, d, dt), at(b.dproj, 2 * d, dt), L.is_decoder ? nullptr : L.self_pad, L.B, L.heads, L.T, L.T, hd,
                       (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * d, d,
                       (long long)L.T * d, d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d, (long long)L.T * 3 * d, 3 * d,
                       L.is_decoder && L.T > 1, L.attn_scale, 0.f, nullptr, dt, L.stream));
    PK_TRY(dx_gemm(b.dproj, L.self.w_in, L.dx, b.dres_s, rows, d, 3 * d, true));
    PK_TRY(ln_finish());
    // ---- every weight gradient of the layer in one grouped launch ----
    return pk_gemm_wgrad_group(probs, nprob, dt, ws, b.ws_group, L.stream);
}
