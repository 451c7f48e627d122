// HBM-bound kernels at the two ends of the Transformer hot path:
//   K1  token embedding * sqrt(d) + positional embedding (+dropout), and its scatter-add backward
//         pasero/models/modules.py:916-933 (Embedding.forward), :435-457 / :467-484 (positions),
//         pasero/models/transformer.py:727-744, :866-878
//   K6  label-smoothed cross-entropy over a (rows, V) chunk of logits, loss and dlogits in one launch
//         pasero/models/transformer.py:354-380 (compute_loss; F.cross_entropy sum-reduced, ignore_index = pad)
//   plus the small reductions / elementwise helpers the autograd glue needs (column sums for bias gradients,
//   dropout, multiply by a device-resident scalar).
// All are written for coalesced 16-byte accesses, wave-shuffle reductions and >= 1024 workgroups per launch.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// embedding
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, const T* __restrict__ E,
                                                        const T* __restrict__ pos, T* __restrict__ out,
                                                        long long ntok, int Tlen, int d, long long V, float scale,
                                                        int pos_start, unsigned thr, float drop_scale,
                                                        unsigned long long seed, unsigned long long offset) {
    constexpr int EPV = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = d / EPV;
    for (long long tok = (long long)blockIdx.x * 4 + wave; tok < ntok; tok += (long long)gridDim.x * 4) {
        long long id = ids ? ids[tok] : tok;  // ids == NULL: E is a dense (ntok, d) input (speech features path)
        if (ids) id = id < 0 ? 0 : (id >= V ? V - 1 : id);  // modules.py:923 clip(min=0); upper clamp: stay in bounds
        const T* erow = E + id * d;
        const T* prow = pos ? pos + (long long)(pos_start + (int)(tok % Tlen)) * d : nullptr;
        for (int ch = lane; ch < nchunks; ch += 64) {
            Vec16<T> ev = load16<T>(erow + ch * EPV), pv;
            float of[EPV];
            if (prow) pv = load16<T>(prow + ch * EPV);
            long long off = tok * d + (long long)ch * EPV;
            bool keep[EPV];
            if (thr) {
                dropout_keep_chunk<EPV>(seed, offset, (unsigned long long)off, thr, keep);
            }
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float x = ev.get(e) * scale;
                if (prow) x += pv.get(e);
                if (thr) x = keep[e] ? x * drop_scale : 0.f;
                of[e] = x;
            }
            store16<T>(out + off, vec16_pack<T>(of));
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// label-smoothed cross entropy: one workgroup per row
// ------------------------------------------------------------------------------------------------------
// e^(a - m).  16-bit kernels: one fma and v_exp_f32, `nml` = -m log2(e) computed once per chunk / row (as `__expf(a - m)` it
// is a subtraction, a multiplication and the exponential: the kernels below are bound by their instruction count — 17
// issue slots per logit over the two passes against 366 us of memory time at 8192 x 70 376).  The fp32 kernel (the
// parity path) keeps the plain expression.
template <typename T> __device__ __forceinline__ float exp_sub(float a, float m, float nml) {
    if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(fmaf(a, 1.4426950408889634f, nml));
    else return __expf(a - m);
}
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

// THREADS = 256: the general two-pass kernel (eight workgroups per CU).  THREADS = 1024 with `extern` LDS padding that leaves ONE
// workgroup per CU: rows too wide for the register-resident kernel (NLLB's 256 206 columns = 512 KB per row).  With eight
// small workgroups per CU the rows in flight are 256 x 8 x 512 KB = 1 GB and the second pass re-reads its row from HBM;
// with one wide workgroup per CU they are 128 MB, inside the 256 MiB Infinity Cache, and the second read stays on the die.
template <typename T, int THREADS, int PAD_FLOATS = 0>
__global__ __launch_bounds__(THREADS) void ce_kernel(const T* __restrict__ logits, long long ld,
                                                     const long long* __restrict__ target, T* __restrict__ dlogits,
                                                     long long ldd, float* __restrict__ row_loss,
                                                     float* __restrict__ row_nll, float* __restrict__ row_lse, long long V,
                                                     long long pad_idx, float eps, bool vec_ok) {
    constexpr int EPV = 16 / sizeof(T), NW = THREADS / 64;
    // (PAD_FLOATS: LDS nobody uses, declared so that fewer workgroups fit a CU — the statistics live at its start, which keeps
    // the whole array allocated)
    __shared__ float red_all[3 * NW + 2 + PAD_FLOATS];
    float* red_m = red_all, *red_s = red_all + NW, *red_t = red_all + 2 * NW, *bc = red_all + 3 * NW;
    const long long row = blockIdx.x;
    const T* x = logits + row * ld;
    const long long tgt = target[row];
    const bool active = tgt != pad_idx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float lse = 0.f;
    // dlogits may alias logits (the fused vocabulary loss works in place): everything a thread reads from x after the
    // workgroup barrier must be its OWN chunk.  The target logit is therefore fetched here, before any store — read
    // after the barrier it raced with the wave that overwrites that column with its gradient (seen as one row's nll
    // off by several nats, once in ~30 runs).
    float x_tgt = 0.f;
    if (active && tid == 0) {
        long long tc = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
        x_tgt = to_f32<T>(x[tc]);
    }
    if (active || row_lse) {
        float m = -INFINITY, s = 0.f, tot = 0.f;
        const long long nvec = vec_ok ? V / EPV : 0;
        // four 16-byte loads in flight per thread (one per iteration left the pass latency-bound: 3.7 TB/s at V = 70 376)
        auto take = [&](const Vec16<T>& v) {
            float cm = v.get(0);
#pragma unroll
            for (int e = 1; e < EPV; ++e) cm = fmaxf(cm, v.get(e));
            const float mn = fmaxf(m, cm);  // one rescale of the running sum per chunk, the chunk's terms against the new maximum
            const float ms = mn == -INFINITY ? 0.f : mn;  // (nothing but -inf so far: every term is exp(-inf) = 0)
            float cs = 0.f;
            const float nml = -ms * 1.4426950408889634f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float a = v.get(e);
                cs += exp_sub<T>(a, ms, nml);
                tot += a;
            }
            s = s * __expf(m - ms) + cs;
            m = mn;
        };
        long long ch = tid;
        for (; ch + 3 * THREADS < nvec; ch += 4 * THREADS) {
            const Vec16<T> v0 = load16<T>(x + ch * EPV), v1 = load16<T>(x + (ch + THREADS) * EPV);
            const Vec16<T> v2 = load16<T>(x + (ch + 2 * THREADS) * EPV), v3 = load16<T>(x + (ch + 3 * THREADS) * EPV);
            take(v0); take(v1); take(v2); take(v3);
        }
        for (; ch < nvec; ch += THREADS) take(load16<T>(x + ch * EPV));
        for (long long c = nvec * EPV + tid; c < V; c += THREADS) {
            float a = to_f32<T>(x[c]);
            tot += a;
            online_merge(m, s, a, 1.f);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            online_merge(m, s, m2, s2);
            tot += __shfl_xor(tot, o, 64);
        }
        if (lane == 0) { red_m[wave] = m; red_s[wave] = s; red_t[wave] = tot; }
        __syncthreads();
        if (tid == 0) {
            float M = red_m[0], S = red_s[0], Tt = red_t[0];
            for (int w = 1; w < NW; ++w) { online_merge(M, S, red_m[w], red_s[w]); Tt += red_t[w]; }
            bc[0] = M + __logf(S);
            bc[1] = Tt;
        }
        __syncthreads();
        lse = bc[0];
        if (tid == 0) {
            if (row_lse) row_lse[row] = lse;
            if (active) {
                float nll = lse - x_tgt;
                float smooth = lse - bc[1] / (float)V;
                row_nll[row] = nll;
                row_loss[row] = eps > 0.f ? (1.f - eps) * nll + eps * smooth : nll;
            }
        }
    }
    if (!active && tid == 0) { row_loss[row] = 0.f; row_nll[row] = 0.f; }
    if (!dlogits) return;
    T* dx = dlogits + row * ldd;
    const float uni = eps / (float)V;
    const long long nvec = vec_ok ? V / EPV : 0;
    const long long tch = (active && tgt >= 0) ? tgt / EPV : -1;  // the one chunk that holds the target column
    const int te = (int)(tgt >= 0 ? tgt % EPV : 0);
    const float onehot = 1.f - eps;
    const float nlse = -lse * 1.4426950408889634f;
    auto grad = [&](const Vec16<T>& v, long long ch) {
        float g[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) g[e] = exp_sub<T>(v.get(e), lse, nlse) - uni;
        if (ch == tch) {  // (one chunk of the row — a real branch: if-converted it is EPV selects in every chunk)
            int te_v = te;
            asm volatile("; target chunk" : "+v"(te_v));
#pragma unroll
            for (int e = 0; e < EPV; ++e)
                if (e == te_v) g[e] -= onehot;
        }
        store16<T>(dx + ch * EPV, vec16_pack<T>(g));
    };
    if (active) {
        long long ch = tid;
        for (; ch + 3 * THREADS < nvec; ch += 4 * THREADS) {
            const Vec16<T> v0 = load16<T>(x + ch * EPV), v1 = load16<T>(x + (ch + THREADS) * EPV);
            const Vec16<T> v2 = load16<T>(x + (ch + 2 * THREADS) * EPV), v3 = load16<T>(x + (ch + 3 * THREADS) * EPV);
            grad(v0, ch); grad(v1, ch + THREADS); grad(v2, ch + 2 * THREADS); grad(v3, ch + 3 * THREADS);
        }
        for (; ch < nvec; ch += THREADS) grad(load16<T>(x + ch * EPV), ch);
    } else {
        Vec16<T> o;
        o.raw = {0, 0, 0, 0};
        for (long long ch = tid; ch < nvec; ch += THREADS) store16<T>(dx + ch * EPV, o);
    }
    for (long long c = nvec * EPV + tid; c < V; c += THREADS) {
        float g = active ? __expf(to_f32<T>(x[c]) - lse) - uni - (c == tgt ? 1.f - eps : 0.f) : 0.f;
        dx[c] = from_f32<T>(g);
    }
    // a row pitch that leaves room for V rounded up to 8: the pad columns become ZERO, so that the gradient GEMM that
    // contracts over the vocabulary may read whole 16-byte chunks (pk_gemm_ex, PK_GEMM_PAD_K)
    const long long vpad = (V + 7) & ~7LL;
    if (tid < vpad - V && ldd >= vpad) dx[V + tid] = from_f32<T>(0.f);
}

// sums[0] = sum(row_loss), sums[1] = sum(row_nll), sums[2] = #(target != pad); one workgroup, fp64 accumulation
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ row_loss,
                                                           const float* __restrict__ row_nll,
                                                           const long long* __restrict__ target, long long rows,
                                                           long long pad_idx, float* __restrict__ sums) {
    __shared__ double sh[3][16];
    double a = 0, b = 0, c = 0;
    for (long long i = threadIdx.x; i < rows; i += 1024) {
        a += row_loss[i];
        b += row_nll[i];
        c += target[i] != pad_idx ? 1.0 : 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
        c += __shfl_xor(c, o, 64);
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; sh[2][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double A = 0, Bv = 0, C = 0;
        for (int w = 0; w < 16; ++w) { A += sh[0][w]; Bv += sh[1][w]; C += sh[2][w]; }
        sums[0] = (float)A; sums[1] = (float)Bv; sums[2] = (float)C;
    }
}

// ------------------------------------------------------------------------------------------------------
// column sums (bias gradients, learned-position gradients): out[n] = sum_m x[m][n]
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long long ld, long long M,
                                                             long long N, float* __restrict__ partials, bool vec_ok) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[4][64 * EPV];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + tx) * EPV;
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    if (col0 < N) {
        const bool full = vec_ok && col0 + EPV <= N;
        for (long long r = (long long)blockIdx.y * 4 + ty; r < M; r += (long long)gridDim.y * 4) {
            if (full) {
                Vec16<T> v = load16<T>(x + r * ld + col0);
#pragma unroll
                for (int e = 0; e < EPV; ++e) acc[e] += v.get(e);
            } else {
#pragma unroll
                for (int e = 0; e < EPV; ++e)
                    if (col0 + e < N) acc[e] += to_f32<T>(x[r * ld + col0 + e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) red[ty][tx * EPV + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * EPV; c += 256) {
        long long col = (long long)blockIdx.x * 64 * EPV + c;
        if (col < N) partials[(long long)blockIdx.y * N + col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partials, T* __restrict__ out,
                                                           int nparts, long long N) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long col = (long long)blockIdx.x * 64 + tx;
    float s = 0.f;
    if (col < N)
        for (int p = ty; p < nparts; p += 4) s += partials[(long long)p * N + col];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && col < N) out[col] = from_f32<T>(red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// ------------------------------------------------------------------------------------------------------
// elementwise
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ out, long long n,
                                                      unsigned thr, float drop_scale, unsigned long long seed,
                                                      unsigned long long offset) {
    constexpr int EPV = 16 / sizeof(T);
    const long long nvec = n / EPV;
    for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < nvec; ch += (long long)gridDim.x * 256) {
        Vec16<T> v = load16<T>(x + ch * EPV);
        bool keep[EPV];
        dropout_keep_chunk<EPV>(seed, offset, (unsigned long long)(ch * EPV), thr, keep);
        float of[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) of[e] = keep[e] ? v.get(e) * drop_scale : 0.f;
        store16<T>(out + ch * EPV, vec16_pack<T>(of));
    }
    if (blockIdx.x == 0) {
        for (long long i = nvec * EPV + threadIdx.x; i < n; i += 256) {
            out[i] = from_f32<T>(dropout_keep1(seed, offset, (unsigned long long)i, thr) ? to_f32<T>(x[i]) * drop_scale : 0.f);
        }
    }
}

// out = x * (*dev_scalar) * host_scalar
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const T* __restrict__ x, T* __restrict__ out, long long n,
                                                    const float* __restrict__ dev_scalar, float host_scalar) {
    constexpr int EPV = 16 / sizeof(T);
    const float a = (dev_scalar ? *dev_scalar : 1.f) * host_scalar;
    const long long nvec = n / EPV;
    for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < nvec; ch += (long long)gridDim.x * 256) {
        Vec16<T> v = load16<T>(x + ch * EPV);
        float of[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) of[e] = v.get(e) * a;
        store16<T>(out + ch * EPV, vec16_pack<T>(of));
    }
    if (blockIdx.x == 0)
        for (long long i = nvec * EPV + threadIdx.x; i < n; i += 256) out[i] = from_f32<T>(to_f32<T>(x[i]) * a);
}

inline int grid_for(long long work_items, int per_block, int cap = 2048) {
    long long b = (work_items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline bool is_aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace


extern "C" int pk_embed_fwd(const long long* ids, const void* E, const void* pos, void* out, long long ntok, int Tlen,
                            int d, long long V, float scale, int pos_start, float drop_p, unsigned long long seed,
                            unsigned long long offset, int dtype, void* stream) {
    PK_CHECK_ARG(E && out, "pk_embed_fwd: null tensor");
    PK_CHECK_ARG(Tlen > 0 && V > 0, "pk_embed_fwd: bad sizes");
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_embed_fwd: bad dropout %f", drop_p);
    if (ntok == 0) return 0;
    unsigned thr = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    PK_DTYPE_SWITCH(dtype, "pk_embed_fwd", {
        constexpr int EPV = 16 / sizeof(T);
        PK_CHECK_ARG(d % EPV == 0, "pk_embed_fwd: d=%d must be a multiple of %d", d, EPV);
        hipLaunchKernelGGL((embed_fwd_kernel<T>), dim3(grid_for(ntok, 4)), dim3(256), 0, (hipStream_t)stream, ids,
                           (const T*)E, (const T*)pos, (T*)out, ntok, Tlen, d, V, scale, pos_start, thr, ds, seed,
                           offset);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

// (pk_embed_bwd: embed_bwd.hip)

// Label-smoothed CE of `rows` logit rows.  row_loss/row_nll [rows] fp32 outputs (0 for pad rows); row_lse optional;
// dlogits optional (may alias logits): d(loss_sum)/dlogits.
// The same with the row held in registers between the two passes (NV 16-byte vectors per thread): the logits cross the
// fabric once each way.  The two-pass kernel above re-reads the row for the gradient, and at vocabulary widths of 70 k - 256 k
// the rows in flight on an XCD (8 workgroups x 32 CUs x 140 - 512 KB) are far beyond its 4 MiB of L2, so the second read came
// from memory again: 3 transfers per element at ~5.9 TB/s instead of 2.  Everything is loaded before the first store, so
// dlogits may alias logits here as well.
template <typename T, int NV, int THREADS>
__global__ __launch_bounds__(THREADS) void ce_reg_kernel(const T* __restrict__ logits, long long ld,
                                                         const long long* __restrict__ target, T* __restrict__ dlogits,
                                                         long long ldd, float* __restrict__ row_loss,
                                                         float* __restrict__ row_nll, float* __restrict__ row_lse, long long V,
                                                         long long pad_idx, float eps) {
    constexpr int EPV = 16 / sizeof(T), NW = THREADS / 64;
    __shared__ float red_m[NW], red_s[NW], red_t[NW];
    __shared__ float bc[2];
    const long long row = blockIdx.x;
    const T* x = logits + row * ld;
    const long long tgt = target[row];
    const bool active = tgt != pad_idx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nvec = (int)(V / EPV);
    float x_tgt = 0.f;
    if (active && tid == 0) {
        long long tc = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
        x_tgt = to_f32<T>(x[tc]);
    }
    if (!active && !row_lse) {  // padding position: zero gradient, nothing to read
        if (tid == 0) { row_loss[row] = 0.f; row_nll[row] = 0.f; }
        if (dlogits) {
            T* dx = dlogits + row * ldd;
            Vec16<T> o;
            o.raw = {0, 0, 0, 0};
            for (int ch = tid; ch < nvec; ch += THREADS) store16<T>(dx + (long long)ch * EPV, o);
            for (long long c = (long long)nvec * EPV + tid; c < V; c += THREADS) dx[c] = from_f32<T>(0.f);
            const long long vpad = (V + 7) & ~7LL;
            if (tid < vpad - V && ldd >= vpad) dx[V + tid] = from_f32<T>(0.f);
        }
        return;
    }
    // (no branch around a load: slots past the row read its last chunk and are skipped below — behind `if (ch < nvec)` every
    // load is a block of its own and hipcc waits for ALL of them at the first use; now the first chunks are reduced while
    // the later ones are still on their way)
    Vec16<T> v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = min(tid + i * THREADS, nvec - 1);
        v[i] = load16<T>(x + (long long)ch * EPV);
    }
    float m = -INFINITY, s = 0.f, tot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (tid + i * THREADS < nvec) {
            float cm = v[i].get(0);
#pragma unroll
            for (int e = 1; e < EPV; ++e) cm = fmaxf(cm, v[i].get(e));
            const float mn = fmaxf(m, cm);
            const float ms = mn == -INFINITY ? 0.f : mn;
            float cs = 0.f;
            const float nml = -ms * 1.4426950408889634f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float a = v[i].get(e);
                cs += exp_sub<T>(a, ms, nml);
                tot += a;
            }
            s = s * __expf(m - ms) + cs;
            m = mn;
        }
    }
    for (long long c = (long long)nvec * EPV + tid; c < V; c += THREADS) {
        const float a = to_f32<T>(x[c]);
        tot += a;
        online_merge(m, s, a, 1.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
        tot += __shfl_xor(tot, o, 64);
    }
    if (lane == 0) { red_m[wave] = m; red_s[wave] = s; red_t[wave] = tot; }
    // the tail columns (V not a multiple of the vector width) are re-read for the gradient: fetch them before anyone stores
    float tail_a = 0.f;
    const long long tail_c = (long long)nvec * EPV + tid;
    if (tail_c < V) tail_a = to_f32<T>(x[tail_c]);
    __syncthreads();
    if (tid == 0) {
        float M = red_m[0], S = red_s[0], Tt = red_t[0];
        for (int w = 1; w < NW; ++w) { online_merge(M, S, red_m[w], red_s[w]); Tt += red_t[w]; }
        bc[0] = M + __logf(S);
        bc[1] = Tt;
    }
    __syncthreads();
    const float lse = bc[0];
    if (tid == 0) {
        if (row_lse) row_lse[row] = lse;
        if (active) {
            const float nll = lse - x_tgt;
            const float smooth = lse - bc[1] / (float)V;
            row_nll[row] = nll;
            row_loss[row] = eps > 0.f ? (1.f - eps) * nll + eps * smooth : nll;
        } else {
            row_loss[row] = 0.f;
            row_nll[row] = 0.f;
        }
    }
    if (!dlogits) return;
    T* dx = dlogits + row * ldd;
    const float uni = eps / (float)V, onehot = 1.f - eps;
    const long long tch = (active && tgt >= 0) ? tgt / EPV : -1;
    const int te = (int)(tgt >= 0 ? tgt % EPV : 0);
    const float nlse = -lse * 1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = tid + i * THREADS;
        if (ch < nvec) {
            Vec16<T> o;
            if (active) {
                float g[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) g[e] = exp_sub<T>(v[i].get(e), lse, nlse) - uni;
                if (ch == tch) {  // (a real branch, as in ce_kernel)
                    int te_v = te;
                    asm volatile("; target chunk" : "+v"(te_v));
#pragma unroll
                    for (int e = 0; e < EPV; ++e)
                        if (e == te_v) g[e] -= onehot;
                }
                o = vec16_pack<T>(g);
            } else {
                o.raw = {0, 0, 0, 0};
            }
            store16<T>(dx + (long long)ch * EPV, o);
        }
    }
    if (tail_c < V) dx[tail_c] = from_f32<T>(active ? __expf(tail_a - lse) - uni - (tail_c == tgt ? onehot : 0.f) : 0.f);
    const long long vpad = (V + 7) & ~7LL;
    if (tid < vpad - V && ldd >= vpad) dx[V + tid] = from_f32<T>(0.f);
}

// the register-resident form where the row fits (true: launched)
template <typename T>
static bool launch_ce_reg(const T* logits, long long ld, const long long* target, T* dlogits, long long ldd, float* row_loss,
                          float* row_nll, float* row_lse, long long V, long long pad_idx, float eps, long long rows, bool vec_ok,
                          hipStream_t stream) {
    constexpr int EPV = 16 / sizeof(T);
    const long long nvec = V / EPV;
    static const bool no_reg = getenv("PK_CE_NO_REG") != nullptr;  // (diagnostic: the two-pass kernel)
    // (rows wider than 12 vectors x 1024 threads — NLLB's 256 206 — would need 128 data registers per thread at 4 waves per SIMD:
    // measured 853 us against the two-pass kernel's 651 at 2048 rows; they stay there)
    // 16-bit logits only: fp32 is the parity path (its training curve is held to 1e-3 of the reference's over 40 steps — any
    // other summation order of the same sums moves it inside that bar, 1.4e-4 -> 8.8e-4 measured — and it has no speed to gain)
    if (!vec_ok || no_reg || nvec == 0 || nvec > 12 * 1024 || sizeof(T) != 2) return false;
#define PK_CE_REG(NV, TH)                                                                                                  \
    hipLaunchKernelGGL((ce_reg_kernel<T, NV, TH>), dim3((unsigned)rows), dim3(TH), 0, stream, logits, ld, target, dlogits, ldd, \
                       row_loss, row_nll, row_lse, V, pad_idx, eps)
    if (nvec <= 4 * 256) PK_CE_REG(4, 256);
    else if (nvec <= 4 * 1024) PK_CE_REG(4, 1024);
    else PK_CE_REG(12, 1024);
#undef PK_CE_REG
    return true;
}

extern "C" int pk_ce_rows(const void* logits, long long ld, const long long* target, void* dlogits, long long ldd,
                          float* row_loss, float* row_nll, float* row_lse, long long rows, long long V,
                          long long pad_idx, float eps, int dtype, void* stream) {
    if (rows == 0) return 0;
    PK_CHECK_ARG(logits && target && row_loss && row_nll, "pk_ce_rows: null tensor");
    PK_CHECK_ARG(V > 0 && eps >= 0.f && eps < 1.f, "pk_ce_rows: bad V / label smoothing");
    PK_CHECK_ARG(rows < (1ll << 31), "pk_ce_rows: too many rows per call");
    if (rows == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_ce_rows", {
        constexpr int EPV = 16 / sizeof(T);
        bool vec_ok = is_aligned16(logits) && ld % EPV == 0 && (!dlogits || (is_aligned16(dlogits) && ldd % EPV == 0));
        if (!launch_ce_reg<T>((const T*)logits, ld, target, (T*)dlogits, ldd, row_loss, row_nll, row_lse, V, pad_idx, eps, rows,
                              vec_ok, (hipStream_t)stream)) {
            // rows beyond the register-resident kernel's 98 304 columns, 16-bit: one 1024-thread workgroup per CU (the rows in
            // flight then fit the Infinity Cache: see ce_kernel).  PK_CE_WIDE_LDS: bytes of LDS padding = workgroups per CU
            // (default 84 KiB: one workgroup per CU; 0: no padding, two fit — the A/B), PK_CE_WIDE=0: the 256-thread form as before round 5.
            static const bool wide_on = [] { const char* e = getenv("PK_CE_WIDE"); return !e || atoi(e) != 0; }();
            static const int wide_lds = [] { const char* e = getenv("PK_CE_WIDE_LDS"); return e ? atoi(e) : 84 * 1024; }();
            if (wide_on && sizeof(T) == 2 && vec_ok && V / EPV > 12 * 1024) {
                if (wide_lds > 0)
                    hipLaunchKernelGGL((ce_kernel<T, 1024, 84 * 256>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                       (const T*)logits, ld, target, (T*)dlogits, ldd, row_loss, row_nll, row_lse, V, pad_idx,
                                       eps, vec_ok);
                else
                    hipLaunchKernelGGL((ce_kernel<T, 1024>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                       (const T*)logits, ld, target, (T*)dlogits, ldd, row_loss, row_nll, row_lse, V, pad_idx,
                                       eps, vec_ok);
            }
            else
                hipLaunchKernelGGL((ce_kernel<T, 256>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                                   (const T*)logits, ld, target, (T*)dlogits, ldd, row_loss, row_nll, row_lse, V, pad_idx, eps,
                                   vec_ok);
        }
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_ce_finalize(const float* row_loss, const float* row_nll, const long long* target, long long rows,
                              long long pad_idx, float* sums3, void* stream) {
    PK_CHECK_ARG(sums3 && (rows == 0 || (row_loss && row_nll && target)), "pk_ce_finalize: null tensor");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, row_nll, target,
                       rows, pad_idx, sums3);
    PK_LAUNCH_CHECK();
    return 0;
}

// row-slices per column block: enough workgroups (~1024) to stream at HBM rate even for narrow matrices
static long long colsum_parts(long long M, long long N, int epv) {
    long long colblocks = (N + 64 * epv - 1) / (64 * epv);
    long long parts = (1024 + colblocks - 1) / colblocks;
    long long maxparts = (M + 15) / 16;  // >= 4 rows per row-lane
    if (parts > maxparts) parts = maxparts;
    if (parts > 1024) parts = 1024;
    if (parts < 1) parts = 1;
    return parts;
}

extern "C" size_t pk_colsum_workspace(long long M, long long N) {
    long long a = colsum_parts(M, N, 4), b = colsum_parts(M, N, 8);
    return (size_t)(a > b ? a : b) * N * sizeof(float);
}

extern "C" int pk_colsum(const void* x, long long ld, void* out, long long M, long long N, void* workspace,
                         size_t ws_bytes, int dtype, void* stream) {
    PK_CHECK_ARG(x && out, "pk_colsum: null tensor");
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    PK_DTYPE_SWITCH(dtype, "pk_colsum", {
        constexpr int EPV = 16 / sizeof(T);
        const long long parts = colsum_parts(M, N, EPV);
        PK_CHECK_ARG(workspace && ws_bytes >= (size_t)parts * N * 4, "pk_colsum: workspace too small");
        bool vec_ok = is_aligned16(x) && ld % EPV == 0;
        dim3 grid((unsigned)((N + 64 * EPV - 1) / (64 * EPV)), (unsigned)parts);
        hipLaunchKernelGGL((colsum_partial_kernel<T>), grid, dim3(256), 0, s, (const T*)x, ld, M, N,
                           (float*)workspace, vec_ok);
        PK_LAUNCH_CHECK();
        hipLaunchKernelGGL((colsum_final_kernel<T>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s,
                           (const float*)workspace, (T*)out, (int)parts, N);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_dropout(const void* x, void* out, long long n, float drop_p, unsigned long long seed,
                          unsigned long long offset, int dtype, void* stream) {
    PK_CHECK_ARG(x && out, "pk_dropout: null tensor");
    PK_CHECK_ARG(drop_p > 0.f && drop_p < 1.f, "pk_dropout: p must be in (0, 1), got %f", drop_p);
    PK_CHECK_ARG(is_aligned16(x) && is_aligned16(out), "pk_dropout: tensors must be 16-byte aligned");
    if (n == 0) return 0;
    unsigned thr = dropout_threshold(drop_p);
    float ds = 1.f / (1.f - drop_p);
    PK_DTYPE_SWITCH(dtype, "pk_dropout", {
        hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)out, n, thr, ds, seed, offset);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_scale(const void* x, void* out, long long n, const float* dev_scalar, float host_scalar, int dtype,
                        void* stream) {
    PK_CHECK_ARG(x && out, "pk_scale: null tensor");
    PK_CHECK_ARG(is_aligned16(x) && is_aligned16(out), "pk_scale: tensors must be 16-byte aligned");
    if (n == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_scale", {
        hipLaunchKernelGGL((scale_kernel<T>), dim3(grid_for(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)out, n, dev_scalar, host_scalar);
    })
    PK_LAUNCH_CHECK();
    return 0;
}
