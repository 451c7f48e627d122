// Residual + dropout + LayerNorm, forward and backward (K4 of SURVEY §2c).
//   reference: x = residual + dropout(x); x = LayerNorm(x)      pasero/models/transformer.py:1043-1054,1073-1086
//              nn.LayerNorm(eps=1e-5, affine)                    pasero/models/transformer.py:941-947
// HBM-bound: one wave (64 lanes) owns one row, each lane keeps its 16-byte chunks of the row in registers, so a row
// is read exactly once; mean / variance are wave-shuffle reductions in fp32 (two-pass variance, like aten).
// The dropout mask is regenerated from (seed, offset, element index) in the backward pass, never stored.
// dgamma / dbeta: per-lane running sums over a wave's rows, one LDS reduction per workgroup into the workgroup's own
// [2][d] fp32 slab, then a column-parallel reduction kernel over the slabs (no atomics).
// RMSNorm (pasero/models/modules.py:192-202: y = x * rsqrt(mean(x^2) + eps) * weight, computed in fp32) is the same
// kernels with the mean fixed at 0: the C ABI selects it with mean == NULL (then beta must be NULL too).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // one wave per row, 256 threads

template <typename T, int NCH>
__global__ __launch_bounds__(256) void residual_ln_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ residual, const T* __restrict__ gamma,
    const T* __restrict__ beta, T* __restrict__ z_out, T* __restrict__ y_out, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, long long rows, int d, float eps, unsigned thr, float drop_scale,
    unsigned long long seed, unsigned long long offset) {
    constexpr int EPV = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = d / EPV;
    const float inv_d = 1.f / (float)d;
    for (long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + wave; row < rows;
         row += (long long)gridDim.x * ROWS_PER_BLOCK) {
        float v[NCH][EPV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ch = lane + 64 * i;
            if (ch < nchunks) {
                long long off = row * d + (long long)ch * EPV;
                Vec16<T> xv = load16<T>(x + off);
                bool keep[EPV];
                if (thr) {
                    dropout_keep_chunk<EPV>(seed, offset, (unsigned long long)off, thr, keep);
                }
                Vec16<T> rv;
                if (residual) rv = load16<T>(residual + off);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float a = xv.get(e);
                    if (thr) a = keep[e] ? a * drop_scale : 0.f;
                    if (residual) a += rv.get(e);
                    v[i][e] = a;
                    sum += a;
                }
                if (z_out) store16<T>(z_out + off, vec16_pack<T>(v[i]));
            } else {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v[i][e] = 0.f;
            }
        }
        if (!gamma) continue;  // residual-only mode (pre-norm blocks)
        // z is a tensor of dtype T in the reference (and is what the backward pass re-reads): take the statistics
        // on the rounded values
        if ((residual || thr) && sizeof(T) == 2) {
            sum = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    v[i][e] = H16<std::conditional_t<sizeof(T) == 2, T, bf16>>::val(H16<std::conditional_t<sizeof(T) == 2, T, bf16>>::bits(v[i][e]));
                    sum += v[i][e];
                }
        }
        const bool rms = mean_out == nullptr;
        float mu = rms ? 0.f : wave_sum(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ch = lane + 64 * i;
            if (ch < nchunks) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float c = v[i][e] - mu;
                    sq += c * c;
                }
            }
        }
        float rstd = rsqrtf(wave_sum(sq) * inv_d + eps);
        if (lane == 0) {
            if (!rms) mean_out[row] = mu;
            rstd_out[row] = rstd;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ch = lane + 64 * i;
            if (ch < nchunks) {
                Vec16<T> gv = load16<T>(gamma + ch * EPV), bv;
                if (beta) bv = load16<T>(beta + ch * EPV);
                float yf[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float y = (v[i][e] - mu) * rstd * gv.get(e);
                    if (beta) y += bv.get(e);
                    yf[e] = y;
                }
                store16<T>(y_out + row * d + (long long)ch * EPV, vec16_pack<T>(yf));
            }
        }
    }
}

// dz = LN_bwd(dy) (+ dz_extra);  dres_out = dz;  dx_out = dz * keep * scale;  partial dgamma/dbeta per block
template <typename T, int NCH>
__global__ __launch_bounds__(256, NCH == 4 ? 3 : NCH == 8 ? 2 : 1) void residual_ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ dz_extra, const T* __restrict__ z, const T* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dres_out, T* __restrict__ dx_out,
    float* __restrict__ partials /* [gridDim.x][2][d], every workgroup writes its slab */, long long rows, int d, unsigned thr, float drop_scale,
    unsigned long long seed, unsigned long long offset) {
    constexpr int EPV = 16 / sizeof(T);
    // Wide rows (d > 1024 for 16-bit types): with gamma, the products dy*gamma and the normalised row all held in
    // registers next to the dgamma / dbeta sums, a wave needs 290-512 registers and runs alone on its SIMD (2.0 TB/s
    // at d = 2048, 1.3 at 4096).  The LEAN form keeps only the sums and the packed dy / z chunks: gamma is re-read
    // (L1-resident) and dy*gamma, xhat are recomputed in the second pass; no next-row prefetch, occupancy hides it.
    constexpr bool LEAN = NCH >= 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = d / EPV;
    const float inv_d = 1.f / (float)d;
    float dg[NCH][EPV], db[NCH][EPV], gm[LEAN ? 1 : NCH][EPV];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int ch = lane + 64 * i;
        Vec16<T> gv;
        if (!LEAN && gamma && ch < nchunks) gv = load16<T>(gamma + ch * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            dg[i][e] = 0.f;
            db[i][e] = 0.f;
            if constexpr (!LEAN) gm[i][e] = (gamma && ch < nchunks) ? gv.get(e) : 0.f;
        }
    }
    if constexpr (LEAN) {
        const bool rms_l = mean == nullptr;
        for (long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + wave; row < rows;
             row += (long long)gridDim.x * ROWS_PER_BLOCK) {
            Vec16<T> dvc[NCH], zvc[NCH];
            float mu = 0.f, rs = 0.f, s1 = 0.f, s2 = 0.f;
            if (gamma) {
                mu = rms_l ? 0.f : mean[row];
                rs = rstd[row];
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    int ch = lane + 64 * i;
                    if (ch < nchunks) {
                        long long off = row * d + (long long)ch * EPV;
                        dvc[i] = load16<T>(dy + off);
                        zvc[i] = load16<T>(z + off);
                    }
                }
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    int ch = lane + 64 * i;
                    if (ch < nchunks) {
                        Vec16<T> gv = load16<T>(gamma + ch * EPV);
#pragma unroll
                        for (int e = 0; e < EPV; ++e) {
                            float dyv = dvc[i].get(e);
                            float xhat = (zvc[i].get(e) - mu) * rs;
                            float gg = dyv * gv.get(e);
                            s1 += gg;
                            s2 += gg * xhat;
                            dg[i][e] += dyv * xhat;
                            db[i][e] += dyv;
                        }
                    }
                }
                s1 = rms_l ? 0.f : wave_sum(s1) * inv_d;
                s2 = wave_sum(s2) * inv_d;
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                int ch = lane + 64 * i;
                if (ch < nchunks) {
                    long long off = row * d + (long long)ch * EPV;
                    float dz[EPV];
                    Vec16<T> ev, gv;
                    if (dz_extra) ev = load16<T>(dz_extra + off);
                    if (gamma) gv = load16<T>(gamma + ch * EPV);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        float t = 0.f;
                        if (gamma) {  // same expressions as the first pass, so the two passes agree bit for bit
                            float xhat = (zvc[i].get(e) - mu) * rs;
                            float gg = dvc[i].get(e) * gv.get(e);
                            t = rs * (gg - s1 - xhat * s2);
                        }
                        if (dz_extra) t += ev.get(e);
                        dz[e] = t;
                    }
                    if (dres_out) {
                        Vec16<T> o;
#pragma unroll
                        for (int e = 0; e < EPV; ++e) o.set(e, dz[e]);
                        store16<T>(dres_out + off, o);
                    }
                    if (dx_out) {
                        bool keep[EPV];
                        if (thr) {
                            dropout_keep_chunk<EPV>(seed, offset, (unsigned long long)off, thr, keep);
                        }
                        Vec16<T> o;
#pragma unroll
                        for (int e = 0; e < EPV; ++e) o.set(e, thr ? (keep[e] ? dz[e] * drop_scale : 0.f) : dz[e]);
                        store16<T>(dx_out + off, o);
                    }
                }
            }
        }
    } else {
    // the loads of the wave's next row are issued before the reductions of the current one (a wave walks ~8 rows)
    const long long row_step = (long long)gridDim.x * ROWS_PER_BLOCK;
    Vec16<T> dv_n[NCH], zv_n[NCH];
    float mu_n = 0.f, rs_n = 0.f;
    const bool rms = mean == nullptr;  // RMSNorm: xhat = z * rstd, no mean-of-gradient term
    auto fetch = [&](long long row) {
        mu_n = rms ? 0.f : mean[row];
        rs_n = rstd[row];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ch = lane + 64 * i;
            if (ch < nchunks) {
                long long off = row * d + (long long)ch * EPV;
                dv_n[i] = load16<T>(dy + off);  // (streaming loads here: no gain beside the streaming store of dres)
                zv_n[i] = load16<T>(z + off);
            }
        }
    };
    const long long row_first = (long long)blockIdx.x * ROWS_PER_BLOCK + wave;
    if (gamma && row_first < rows) fetch(row_first);
    for (long long row = row_first; row < rows; row += row_step) {
        float g[NCH][EPV], xh[NCH][EPV];
        float s1 = 0.f, s2 = 0.f;
        float mu = 0.f, rs = 0.f;
        if (gamma) {
            mu = mu_n;
            rs = rs_n;
            Vec16<T> dv_c[NCH], zv_c[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                dv_c[i] = dv_n[i];
                zv_c[i] = zv_n[i];
            }
            if (row + row_step < rows) fetch(row + row_step);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                int ch = lane + 64 * i;
                if (ch < nchunks) {
                    const Vec16<T>&dv = dv_c[i], &zv = zv_c[i];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        float dyv = dv.get(e);
                        float xhat = (zv.get(e) - mu) * rs;
                        float gg = dyv * gm[i][e];
                        g[i][e] = gg;
                        xh[i][e] = xhat;
                        s1 += gg;
                        s2 += gg * xhat;
                        dg[i][e] += dyv * xhat;
                        db[i][e] += dyv;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) g[i][e] = xh[i][e] = 0.f;
                }
            }
            s1 = rms ? 0.f : wave_sum(s1) * inv_d;
            s2 = wave_sum(s2) * inv_d;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int ch = lane + 64 * i;
            if (ch < nchunks) {
                long long off = row * d + (long long)ch * EPV;
                float dz[EPV];
                Vec16<T> ev;
                if (dz_extra) ev = load16<T>(dz_extra + off);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float t = gamma ? rs * (g[i][e] - s1 - xh[i][e] * s2) : 0.f;
                    if (dz_extra) t += ev.get(e);
                    dz[e] = t;
                }
                if (dres_out) {
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) o.set(e, dz[e]);
                    // the residual branch's gradient is read again several kernels later (the block's last dX GEMM): a
                    // streaming store — C2 step 13.84 -> 13.69 ms, same box; the masked gradient below feeds the NEXT kernel,
                    // once per column strip: streamed, the step lost 0.12 ms
                    store16_nt<T>(dres_out + off, o);
                }
                if (dx_out) {
                    bool keep[EPV];
                    if (thr) {
                        dropout_keep_chunk<EPV>(seed, offset, (unsigned long long)off, thr, keep);
                    }
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) o.set(e, thr ? (keep[e] ? dz[e] * drop_scale : 0.f) : dz[e]);
                    store16<T>(dx_out + off, o);
                }
            }
        }
    }
    }  // !LEAN
    if (!partials) return;
    // 4 waves -> one partial per workgroup, written to the workgroup's own slab.  (fp32 atomics into one [2][d] row were
    // the slowest part of this kernel: a thousand workgroups adding to the same 2*d addresses serialise at the memory
    // side — MI355X_MICROARCH "every workgroup into ONE row: 14x slower".)
    __shared__ float red[ROWS_PER_BLOCK][2][64 * EPV];  // per wave, one chunk-column set at a time
    float* slab = partials + (long long)blockIdx.x * 2 * d;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            red[wave][0][lane * EPV + e] = dg[i][e];
            red[wave][1][lane * EPV + e] = db[i][e];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * 64 * EPV; c += 256) {
            int which = c / (64 * EPV), col = c % (64 * EPV);
            int gcol = i * 64 * EPV + col;
            if (gcol < d)
                slab[(long long)which * d + gcol] =
                    red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
        }
        __syncthreads();
    }
}

// ---- the backward kernel for the rows the models train on: 16-bit storage, d = 512 or 1024, LayerNorm with gamma ----
// residual_ln_bwd_kernel above decides everything at run time (gamma? dz_extra? which outputs? RMSNorm? dropout?
// a partial last chunk?) and hipcc turns those decisions into per-element selects and register copies: 480 vector
// instructions per 8-element chunk, of which 80 are moves, 68 selects, 48 the shuffle reductions and 16 64-bit
// address updates.  A wave64 instruction occupies its SIMD for four cycles, a SIMD owns 32 rows of a 32768-row
// tensor: 32 x 480 x 4 cycles = 29 us of instruction issue for a kernel whose 128 MB take 20 us at the copy rate —
// the kernel was bound by its own instruction stream (33 us measured).  Here the decisions are template flags, a row
// is wave-uniform (scalar base address, mean / rstd by scalar loads), the next row's loads go into a second register
// set (loop unrolled twice instead of copying), arithmetic is on float pairs (v_pk_*), and the two row sums are DPP
// butterflies.  FL: 1 = dz_extra, 2 = dx_out with dropout (dres_out is always written).
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <typename T> __device__ __forceinline__ f32x2 unpack2(unsigned w) {
    f32x2 r;
    if constexpr (sizeof(T) == 2 && !__is_same(T, f16)) {
        r.x = __uint_as_float(w << 16);
        r.y = __uint_as_float(w & 0xffff0000u);
    } else {
        r.x = H16<f16>::val((unsigned short)(w & 0xffffu));
        r.y = H16<f16>::val((unsigned short)(w >> 16));
    }
    return r;
}
template <typename T> __device__ __forceinline__ unsigned pack2(f32x2 v) {
    return (unsigned)H16<T>::bits(v.x) | ((unsigned)H16<T>::bits(v.y) << 16);
}

// ---- the forward kernel for the same rows (16-bit storage, d = 512 or 1024) ----
// residual_ln_fwd_kernel above spends 353 (d = 512) / 628 (d = 1024, with scalar registers spilled to lanes) vector
// instructions per row on run-time decisions, per-element selects and one-at-a-time conversions: at 8192 rows x 1024 that
// is 8 rows x 628 x 4 cycles = 9.6 us of issue per SIMD in a 13.6 us kernel whose 33.5 MB take 6 us.  The combinations
// the models use as template flags (FL: 1 = residual, 2 = dropout on x, 4 = LayerNorm — without it the residual-only
// pass of the pre-norm blocks; z is written exactly when 1 or 2 is set; beta is required with 4), a wave-uniform row,
// the next row in a second register set, float pairs, packed conversions.
template <typename T, int NCH, int FL>
__global__ __launch_bounds__(256) void residual_ln_fwd16_kernel(
    const T* __restrict__ x, const T* __restrict__ residual, const T* __restrict__ gamma, const T* __restrict__ beta,
    T* __restrict__ z_out, T* __restrict__ y_out, float* __restrict__ mean_out, float* __restrict__ rstd_out, long long rows,
    float eps, unsigned thr, float drop_scale, unsigned long long seed, unsigned long long offset) {
    constexpr int D = NCH * 512;
    constexpr bool RES = (FL & 1) != 0, DROP = (FL & 2) != 0, LN = (FL & 4) != 0, ZOUT = RES || DROP;
    constexpr float inv_d = 1.f / (float)D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f32x2 gm[NCH][4], bt[NCH][4];
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const uint4 gv = *(reinterpret_cast<const uint4*>(gamma) + lane + 64 * i);
            const uint4 bv = *(reinterpret_cast<const uint4*>(beta) + lane + 64 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gm[i][j] = unpack2<T>((&gv.x)[j]);
                bt[i][j] = unpack2<T>((&bv.x)[j]);
            }
        }
    }
    const long long row_step = (long long)gridDim.x * ROWS_PER_BLOCK;
    uint4 xA[NCH], rA[NCH], xB[NCH], rB[NCH];
    auto fetch = [&](long long row, uint4(&xv)[NCH], uint4(&rv)[NCH]) {
        const uint4* xp = reinterpret_cast<const uint4*>(x + row * D) + lane;
#pragma unroll
        for (int i = 0; i < NCH; ++i) xv[i] = xp[64 * i];
        if constexpr (RES) {
            const uint4* rp = reinterpret_cast<const uint4*>(residual + row * D) + lane;
#pragma unroll
            for (int i = 0; i < NCH; ++i) rv[i] = rp[64 * i];
        }
    };
    auto work = [&](long long row, const uint4(&xv)[NCH], const uint4(&rv)[NCH]) {
        f32x2 v[NCH][4];
        f32x2 a1 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            bool keep[8];
            if constexpr (DROP) dropout_keep8(seed, offset, ((unsigned long long)row * D >> 3) + lane + 64 * i, thr, keep);
            uint4 zw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 a = unpack2<T>((&xv[i].x)[j]);
                if constexpr (DROP) {
                    a.x = keep[2 * j] ? a.x * drop_scale : 0.f;
                    a.y = keep[2 * j + 1] ? a.y * drop_scale : 0.f;
                }
                if constexpr (RES) a += unpack2<T>((&rv[i].x)[j]);
                if constexpr (ZOUT) {
                    // z is a tensor of the storage type in the reference (and what the backward pass re-reads): statistics on
                    // the rounded values
                    (&zw.x)[j] = pack2<T>(a);
                    a = unpack2<T>((&zw.x)[j]);
                }
                v[i][j] = a;
                a1 += a;
            }
            if constexpr (ZOUT) *(reinterpret_cast<uint4*>(z_out + row * D) + lane + 64 * i) = zw;
        }
        if constexpr (LN) {
            const float mu = wave_sum(a1.x + a1.y) * inv_d;
            f32x2 a2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[i][j] -= mu;
                    a2 += v[i][j] * v[i][j];
                }
            const float rstd = rsqrtf(wave_sum(a2.x + a2.y) * inv_d + eps);
            if (lane == 0) {
                mean_out[row] = mu;
                rstd_out[row] = rstd;
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                uint4 yw;
#pragma unroll
                for (int j = 0; j < 4; ++j) (&yw.x)[j] = pack2<T>(v[i][j] * rstd * gm[i][j] + bt[i][j]);
                *(reinterpret_cast<uint4*>(y_out + row * D) + lane + 64 * i) = yw;
            }
        }
    };
    long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + wave;
    if (row < rows) fetch(row, xA, rA);
    while (row < rows) {  // (unconditional prefetch, as in the backward kernel below)
        long long nx = row + row_step;
        fetch(nx < rows ? nx : row, xB, rB);
        work(row, xA, rA);
        row = nx;
        if (row >= rows) break;
        nx = row + row_step;
        fetch(nx < rows ? nx : row, xA, rA);
        work(row, xB, rB);
        row = nx;
    }
}

template <typename T, int NCH, int FL>
__global__ __launch_bounds__(256) void residual_ln_bwd16_kernel(
    const T* __restrict__ dy, const T* __restrict__ dz_extra, const T* __restrict__ z, const T* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dres_out, T* __restrict__ dx_out,
    float* __restrict__ partials, long long rows, unsigned thr, float drop_scale, unsigned long long seed,
    unsigned long long offset) {
    constexpr int D = NCH * 512;
    constexpr bool EXTRA = (FL & 1) != 0, DROPX = (FL & 2) != 0;
    constexpr float inv_d = 1.f / (float)D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f32x2 gm[NCH][4], dg[NCH][4], db[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const uint4 gv = *(reinterpret_cast<const uint4*>(gamma) + lane + 64 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gm[i][j] = unpack2<T>((&gv.x)[j]);
            dg[i][j] = f32x2{0.f, 0.f};
            db[i][j] = f32x2{0.f, 0.f};
        }
    }
    const long long row_step = (long long)gridDim.x * ROWS_PER_BLOCK;
    uint4 dA[NCH], zA[NCH], dB[NCH], zB[NCH];
    float muA = 0.f, rsA = 0.f, muB = 0.f, rsB = 0.f;
    auto fetch = [&](long long row, uint4(&dv)[NCH], uint4(&zv)[NCH], float& mu, float& rs) {
        mu = mean[row];
        rs = rstd[row];
        const uint4* dp = reinterpret_cast<const uint4*>(dy + row * D) + lane;
        const uint4* zp = reinterpret_cast<const uint4*>(z + row * D) + lane;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            dv[i] = dp[64 * i];
            zv[i] = zp[64 * i];
        }
    };
    auto work = [&](long long row, const uint4(&dv)[NCH], const uint4(&zv)[NCH], float mu, float rs) {
        f32x2 g[NCH][4], xh[NCH][4];
        f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
        uint4 ev[NCH];
        if constexpr (EXTRA) {
            const uint4* ep = reinterpret_cast<const uint4*>(dz_extra + row * D) + lane;
#pragma unroll
            for (int i = 0; i < NCH; ++i) ev[i] = ep[64 * i];
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 dyv = unpack2<T>((&dv[i].x)[j]);
                const f32x2 x = (unpack2<T>((&zv[i].x)[j]) - mu) * rs;
                const f32x2 gg = dyv * gm[i][j];
                g[i][j] = gg;
                xh[i][j] = x;
                a1 += gg;
                a2 += gg * x;
                dg[i][j] += dyv * x;
                db[i][j] += dyv;
            }
        const float s1 = wave_sum(a1.x + a1.y) * inv_d;
        const float s2 = wave_sum(a2.x + a2.y) * inv_d;
        uint4* rp = reinterpret_cast<uint4*>(dres_out + row * D) + lane;
        uint4* xp = DROPX ? reinterpret_cast<uint4*>(dx_out + row * D) + lane : nullptr;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            f32x2 t[4];
            uint4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
                if constexpr (EXTRA) t[j] += unpack2<T>((&ev[i].x)[j]);
                (&o.x)[j] = pack2<T>(t[j]);
            }
            // (streaming: the residual branch's gradient is read again several kernels later — see the kernel above)
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(rp + 64 * i));
            if constexpr (DROPX) {
                bool keep[8];
                dropout_keep8(seed, offset, ((unsigned long long)row * D >> 3) + lane + 64 * i, thr, keep);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 m;
                    m.x = keep[2 * j] ? t[j].x * drop_scale : 0.f;
                    m.y = keep[2 * j + 1] ? t[j].y * drop_scale : 0.f;
                    (&o.x)[j] = pack2<T>(m);
                }
                xp[64 * i] = o;
            }
        }
    };
    long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + wave;
    // (the prefetch is unconditional — past the wave's last row it reads that row again: behind a branch the number of
    // loads in flight is unknown at the join and hipcc waits for all of them, the prefetch included)
    if (row < rows) fetch(row, dA, zA, muA, rsA);
    while (row < rows) {
        long long nx = row + row_step;
        fetch(nx < rows ? nx : row, dB, zB, muB, rsB);
        work(row, dA, zA, muA, rsA);
        row = nx;
        if (row >= rows) break;
        nx = row + row_step;
        fetch(nx < rows ? nx : row, dA, zA, muA, rsA);
        work(row, dB, zB, muB, rsB);
        row = nx;
    }
    if (!partials) return;
    // one partial per workgroup, in the slab layout of the kernel above
    __shared__ float red[ROWS_PER_BLOCK][2][512];
    float* slab = partials + (long long)blockIdx.x * 2 * D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x2*>(&red[wave][0][lane * 8 + 2 * j]) = dg[i][j];
            *reinterpret_cast<f32x2*>(&red[wave][1][lane * 8 + 2 * j]) = db[i][j];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * 512; c += 256) {
            const int which = c >> 9, col = c & 511;
            slab[which * D + i * 512 + col] = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
        }
        __syncthreads();
    }
}

// partials[nslabs][2][d] fp32 -> dgamma / dbeta in T.  Workgroup (x, y): 16 columns of gamma (y = 0) or beta (y = 1);
// its 64 thread-rows each sum every 64th slab with all loads in flight at once, then a tree reduction through LDS
// (blockIdx.z: one of up to PK_LN_GROUP_MAX LayerNorms whose backward passes left their slabs — pk_ln_param_grads)
struct LnReduceItems {
    const float* partials[PK_LN_GROUP_MAX];
    void* dgamma[PK_LN_GROUP_MAX];
    void* dbeta[PK_LN_GROUP_MAX];
};
template <typename T>
__global__ __launch_bounds__(1024) void ln_param_grad_kernel(LnReduceItems items, int nslabs, int d) {
    __shared__ float red[64][16];
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c, which = blockIdx.y;
    const float* partials = items.partials[blockIdx.z];
    T* out = reinterpret_cast<T*>(which == 0 ? items.dgamma[blockIdx.z] : items.dbeta[blockIdx.z]);
    if (!out) return;
    float s = 0.f;
    if (col < d) {
        const float* src = partials + (long long)which * d + col;
#pragma unroll 16
        for (int p = r; p < nslabs; p += 64) s += src[(long long)p * 2 * d];
    }
    red[r][c] = s;
    __syncthreads();
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        if (r < k) red[r][c] += red[r + k][c];
        __syncthreads();
    }
    if (r == 0 && col < d) out[col] = from_f32<T>(red[0][c]);
}

// backward grid: every workgroup writes a [2][d] fp32 slab of partial dgamma / dbeta sums that the reduction kernel
// reads back, so the slab traffic grows with blocks * d.  Measured best (32768 * 512 elements, bf16, us incl. the
// reduction kernel): d = 512: 1024 workgroups (27.7 vs 31.2 at 512); d = 1024..4096: 512 (29.6 / 37.4 / 36.0 / 50.9 vs
// 31.9 / 42.0 / 41.5 / 63.4 at 1024).
inline int ln_bwd_max_blocks(int d) {
    static const int forced = getenv("PK_LN_BWD_BLOCKS") ? atoi(getenv("PK_LN_BWD_BLOCKS")) : 0;  // (experiments)
    if (forced > 0) return forced;
    return d <= 512 ? 1024 : 512;
}

inline int ln_grid(long long rows) {
    long long blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    static const long long cap = [] { const char* e = getenv("PK_LN_GRID_CAP"); return e ? atoll(e) : 2048LL; }();
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename T>
int launch_fwd(const void* x, const void* res, const void* gamma, const void* beta, void* z, void* y, float* mean,
               float* rstd, long long rows, int d, float eps, float p, unsigned long long seed,
               unsigned long long offset, hipStream_t s) {
    constexpr int EPV = 16 / sizeof(T);
    PK_CHECK_ARG(d % EPV == 0 && d <= 64 * EPV * 16, "pk_residual_ln_fwd: d=%d unsupported", d);
    unsigned thr = p > 0.f ? dropout_threshold(p) : 0u;
    float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    int nch = (d / EPV + 63) / 64;
    dim3 grid(ln_grid(rows)), block(256);
    if constexpr (sizeof(T) == 2) {
        static const bool off16 = getenv("PK_LN_FWD16") && atoi(getenv("PK_LN_FWD16")) == 0;  // (experiments)
        const bool ln = gamma != nullptr, zout = res || thr;
        if (!off16 && (d == 512 || d == 1024) && (!ln || (beta && mean && y && rstd)) && (ln || zout) && (zout == (z != nullptr)) &&
            !(thr && !res && !ln)) {
            const int fl = (res ? 1 : 0) | (thr ? 2 : 0) | (ln ? 4 : 0);
#define PK_LFWD(N, F)                                                                                                \
    hipLaunchKernelGGL((residual_ln_fwd16_kernel<T, N, F>), grid, block, 0, s, (const T*)x, (const T*)res,          \
                       (const T*)gamma, (const T*)beta, (T*)z, (T*)y, mean, rstd, rows, eps, thr, scale, seed, offset)
#define PK_LFWDN(N)                      \
    do {                                \
        if (fl == 4) PK_LFWD(N, 4);      \
        else if (fl == 5) PK_LFWD(N, 5); \
        else if (fl == 7) PK_LFWD(N, 7); \
        else if (fl == 1) PK_LFWD(N, 1); \
        else if (fl == 3) PK_LFWD(N, 3); \
        else fl16_done = false;         \
    } while (0)
            bool fl16_done = true;
            if (d == 512) PK_LFWDN(1);
            else PK_LFWDN(2);
#undef PK_LFWDN
#undef PK_LFWD
            if (fl16_done) {
                PK_LAUNCH_CHECK();
                return 0;
            }
        }
    }
#define PK_L(N)                                                                                                   \
    hipLaunchKernelGGL((residual_ln_fwd_kernel<T, N>), grid, block, 0, s, (const T*)x, (const T*)res,             \
                       (const T*)gamma, (const T*)beta, (T*)z, (T*)y, mean, rstd, rows, d, eps, thr, scale, seed, \
                       offset)
    if (nch <= 1) PK_L(1);
    else if (nch <= 2) PK_L(2);
    else if (nch <= 4) PK_L(4);
    else if (nch <= 8) PK_L(8);
    else PK_L(16);
#undef PK_L
    PK_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int launch_bwd(const void* dy, const void* dz_extra, const void* z, const void* gamma, const float* mean,
               const float* rstd, void* dres, void* dx, void* dgamma, void* dbeta, float* ws, size_t ws_bytes,
               long long rows, int d, float p, unsigned long long seed, unsigned long long offset, hipStream_t s,
               bool defer = false) {
    constexpr int EPV = 16 / sizeof(T);
    PK_CHECK_ARG(d % EPV == 0 && d <= 64 * EPV * 8, "pk_residual_ln_bwd: d=%d unsupported", d);
    unsigned thr = p > 0.f ? dropout_threshold(p) : 0u;
    float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    int nch = (d / EPV + 63) / 64;
    int nblocks = ln_grid(rows);
    if (nblocks > ln_bwd_max_blocks(d)) nblocks = ln_bwd_max_blocks(d);
    bool want_pg = gamma && (dgamma || dbeta || defer);
    if (want_pg) {
        size_t need = (size_t)nblocks * 2 * d * sizeof(float);
        PK_CHECK_ARG(ws && ws_bytes >= need, "pk_residual_ln_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    }
    dim3 grid(nblocks), block(256);
    if constexpr (sizeof(T) == 2) {
        static const bool off16 = getenv("PK_LN_BWD16") && atoi(getenv("PK_LN_BWD16")) == 0;  // (experiments)
        const bool dropx = dx && thr;
        if (!off16 && (d == 512 || d == 1024) && gamma && mean && dres && (dropx || !dx)) {
            const int fl = (dz_extra ? 1 : 0) | (dropx ? 2 : 0);
#define PK_L16(N, F)                                                                                                   \
    hipLaunchKernelGGL((residual_ln_bwd16_kernel<T, N, F>), grid, block, 0, s, (const T*)dy, (const T*)dz_extra,       \
                       (const T*)z, (const T*)gamma, mean, rstd, (T*)dres, (T*)dx, want_pg ? ws : nullptr, rows, thr, \
                       scale, seed, offset)
#define PK_L16N(N)                      \
    do {                                \
        if (fl == 0) PK_L16(N, 0);      \
        else if (fl == 1) PK_L16(N, 1); \
        else if (fl == 2) PK_L16(N, 2); \
        else PK_L16(N, 3);              \
    } while (0)
            if (d == 512) PK_L16N(1);
            else PK_L16N(2);
#undef PK_L16N
#undef PK_L16
            PK_LAUNCH_CHECK();
            if (want_pg && !defer) {
                LnReduceItems it = {};
                it.partials[0] = ws; it.dgamma[0] = dgamma; it.dbeta[0] = dbeta;
                hipLaunchKernelGGL((ln_param_grad_kernel<T>), dim3((d + 15) / 16, 2, 1), dim3(1024), 0, s, it, nblocks, d);
                PK_LAUNCH_CHECK();
            }
            return 0;
        }
    }
#define PK_L(N)                                                                                                  \
    hipLaunchKernelGGL((residual_ln_bwd_kernel<T, N>), grid, block, 0, s, (const T*)dy, (const T*)dz_extra,      \
                       (const T*)z, (const T*)gamma, mean, rstd, (T*)dres, (T*)dx, want_pg ? ws : nullptr, rows, \
                       d, thr, scale, seed, offset)
    if (nch <= 1) PK_L(1);
    else if (nch <= 2) PK_L(2);
    else if (nch <= 4) PK_L(4);
    else PK_L(8);
#undef PK_L
    PK_LAUNCH_CHECK();
    if (want_pg && !defer) {
        LnReduceItems it = {};
        it.partials[0] = ws; it.dgamma[0] = dgamma; it.dbeta[0] = dbeta;
        hipLaunchKernelGGL((ln_param_grad_kernel<T>), dim3((d + 15) / 16, 2, 1), dim3(1024), 0, s, it, nblocks, d);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" int pk_residual_ln_fwd(const void* x, const void* residual, const void* gamma, const void* beta,
                                  void* z_out, void* y_out, float* mean, float* rstd, long long rows, int d,
                                  float eps, float drop_p, unsigned long long seed, unsigned long long offset,
                                  int dtype, void* stream) {
    if (rows == 0) return 0;
    PK_CHECK_ARG(x, "pk_residual_ln_fwd: x is null");
    PK_CHECK_ARG(!gamma || (y_out && rstd), "pk_residual_ln_fwd: gamma given but y/rstd missing");
    PK_CHECK_ARG(!gamma || mean || !beta, "pk_residual_ln_fwd: RMSNorm (mean == NULL) has no beta");
    PK_CHECK_ARG(gamma || z_out, "pk_residual_ln_fwd: nothing to compute");
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_residual_ln_fwd: bad dropout %f", drop_p);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PK_BF16)
        return launch_fwd<bf16>(x, residual, gamma, beta, z_out, y_out, mean, rstd, rows, d, eps, drop_p, seed, offset, s);
    if (dtype == PK_F16)
        return launch_fwd<f16>(x, residual, gamma, beta, z_out, y_out, mean, rstd, rows, d, eps, drop_p, seed, offset, s);
    if (dtype == PK_F32)
        return launch_fwd<float>(x, residual, gamma, beta, z_out, y_out, mean, rstd, rows, d, eps, drop_p, seed, offset, s);
    PK_CHECK_ARG(false, "pk_residual_ln_fwd: dtype %d not supported", dtype);
}

extern "C" size_t pk_residual_ln_bwd_workspace(long long rows, int d) {
    long long blocks = ln_grid(rows);
    if (blocks > ln_bwd_max_blocks(d)) blocks = ln_bwd_max_blocks(d);
    return (size_t)blocks * 2 * d * sizeof(float);
}

extern "C" int pk_residual_ln_bwd(const void* dy, const void* dz_extra, const void* z, const void* gamma,
                                  const float* mean, const float* rstd, void* dres_out, void* dx_out,
                                  void* dgamma, void* dbeta, void* workspace, size_t ws_bytes, long long rows,
                                  int d, float drop_p, unsigned long long seed, unsigned long long offset,
                                  int dtype, void* stream) {
    PK_CHECK_ARG(!gamma || (dy && z && rstd), "pk_residual_ln_bwd: LN inputs missing");
    PK_CHECK_ARG(!gamma || mean || !dbeta, "pk_residual_ln_bwd: RMSNorm (mean == NULL) has no beta gradient");
    PK_CHECK_ARG(gamma || dz_extra, "pk_residual_ln_bwd: nothing to compute");
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_residual_ln_bwd: bad dropout %f", drop_p);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PK_BF16)
        return launch_bwd<bf16>(dy, dz_extra, z, gamma, mean, rstd, dres_out, dx_out, dgamma, dbeta, (float*)workspace,
                                ws_bytes, rows, d, drop_p, seed, offset, s);
    if (dtype == PK_F16)
        return launch_bwd<f16>(dy, dz_extra, z, gamma, mean, rstd, dres_out, dx_out, dgamma, dbeta, (float*)workspace,
                               ws_bytes, rows, d, drop_p, seed, offset, s);
    if (dtype == PK_F32)
        return launch_bwd<float>(dy, dz_extra, z, gamma, mean, rstd, dres_out, dx_out, dgamma, dbeta,
                                 (float*)workspace, ws_bytes, rows, d, drop_p, seed, offset, s);
    PK_CHECK_ARG(false, "pk_residual_ln_bwd: dtype %d not supported", dtype);
}

// pk_residual_ln_bwd that leaves dgamma / dbeta as per-workgroup partial sums in `workspace`; pk_ln_param_grads finishes them
extern "C" int pk_residual_ln_bwd_partials(const void* dy, const void* dz_extra, const void* z, const void* gamma,
                                           const float* mean, const float* rstd, void* dres_out, void* dx_out,
                                           void* workspace, size_t ws_bytes, long long rows, int d, float drop_p,
                                           unsigned long long seed, unsigned long long offset, int dtype, void* stream) {
    PK_CHECK_ARG(gamma && dy && z && rstd && workspace, "pk_residual_ln_bwd_partials: LN inputs missing");
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_residual_ln_bwd_partials: bad dropout %f", drop_p);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PK_BF16)
        return launch_bwd<bf16>(dy, dz_extra, z, gamma, mean, rstd, dres_out, dx_out, nullptr, nullptr, (float*)workspace,
                                ws_bytes, rows, d, drop_p, seed, offset, s, true);
    if (dtype == PK_F16)
        return launch_bwd<f16>(dy, dz_extra, z, gamma, mean, rstd, dres_out, dx_out, nullptr, nullptr, (float*)workspace,
                               ws_bytes, rows, d, drop_p, seed, offset, s, true);
    if (dtype == PK_F32)
        return launch_bwd<float>(dy, dz_extra, z, gamma, mean, rstd, dres_out, dx_out, nullptr, nullptr, (float*)workspace,
                                 ws_bytes, rows, d, drop_p, seed, offset, s, true);
    PK_CHECK_ARG(false, "pk_residual_ln_bwd_partials: dtype %d not supported", dtype);
}

// the parameter gradients of up to PK_LN_GROUP_MAX LayerNorms (same rows, same d) from their partial sums, in ONE launch —
// the same reduction, slab order included, as pk_residual_ln_bwd runs for one
extern "C" int pk_ln_param_grads(const PkLnParamGrad* items, int n, long long rows, int d, int dtype, void* stream) {
    PK_CHECK_ARG(items && n >= 1 && n <= PK_LN_GROUP_MAX, "pk_ln_param_grads: 1..%d items", PK_LN_GROUP_MAX);
    if (rows == 0) return 0;
    int nblocks = ln_grid(rows);
    if (nblocks > ln_bwd_max_blocks(d)) nblocks = ln_bwd_max_blocks(d);
    LnReduceItems it = {};
    for (int i = 0; i < n; ++i) {
        PK_CHECK_ARG(items[i].workspace, "pk_ln_param_grads: item %d has no workspace", i);
        it.partials[i] = (const float*)items[i].workspace; it.dgamma[i] = items[i].dgamma; it.dbeta[i] = items[i].dbeta;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((d + 15) / 16, 2, n), block(1024);
    if (dtype == PK_BF16) hipLaunchKernelGGL((ln_param_grad_kernel<bf16>), grid, block, 0, s, it, nblocks, d);
    else if (dtype == PK_F16) hipLaunchKernelGGL((ln_param_grad_kernel<f16>), grid, block, 0, s, it, nblocks, d);
    else if (dtype == PK_F32) hipLaunchKernelGGL((ln_param_grad_kernel<float>), grid, block, 0, s, it, nblocks, d);
    else PK_CHECK_ARG(false, "pk_ln_param_grads: dtype %d not supported", dtype);
    PK_LAUNCH_CHECK();
    return 0;
}
