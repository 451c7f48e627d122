// Elementwise helpers of the speech frontend (K7) and of the non-fused fallback paths:
//   activation fwd/bwd (pasero/models/modules.py:220-228), GLU over the channel dim (modules.py:802, nn.GLU),
//   col2im for the Conv1d input gradient (modules.py:793-799 backward).
// HBM-bound: 16-byte accesses, grid-stride over <= 2048 workgroups.
#include "common.h"

namespace {

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                  T* __restrict__ out, long long n, int act) {
    constexpr int EPV = 16 / sizeof(T);
    const long long nvec = n / EPV;
    for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < nvec; ch += (long long)gridDim.x * 256) {
        Vec16<T> xv = load16<T>(x + ch * EPV), o, g;
        if (BWD) g = load16<T>(dy + ch * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) o.set(e, BWD ? g.get(e) * act_bwd_t<T>(act, xv.get(e)) : act_fwd_t<T>(act, xv.get(e)));
        store16<T>(out + ch * EPV, o);
    }
    if (blockIdx.x == 0)
        for (long long i = nvec * EPV + threadIdx.x; i < n; i += 256) {
            float xv = to_f32<T>(x[i]);
            out[i] = from_f32<T>(BWD ? to_f32<T>(dy[i]) * act_bwd_t<T>(act, xv) : act_fwd_t<T>(act, xv));
        }
}

template <typename T>
__global__ __launch_bounds__(256) void gated_act_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ z,
                                                            const T* __restrict__ u, T* __restrict__ dz,
                                                            T* __restrict__ du, long long n, int act) {
    constexpr int EPV = 16 / sizeof(T);
    const long long nvec = n / EPV;
    for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < nvec; ch += (long long)gridDim.x * 256) {
        Vec16<T> g = load16<T>(dh + ch * EPV), zv = load16<T>(z + ch * EPV), uv = load16<T>(u + ch * EPV), o1, o2;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            o1.set(e, g.get(e) * uv.get(e) * act_bwd_t<T>(act, zv.get(e)));
            o2.set(e, g.get(e) * act_fwd_t<T>(act, zv.get(e)));
        }
        store16<T>(dz + ch * EPV, o1);
        store16<T>(du + ch * EPV, o2);
    }
    if (blockIdx.x == 0)
        for (long long i = nvec * EPV + threadIdx.x; i < n; i += 256) {
            float gi = to_f32<T>(dh[i]), zi = to_f32<T>(z[i]), ui = to_f32<T>(u[i]);
            dz[i] = from_f32<T>(gi * ui * act_bwd_t<T>(act, zi));
            du[i] = from_f32<T>(gi * act_fwd_t<T>(act, zi));
        }
}

// x [rows][2C] -> out [rows][C] = a * sigmoid(b)
template <typename T>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, long long rows,
                                                      int C) {
    long long total = rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i / C;
        int c = (int)(i % C);
        float a = to_f32<T>(x[r * 2 * C + c]), b = to_f32<T>(x[r * 2 * C + C + c]);
        out[i] = from_f32<T>(a / (1.f + __expf(-b)));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      T* __restrict__ dx, long long rows, int C) {
    long long total = rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i / C;
        int c = (int)(i % C);
        float a = to_f32<T>(x[r * 2 * C + c]), b = to_f32<T>(x[r * 2 * C + C + c]), g = to_f32<T>(dy[i]);
        float s = 1.f / (1.f + __expf(-b));
        dx[r * 2 * C + c] = from_f32<T>(g * s);
        dx[r * 2 * C + C + c] = from_f32<T>(g * a * s * (1.f - s));
    }
}

// dx[b][l][c] = sum_{j} dA[b*R + r][j*C + c]  over (r, j) with r*stride + j - pad == l, 0 <= r < Lout
template <typename T>
__global__ __launch_bounds__(256) void col2im1d_kernel(const T* __restrict__ dA, T* __restrict__ dx, int B, int L,
                                                       int C, int R, int Lout, int ksize, int stride, int pad) {
    long long total = (long long)B * L * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int c = (int)(i % C);
        long long t = i / C;
        int l = (int)(t % L), b = (int)(t / L);
        float s = 0.f;
        for (int j = 0; j < ksize; ++j) {
            int num = l + pad - j;
            if (num < 0 || num % stride) continue;
            int r = num / stride;
            if (r >= Lout) continue;
            s += to_f32<T>(dA[((long long)b * R + r) * ksize * C + (long long)j * C + c]);
        }
        dx[i] = from_f32<T>(s);
    }
}

// the same for 16-byte chunks of channels (C a multiple of the chunk width, 16-byte aligned rows): the scalar kernel above moved
// Whisper's second conv gradient (16 x 3000 x 512) at 1 TB/s — two-byte accesses and two integer divisions per element
template <typename T>
__global__ __launch_bounds__(256) void col2im1d_vec_kernel(const T* __restrict__ dA, T* __restrict__ dx, int B, int L,
                                                           int C, int R, int Lout, int ksize, int stride, int pad) {
    constexpr int EPV = 16 / sizeof(T);
    const int cpr = C / EPV;
    const long long total = (long long)B * L * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const long long t = i / cpr;
        const int l = (int)(t % L), b = (int)(t / L);
        float s[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) s[e] = 0.f;
        for (int j = 0; j < ksize; ++j) {  // (the same order of additions as the scalar kernel: bit for bit its result)
            const int num = l + pad - j;
            if (num < 0 || num % stride) continue;
            const int r = num / stride;
            if (r >= Lout) continue;
            const Vec16<T> v = load16<T>(dA + ((long long)b * R + r) * ksize * C + (long long)j * C + ch * EPV);
#pragma unroll
            for (int e = 0; e < EPV; ++e) s[e] += v.get(e);
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPV; ++e) o.set(e, s[e]);
        store16<T>(dx + i * EPV, o);
    }
}

// rotary position embedding on the first `ncols` columns of each row (heads of 64: q|k of a packed projection):
//   y[i] = x[i] cos_i - x[i+32] sin_i ;  y[i+32] = x[i+32] cos_i + x[i] sin_i       (i < 32, angle = pos * inv_freq_i)
// inverse = 1 rotates by -angle (the backward pass).  cos/sin: fp32 tables [max_pos][head_dim / 2].
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows, int Tlen,
                                                   long long ld, int ncols, int total_cols,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                   int pos_offset, int inverse, int hd) {
    constexpr int EPV = 16 / sizeof(T);
    const int cpr = total_cols / EPV;  // 16-byte chunks per row
    const long long total = rows * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / cpr;
        const int col = (int)(i % cpr) * EPV;
        Vec16<T> v = load16<T>(x + row * ld + col);
        if (col < ncols) {
            const int half = hd >> 1;
            const int d = col & (hd - 1);      // position inside the head
            const int lo = d & (half - 1);     // frequency index of element 0 of this chunk
            const bool upper = d >= half;
            Vec16<T> w = load16<T>(x + row * ld + (upper ? col - half : col + half));  // the partner half
            const int pos = pos_offset + (int)(row % Tlen);
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float c = cos_t[pos * half + lo + e], sn = sin_t[pos * half + lo + e];
                if (inverse) sn = -sn;
                float a = v.get(e), b = w.get(e);
                o.set(e, upper ? a * c + b * sn : a * c - b * sn);
            }
            v = o;
        }
        store16<T>(y + row * ld + col, v);
    }
}

inline int grid_for(long long work, int per_block) {
    long long b = (work + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace


extern "C" int pk_act_fwd(const void* x, void* out, long long n, int act, int dtype, void* stream) {
    PK_CHECK_ARG(x && out, "pk_act_fwd: null tensor");
    PK_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0, "pk_act_fwd: 16-byte alignment required");
    if (n == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_act_fwd", {
        hipLaunchKernelGGL((act_kernel<T, false>), dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)nullptr, (const T*)x, (T*)out, n, act);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_act_bwd(const void* dy, const void* x, void* out, long long n, int act, int dtype, void* stream) {
    PK_CHECK_ARG(dy && x && out, "pk_act_bwd: null tensor");
    PK_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)dy % 16) == 0,
                 "pk_act_bwd: 16-byte alignment required");
    if (n == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_act_bwd", {
        hipLaunchKernelGGL((act_kernel<T, true>), dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dy, (const T*)x, (T*)out, n, act);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_glu_fwd(const void* x, void* out, long long rows, int C, int dtype, void* stream) {
    PK_CHECK_ARG(x && out && C > 0, "pk_glu_fwd: bad arguments");
    if (rows == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_glu_fwd", {
        hipLaunchKernelGGL((glu_fwd_kernel<T>), dim3(grid_for(rows * C, 1024)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)out, rows, C);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_glu_bwd(const void* dy, const void* x, void* dx, long long rows, int C, int dtype, void* stream) {
    PK_CHECK_ARG(dy && x && dx && C > 0, "pk_glu_bwd: bad arguments");
    if (rows == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_glu_bwd", {
        hipLaunchKernelGGL((glu_bwd_kernel<T>), dim3(grid_for(rows * C, 1024)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dy, (const T*)x, (T*)dx, rows, C);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_col2im1d(const void* dA, void* dx, int B, int L, int C, int R, int Lout, int ksize, int stride,
                           int pad, int dtype, void* stream) {
    PK_CHECK_ARG(dA && dx && stride > 0 && ksize > 0, "pk_col2im1d: bad arguments");
    if ((long long)B * L * C == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_col2im1d", {
        constexpr int EPV = 16 / sizeof(T);
        if (C % EPV == 0 && ((uintptr_t)dA % 16) == 0 && ((uintptr_t)dx % 16) == 0)
            hipLaunchKernelGGL((col2im1d_vec_kernel<T>), dim3(grid_for((long long)B * L * (C / EPV), 2048)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)dA, (T*)dx, B, L, C, R, Lout, ksize, stride, pad);
        else
            hipLaunchKernelGGL((col2im1d_kernel<T>), dim3(grid_for((long long)B * L * C, 1024)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)dA, (T*)dx, B, L, C, R, Lout, ksize, stride, pad);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_rope(const void* x, void* y, long long rows, int Tlen, long long ld, int ncols, int total_cols,
                       const float* cos_t, const float* sin_t, int max_pos, int pos_offset, int inverse, int head_dim,
                       int dtype, void* stream) {
    PK_CHECK_ARG(x && y && cos_t && sin_t, "pk_rope: null tensor");
    PK_CHECK_ARG(x != y, "pk_rope: in-place rotation is not supported (partner halves are read from x)");
    PK_CHECK_ARG(head_dim == 64 || head_dim == 128, "pk_rope: head_dim %d (64 or 128)", head_dim);
    PK_CHECK_ARG(ncols % head_dim == 0 && ncols <= total_cols && total_cols % 8 == 0 && ld % 8 == 0, "pk_rope: bad column layout");
    PK_CHECK_ARG(Tlen > 0 && pos_offset >= 0 && pos_offset + Tlen <= max_pos, "pk_rope: positions exceed the cos/sin table");
    if (rows == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_rope", {
        hipLaunchKernelGGL((rope_kernel<T>), dim3(grid_for(rows * (total_cols / (16 / (int)sizeof(T))), 512)), dim3(256),
                           0, (hipStream_t)stream, (const T*)x, (T*)y, rows, Tlen, ld, ncols, total_cols, cos_t, sin_t,
                           pos_offset, inverse, head_dim);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_gated_act_bwd(const void* dh, const void* z, const void* u, void* dz, void* du, long long n, int act,
                                int dtype, void* stream) {
    PK_CHECK_ARG(dh && z && u && dz && du, "pk_gated_act_bwd: null tensor");
    if (n == 0) return 0;
    PK_DTYPE_SWITCH(dtype, "pk_gated_act_bwd", {
        hipLaunchKernelGGL((gated_act_bwd_kernel<T>), dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dh, (const T*)z, (const T*)u, (T*)dz, (T*)du, n, act);
    })
    PK_LAUNCH_CHECK();
    return 0;
}

// ---- feature collate: ragged rows -> zero-padded (B, Tmax, D) batch with dtype conversion ----
#include <hip/hip_fp16.h>
namespace {
template <typename S> __device__ __forceinline__ float src_to_f32(S v);
template <> __device__ __forceinline__ float src_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float src_to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float src_to_f32<bf16>(bf16 v) { return __bfloat162float(v); }

// one workgroup per output row (b, t); offsets[b]..offsets[b+1] are the rows of sequence b in `src`
template <typename S, typename T>
__global__ __launch_bounds__(256) void pad_rows_kernel(const S* __restrict__ src, const long long* __restrict__ offsets,
                                                       T* __restrict__ out, long long Tmax, int D) {
    const long long row = blockIdx.x, b = row / Tmax, t = row % Tmax;
    const long long beg = offsets[b], len = offsets[b + 1] - beg;
    T* o = out + row * D;
    if (t < len) {
        const S* s = src + (beg + t) * D;
        for (int c = threadIdx.x; c < D; c += blockDim.x) o[c] = from_f32<T>(src_to_f32<S>(s[c]));
    } else {
        for (int c = threadIdx.x; c < D; c += blockDim.x) o[c] = from_f32<T>(0.f);
    }
}
}  // namespace

extern "C" int pk_pad_rows(const void* src, int src_dtype, const long long* offsets, void* out, int out_dtype, int B,
                           long long Tmax, int D, void* stream) {
    if (B == 0 || Tmax == 0 || D == 0) return 0;
    PK_CHECK_ARG(src && offsets && out && B > 0 && Tmax > 0 && D > 0, "pk_pad_rows: bad arguments");
    PK_CHECK_ARG((long long)B * Tmax < (1ll << 31), "pk_pad_rows: too many rows");
    dim3 grid((unsigned)((long long)B * Tmax)), block(D >= 256 ? 256 : 64);
    hipStream_t s = (hipStream_t)stream;
#define PK_PAD(S, T) hipLaunchKernelGGL((pad_rows_kernel<S, T>), grid, block, 0, s, (const S*)src, offsets, (T*)out, Tmax, D)
    const int key = src_dtype * 10 + out_dtype;  // src / out: 0 f32, 1 bf16, 2 f16
    switch (key) {
        case 0: PK_PAD(float, float); break;
        case 1: PK_PAD(float, bf16); break;
        case 10: PK_PAD(bf16, float); break;
        case 11: PK_PAD(bf16, bf16); break;
        case 20: PK_PAD(__half, float); break;
        case 21: PK_PAD(__half, bf16); break;
        case 2: PK_PAD(float, f16); break;
        case 12: PK_PAD(bf16, f16); break;
        case 22: PK_PAD(__half, f16); break;
        default: PK_CHECK_ARG(false, "pk_pad_rows: dtypes (%d -> %d) not supported", src_dtype, out_dtype);
    }
#undef PK_PAD
    PK_LAUNCH_CHECK();
    return 0;
}
