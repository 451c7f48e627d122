// "Next" row — fused gradient normalisation + global-norm clipping + Adam, multi-tensor:
//   reference: per-parameter Python loops in Trainer.train_step `p.grad.mul_(dp_size / num_tokens)` (pasero/training.py:
//   455-470), optimization.clip_grad_norm_ (pasero/optimization.py:390-427) and the fairseq-style Adam.step with fp32
//   moments and an fp32 copy of bf16 parameters (pasero/optimization.py:56-149): ~250 tensors x ~6 tiny launches/step.
// Here: two launches over a chunk list that spans all tensors (65536 elements per workgroup):
//   pk_mt_sqnorm : sum of squares of every gradient (fp32 accumulate, fp64 final) -> device scalar ||g * scale||
//   pk_mt_adam   : g' = g * scale * clip(||.||);  m, v (fp32) update;  p <- p - lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
//                  with decoupled weight decay; bf16 params are read as fp32, updated, rounded once.
// The clip coefficient is read from device memory: no host synchronisation between backward and the update.
#include <algorithm>
#include "common.h"

namespace {

constexpr int CHUNK = 65536;

struct MTList {
    const long long* p;        // [ntensors] parameter pointers
    const long long* g;        // gradient pointers
    const long long* m;        // exp_avg (fp32)
    const long long* v;        // exp_avg_sq (fp32)
    const long long* numel;    // [ntensors]
    const int* chunk_tensor;   // [nchunks]
    const long long* chunk_start;
    int ntensors;
    const float* bias_corr;    // [2 * ntensors]: 1 - beta1^step_t | sqrt(1 - beta2^step_t), or null (one step for all)
};

template <typename T>
__global__ __launch_bounds__(256) void mt_sqnorm_kernel(MTList L, float* __restrict__ partial) {
    const int c = blockIdx.x, t = L.chunk_tensor[c];
    const long long start = L.chunk_start[c];
    const long long n = min((long long)CHUNK, L.numel[t] - start);
    const T* g = reinterpret_cast<const T*>(L.g[t]) + start;
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        float x = to_f32<T>(g[i]);
        s += x * x;
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[c] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = scale * sqrt(sum partial)   (global gradient norm after normalisation)
__global__ __launch_bounds__(1024) void mt_norm_finalize_kernel(const float* __restrict__ partial, int n, float scale,
                                                                float* __restrict__ out) {
    __shared__ double sh[16];
    double a = 0;
    for (int i = threadIdx.x; i < n; i += 1024) a += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < 16; ++w) s += sh[w];
        out[0] = (float)(sqrt(s) * (double)scale);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mt_adam_kernel(MTList L, const float* __restrict__ gnorm, float scale,
                                                      float max_norm, float lr, float beta1, float beta2, float eps,
                                                      float weight_decay, float bc1, float bc2_sqrt) {  // (bc*: defaults)
    const int c = blockIdx.x, t = L.chunk_tensor[c];
    const long long start = L.chunk_start[c];
    const long long n = min((long long)CHUNK, L.numel[t] - start);
    T* p = reinterpret_cast<T*>(L.p[t]) + start;
    const T* g = reinterpret_cast<const T*>(L.g[t]) + start;
    float* m = reinterpret_cast<float*>(L.m[t]) + start;
    float* v = reinterpret_cast<float*>(L.v[t]) + start;
    float coef = scale;
    if (max_norm > 0.f && gnorm) coef *= fminf(max_norm / (gnorm[0] + 1e-6f), 1.f);
    if (L.bias_corr) {  // per-parameter step counts (parameters that skipped steps: optimization.py:120-125)
        bc1 = L.bias_corr[t];
        bc2_sqrt = L.bias_corr[L.ntensors + t];
    }
    const float step_size = lr / bc1;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float gi = to_f32<T>(g[i]) * coef;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float pi = to_f32<T>(p[i]);
        if (weight_decay != 0.f) pi -= weight_decay * lr * pi;
        pi -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = from_f32<T>(pi);
    }
}

// dst_t[i] = src_t[i] (optionally * alpha) for every tensor of a table: the gradient pack of a data-parallel bucket
// (ddp.py), one launch per bucket; 16-byte accesses where both sides allow
template <typename T>
__global__ __launch_bounds__(256) void mt_copy_kernel(const long long* __restrict__ src, const long long* __restrict__ dst,
                                                      const long long* __restrict__ numel,
                                                      const int* __restrict__ chunk_tensor,
                                                      const long long* __restrict__ chunk_start) {
    const int c = blockIdx.x, t = chunk_tensor[c];
    const long long start = chunk_start[c];
    const long long n = min((long long)CHUNK, numel[t] - start);
    const T* s = reinterpret_cast<const T*>(src[t]) + start;
    T* d = reinterpret_cast<T*>(dst[t]) + start;
    constexpr int V = 16 / sizeof(T);
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const long long nv = n / V;
        for (long long i = threadIdx.x; i < nv; i += 256)
            reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (long long i = nv * V + threadIdx.x; i < n; i += 256) d[i] = s[i];
    } else {
        for (long long i = threadIdx.x; i < n; i += 256) d[i] = s[i];
    }
}

MTList make_list(const long long* ptrs, int ntensors, const int* chunk_tensor, const long long* chunk_start,
                 const float* bias_corr = nullptr) {
    MTList L;
    L.ntensors = ntensors;
    L.bias_corr = bias_corr;
    L.p = ptrs;
    L.g = ptrs + ntensors;
    L.m = ptrs + 2 * ntensors;
    L.v = ptrs + 3 * ntensors;
    L.numel = ptrs + 4 * ntensors;
    L.chunk_tensor = chunk_tensor;
    L.chunk_start = chunk_start;
    return L;
}

}  // namespace

extern "C" int pk_mt_chunk_size(void) { return CHUNK; }

// `table` (device, int64): [p ptrs | g ptrs | m ptrs | v ptrs | numel], each of length ntensors.
// Writes the chunk sums of squares of this table's gradients to partial[0 .. nchunks).  With `gnorm_out`:
// gnorm_out[0] = scale * sqrt(sum of partial_all[0 .. n_all)) — the ONE global norm over every table (parameter group /
// dtype) whose partials were written into `partial_all` by earlier calls (optimization.py:390-427 clip_grad_norm_).
extern "C" int pk_mt_sqnorm(const long long* table, int ntensors, const int* chunk_tensor, const long long* chunk_start,
                            int nchunks, float scale, float* partial, const float* partial_all, int n_all,
                            float* gnorm_out, int dtype, void* stream) {
    PK_CHECK_ARG(table && chunk_tensor && chunk_start && partial, "pk_mt_sqnorm: null argument");
    PK_CHECK_ARG(!gnorm_out || (partial_all && n_all >= 0), "pk_mt_sqnorm: gnorm_out needs partial_all");
    hipStream_t s = (hipStream_t)stream;
    MTList L = make_list(table, ntensors, chunk_tensor, chunk_start);
    if (nchunks > 0) {
        if (dtype == PK_BF16) hipLaunchKernelGGL((mt_sqnorm_kernel<bf16>), dim3(nchunks), dim3(256), 0, s, L, partial);
        else if (dtype == PK_F16) hipLaunchKernelGGL((mt_sqnorm_kernel<f16>), dim3(nchunks), dim3(256), 0, s, L, partial);
        else if (dtype == PK_F32) hipLaunchKernelGGL((mt_sqnorm_kernel<float>), dim3(nchunks), dim3(256), 0, s, L, partial);
        else PK_CHECK_ARG(false, "pk_mt_sqnorm: dtype %d not supported", dtype);
        PK_LAUNCH_CHECK();
    }
    if (gnorm_out) {
        hipLaunchKernelGGL(mt_norm_finalize_kernel, dim3(1), dim3(1024), 0, s, partial_all, n_all, scale, gnorm_out);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

// One Adam step over every tensor of the table.  `gnorm` (device scalar from pk_mt_sqnorm, or NULL) drives clipping.
// `bias_corr` (device, fp32, [2 * ntensors]: 1 - beta1^step_t | sqrt(1 - beta2^step_t)) carries per-parameter step
// counts; NULL = every tensor is at `step`.
extern "C" int pk_mt_adam(const long long* table, int ntensors, const int* chunk_tensor, const long long* chunk_start,
                          int nchunks, const float* gnorm, float scale, float max_norm, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int step, const float* bias_corr, int dtype,
                          void* stream) {
    PK_CHECK_ARG(table && chunk_tensor && chunk_start, "pk_mt_adam: null argument");
    PK_CHECK_ARG(step >= 1 || bias_corr, "pk_mt_adam: step must be >= 1");
    if (nchunks == 0) return 0;
    const float bc1 = 1.f - powf(beta1, (float)std::max(step, 1));
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)std::max(step, 1)));
    hipStream_t s = (hipStream_t)stream;
    MTList L = make_list(table, ntensors, chunk_tensor, chunk_start, bias_corr);
    if (dtype == PK_BF16)
        hipLaunchKernelGGL((mt_adam_kernel<bf16>), dim3(nchunks), dim3(256), 0, s, L, gnorm, scale, max_norm, lr, beta1,
                           beta2, eps, weight_decay, bc1, bc2_sqrt);
    else if (dtype == PK_F16)
        hipLaunchKernelGGL((mt_adam_kernel<f16>), dim3(nchunks), dim3(256), 0, s, L, gnorm, scale, max_norm, lr, beta1,
                           beta2, eps, weight_decay, bc1, bc2_sqrt);
    else if (dtype == PK_F32)
        hipLaunchKernelGGL((mt_adam_kernel<float>), dim3(nchunks), dim3(256), 0, s, L, gnorm, scale, max_norm, lr,
                           beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    else PK_CHECK_ARG(false, "pk_mt_adam: dtype %d not supported", dtype);
    PK_LAUNCH_CHECK();
    return 0;
}

// Multi-tensor copy: `table` (device, int64) = [src ptrs | dst ptrs | numel], each `ntensors` long; same chunk list as
// above.  Replaces the `torch._foreach_copy_` pack of a gradient bucket (DDP reducer, pasero/training.py:243-250).
extern "C" int pk_mt_copy(const long long* table, int ntensors, const int* chunk_tensor, const long long* chunk_start,
                          int nchunks, int dtype, void* stream) {
    PK_CHECK_ARG(table && chunk_tensor && chunk_start, "pk_mt_copy: null argument");
    if (nchunks == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long *src = table, *dst = table + ntensors, *numel = table + 2 * ntensors;
    if (dtype == PK_F32)
        hipLaunchKernelGGL((mt_copy_kernel<float>), dim3(nchunks), dim3(256), 0, s, src, dst, numel, chunk_tensor, chunk_start);
    else if (dtype == PK_BF16 || dtype == PK_F16)
        hipLaunchKernelGGL((mt_copy_kernel<unsigned short>), dim3(nchunks), dim3(256), 0, s, src, dst, numel, chunk_tensor, chunk_start);
    else PK_CHECK_ARG(false, "pk_mt_copy: dtype %d not supported", dtype);
    PK_LAUNCH_CHECK();
    return 0;
}
