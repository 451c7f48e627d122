// K1 backward — gradient of the embedding lookup (pasero/models/modules.py:916-933: dE[id] += dout[tok] for every token,
// row padding_idx excluded), DETERMINISTIC: no float atomics.
//
// The contributions are already in memory, one row of `dout` per token; what is missing is, per vocabulary row, the list
// of its tokens in a fixed order.  That list is a stable sort of the token positions by id:
//   1. keys = clamp(ids) (32-bit), values = 0 .. ntok-1                               (one small kernel)
//   2. rocPRIM radix sort of (key, value) over the bits of V (stable: ties keep ascending token order)
//   3. dE <- 0 (rows without tokens), then one workgroup per 64 sorted positions: for every SEGMENT HEAD in its range
//      (first position of a run of equal ids) the four waves sum the segment's rows — wave w takes rows w, w + 4, ...
//      in order, fp32, 16-byte loads, dropout mask regenerated from (seed, offset, element) — combine their partial
//      sums through LDS in wave order and write the row once, rounded once.
// Every row is therefore a fixed-order sum: bitwise reproducible from run to run (the round-1 kernel added with fp32
// atomics: fast enough, 66 us at C2, but the one non-deterministic kernel of the training step).  Traffic: ntok rows
// read once, V rows zeroed, unique rows written — no fp32 V x d accumulation buffer and no conversion pass over it
// (at V = 256 206, d = 1024 that buffer alone was 1 GB of memset + 1 GB read per call).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void embed_keys_kernel(const long long* __restrict__ ids, unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ vals, long long ntok, long long V) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ntok; i += (long long)gridDim.x * 256) {
        long long id = ids[i];
        keys[i] = (unsigned)(id < 0 ? 0 : (id >= V ? V - 1 : id));  // same clamp as the forward kernel
        vals[i] = (unsigned)i;
    }
}

constexpr int POS_PER_WG = 64;

// columns [c0, c0 + 8) of row `tok` of dout, scaled, dropout mask applied, added to a[0..8)
template <typename T>
__device__ __forceinline__ void add_row_chunk(float (&a)[8], const T* __restrict__ dout, long long tok, int d, int c0,
                                              float scale, unsigned thr, float drop_scale, unsigned long long seed,
                                              unsigned long long offset) {
    const long long off = tok * d + c0;
    float g[8];
    if constexpr (sizeof(T) == 2) {
        Vec16<T> v = load16<T>(dout + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = v.get(e);
    } else {
        const float4 lo = *reinterpret_cast<const float4*>(dout + off), hi = *reinterpret_cast<const float4*>(dout + off + 4);
        g[0] = lo.x; g[1] = lo.y; g[2] = lo.z; g[3] = lo.w; g[4] = hi.x; g[5] = hi.y; g[6] = hi.z; g[7] = hi.w;
    }
    if (thr) {  // the mask embed_fwd_kernel drew (common.h: dropout_keep8 — `off` is a multiple of 8)
        bool keep[8];
        dropout_keep8(seed, offset, (unsigned long long)off >> 3, thr, keep);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = keep[e] ? g[e] * drop_scale : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += g[e] * scale;
}

// NB column blocks of 512 per lane (d <= 512 * NB); d % 8 == 0
template <typename T, int NB>
__global__ __launch_bounds__(256) void embed_segsum_kernel(const unsigned* __restrict__ keys,
                                                           const unsigned* __restrict__ toks,
                                                           const T* __restrict__ dout, T* __restrict__ dE,
                                                           long long ntok, int d, long long pad_idx, float scale,
                                                           unsigned thr, float drop_scale, unsigned long long seed,
                                                           unsigned long long offset) {
    __shared__ float red[3][NB * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * POS_PER_WG;
    const long long p1 = min(ntok, p0 + POS_PER_WG);
    for (long long p = p0; p < p1; ++p) {
        const unsigned id = keys[p];
        if ((p > 0 && keys[p - 1] == id) || (long long)id == pad_idx) continue;  // not a segment head / nn.Embedding(padding_idx)
        float a[NB][8];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[b][e] = 0.f;
        for (long long q = p + wave; q < ntok && keys[q] == id; q += 4) {
            const long long tok = toks[q];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c0 = b * 512 + lane * 8;
                if (c0 < d) add_row_chunk<T>(a[b], dout, tok, d, c0, scale, thr, drop_scale, seed, offset);
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[wave - 1][b * 512 + lane * 8 + e] = a[b][e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c0 = b * 512 + lane * 8;
                if (c0 >= d) continue;
#pragma unroll
                for (int w = 0; w < 3; ++w)  // fixed order: wave 0 + wave 1 + wave 2 + wave 3
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[b][e] += red[w][c0 + e];
                T* dst = dE + (long long)id * d + c0;
                if constexpr (sizeof(T) == 2) {
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.set(e, a[b][e]);
                    store16<T>(dst, o);
                } else {
                    *reinterpret_cast<float4*>(dst) = float4{a[b][0], a[b][1], a[b][2], a[b][3]};
                    *reinterpret_cast<float4*>(dst + 4) = float4{a[b][4], a[b][5], a[b][6], a[b][7]};
                }
            }
        }
        __syncthreads();
    }
}

int key_bits(long long V) {
    int b = 1;
    while (b < 32 && (1ll << b) < V) ++b;
    return b;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp_bytes(long long ntok, long long V) {
    size_t bytes = 0;
    unsigned* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)ntok, 0u, (unsigned)key_bits(V), (hipStream_t)0);
    return bytes;
}

}  // namespace

// bytes of `workspace` pk_embed_bwd needs for `ntok` tokens over a vocabulary of V rows (sort keys / values in and out +
// the sort's own scratch)
extern "C" size_t pk_embed_bwd_workspace(long long ntok, long long V) {
    if (ntok <= 0) return 256;
    return 4 * align_up((size_t)ntok * 4) + align_up(sort_temp_bytes(ntok, V)) + 256;
}

// dE[V,d] = sum over tokens of dout[tok] * keep/(1-p) * scale into row ids[tok]; row pad_idx and rows without tokens 0.
extern "C" int pk_embed_bwd(const long long* ids, const void* dout, void* dE, void* workspace, size_t ws_bytes,
                            long long ntok, int d, long long V, long long pad_idx, float scale, float drop_p,
                            unsigned long long seed, unsigned long long offset, int dtype, void* stream) {
    PK_CHECK_ARG(dE && (ntok == 0 || (ids && dout)), "pk_embed_bwd: null tensor");
    PK_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 4096, "pk_embed_bwd: embedding width %d (needs a multiple of 8, <= 4096)", d);
    PK_CHECK_ARG(V > 0 && V < (1ll << 31) && ntok < (1ll << 31), "pk_embed_bwd: sizes beyond 2^31");
    PK_CHECK_ARG(((uintptr_t)dout % 16) == 0 && ((uintptr_t)dE % 16) == 0, "pk_embed_bwd: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == PK_F32 ? 4 : 2;
    hipError_t e = hipMemsetAsync(dE, 0, (size_t)V * d * esz, s);
    if (e != hipSuccess) { pk_set_error("pk_embed_bwd: memset: %s", hipGetErrorString(e)); return (int)e; }
    if (ntok == 0) return 0;
    PK_CHECK_ARG(workspace && ws_bytes >= pk_embed_bwd_workspace(ntok, V), "pk_embed_bwd: workspace too small");
    const size_t seg = align_up((size_t)ntok * 4);
    char* w = (char*)workspace;
    unsigned *k_in = (unsigned*)w, *k_out = (unsigned*)(w + seg), *v_in = (unsigned*)(w + 2 * seg), *v_out = (unsigned*)(w + 3 * seg);
    void* temp = w + 4 * seg;
    size_t temp_bytes = sort_temp_bytes(ntok, V);
    hipLaunchKernelGGL(embed_keys_kernel, dim3((unsigned)std::min<long long>(1024, (ntok + 255) / 256)), dim3(256), 0, s, ids,
                       k_in, v_in, ntok, V);
    PK_LAUNCH_CHECK();
    e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (size_t)ntok, 0u, (unsigned)key_bits(V), s);
    if (e != hipSuccess) { pk_set_error("pk_embed_bwd: radix sort: %s", hipGetErrorString(e)); return (int)e; }
    const unsigned thr = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const dim3 grid((unsigned)((ntok + POS_PER_WG - 1) / POS_PER_WG));
    const int nb = (d + 511) / 512;
#define PK_SEG(TT, NBV)                                                                                               \
    hipLaunchKernelGGL((embed_segsum_kernel<TT, NBV>), grid, dim3(256), 0, s, k_out, v_out, (const TT*)dout, (TT*)dE, ntok, d, \
                       pad_idx, scale, thr, ds, seed, offset)
#define PK_SEG_NB(TT)                                      \
    do {                                                   \
        if (nb <= 1) PK_SEG(TT, 1);                        \
        else if (nb <= 2) PK_SEG(TT, 2);                   \
        else if (nb <= 4) PK_SEG(TT, 4);                   \
        else PK_SEG(TT, 8);                                \
    } while (0)
    if (dtype == PK_BF16) PK_SEG_NB(bf16);
    else if (dtype == PK_F16) PK_SEG_NB(f16);
    else if (dtype == PK_F32) PK_SEG_NB(float);
    else PK_CHECK_ARG(false, "pk_embed_bwd: dtype %d not supported", dtype);
#undef PK_SEG_NB
#undef PK_SEG
    PK_LAUNCH_CHECK();
    return 0;
}
