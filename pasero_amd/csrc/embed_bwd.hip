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
// One workgroup per POS_PER_WG = 64 sorted positions.  A RUN is a maximal sequence of equal ids, clipped to the range of
// the workgroup: short runs (the common case: a vocabulary row seen a few times) are summed by ONE wave each, in position
// order, without LDS or barriers (wave w takes the runs number w, w + 4, ... of the range); runs longer than SHORT_RUN are
// summed by the four waves together (wave w takes positions w, w + 4, ...; partial sums combined through LDS in wave
// order).  A run that ends inside the range it began in is written to its row, rounded once.  A run that crosses a range
// boundary — a frequent token: BOS / EOS, punctuation, a language tag — leaves fp32 partial sums, one per workgroup it
// touches, which embed_fixup_kernel adds in workgroup order.  So the time of the kernel does not depend on how the ids
// are distributed (the first version gave a whole run to the workgroup of its head: a token seen 7 000 times in a batch of
// 32 768 cost 1.4 ms), and every row is still a fixed-order sum.
constexpr int SHORT_RUN = 16;
constexpr int FLAG_BEGIN = 1, FLAG_END = 2, FLAG_SPAN = 4;  // partial of a run cut at the range start / end / both

// row `id` of dE (columns c0 .. c0 + 8) added to a[0..8): the accumulate form (pk_embed_bwd_acc) sums into a gradient that is
// already there — the tied projection's dW — before the one rounding
template <typename T>
__device__ __forceinline__ void add_existing(float (&a)[8], const T* __restrict__ src) {
    if constexpr (sizeof(T) == 2) {
        Vec16<T> v = load16<T>(src);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += v.get(e);
    } else {
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w; a[4] += hi.x; a[5] += hi.y; a[6] += hi.z; a[7] += hi.w;
    }
}

template <typename T, int NB, bool ACC>
__global__ __launch_bounds__(256) void embed_segsum_kernel(const unsigned* __restrict__ keys,
                                                           const unsigned* __restrict__ toks,
                                                           const T* __restrict__ dout, T* __restrict__ dE,
                                                           float* __restrict__ part, int* __restrict__ flags,
                                                           long long ntok, int d, long long pad_idx, float scale,
                                                           unsigned thr, float drop_scale, unsigned long long seed,
                                                           unsigned long long offset) {
    __shared__ float red[3][NB * 512];
    __shared__ int wg_flags;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * POS_PER_WG;
    const int n_valid = (int)min((long long)POS_PER_WG, ntok - p0);
    if (threadIdx.x == 0) wg_flags = 0;
    const unsigned kv = lane < n_valid ? keys[p0 + lane] : 0xffffffffu;
    const unsigned tv = lane < n_valid ? toks[p0 + lane] : 0u;
    const unsigned kup = __shfl_up(kv, 1);
    unsigned long long starts = __ballot(lane < n_valid && (lane == 0 || kv != kup));
    const bool cut0 = p0 > 0 && keys[p0 - 1] == (unsigned)__builtin_amdgcn_readfirstlane(kv);
    const unsigned klast = (unsigned)__builtin_amdgcn_readlane(kv, n_valid - 1);
    const bool cut1 = p0 + n_valid < ntok && keys[p0 + n_valid] == klast;
    __syncthreads();

    auto finish = [&](float (&a)[NB][8], int s, int e, unsigned id) {  // (called by ONE wave per run)
        const bool cb = s == 0 && cut0, ce = e == n_valid && cut1;
        if (!cb && !ce) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c0 = b * 512 + lane * 8;
                if (c0 >= d) continue;
                T* dst = dE + (long long)id * d + c0;
                if constexpr (ACC) add_existing<T>(a[b], dst);
                if constexpr (sizeof(T) == 2) {
                    Vec16<T> o;
#pragma unroll
                    for (int x = 0; x < 8; ++x) o.set(x, a[b][x]);
                    store16<T>(dst, o);
                } else {
                    *reinterpret_cast<float4*>(dst) = float4{a[b][0], a[b][1], a[b][2], a[b][3]};
                    *reinterpret_cast<float4*>(dst + 4) = float4{a[b][4], a[b][5], a[b][6], a[b][7]};
                }
            }
            return;
        }
        float* dst = part + ((long long)blockIdx.x * 2 + (cb ? 0 : 1)) * d;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c0 = b * 512 + lane * 8;
            if (c0 >= d) continue;
            *reinterpret_cast<float4*>(dst + c0) = float4{a[b][0], a[b][1], a[b][2], a[b][3]};
            *reinterpret_cast<float4*>(dst + c0 + 4) = float4{a[b][4], a[b][5], a[b][6], a[b][7]};
        }
        if (lane == 0) atomicOr(&wg_flags, cb ? (ce ? FLAG_BEGIN | FLAG_SPAN : FLAG_BEGIN) : FLAG_END);
    };

    int r = 0;
    while (starts) {  // (uniform over the workgroup: every wave walks the same list of runs)
        const int s = __builtin_ctzll(starts);
        starts &= starts - 1;
        const int e = starts ? __builtin_ctzll(starts) : n_valid;
        const unsigned id = (unsigned)__builtin_amdgcn_readlane(kv, s);
        const bool is_long = e - s > SHORT_RUN;
        const int mine = r++ & 3;
        if ((long long)id == pad_idx) continue;  // nn.Embedding(padding_idx): its row stays zero
        if (!is_long && mine != wave) continue;
        float a[NB][8];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int x = 0; x < 8; ++x) a[b][x] = 0.f;
        for (int q = is_long ? s + wave : s; q < e; q += is_long ? 4 : 1) {
            const long long tok = (long long)(unsigned)__builtin_amdgcn_readlane(tv, q);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c0 = b * 512 + lane * 8;
                if (c0 < d) add_row_chunk<T>(a[b], dout, tok, d, c0, scale, thr, drop_scale, seed, offset);
            }
        }
        if (!is_long) {
            finish(a, s, e, id);
            continue;
        }
        if (wave > 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int x = 0; x < 8; ++x) red[wave - 1][b * 512 + lane * 8 + x] = a[b][x];
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
                    for (int x = 0; x < 8; ++x) a[b][x] += red[w][c0 + x];
            }
            finish(a, s, e, id);
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = wg_flags;
}

// rows whose run crosses workgroup ranges: the workgroup in whose range the run BEGINS adds the partial sums of the
// ranges it runs through, in range order, and writes the row (one wave per chain; lane = 8 columns per block of 512)
template <typename T, int NB, bool ACC>
__global__ __launch_bounds__(64) void embed_fixup_kernel(const unsigned* __restrict__ keys, const float* __restrict__ part,
                                                        const int* __restrict__ flags, T* __restrict__ dE, int nwg,
                                                        long long ntok, int d, long long pad_idx) {
    const int w = blockIdx.x, lane = threadIdx.x;
    if (!(flags[w] & FLAG_END)) return;  // no run that begins here and goes on
    const long long last = min(ntok, (long long)(w + 1) * POS_PER_WG) - 1;
    const unsigned id = keys[last];
    float a[NB][8];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c0 = b * 512 + lane * 8;
        if (c0 < d) {
            const float* src = part + ((long long)w * 2 + 1) * d + c0;
            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            a[b][0] = lo.x; a[b][1] = lo.y; a[b][2] = lo.z; a[b][3] = lo.w;
            a[b][4] = hi.x; a[b][5] = hi.y; a[b][6] = hi.z; a[b][7] = hi.w;
        }
    }
    for (int j = w + 1; j < nwg; ++j) {
        const int f = flags[j];
        if (!(f & FLAG_BEGIN)) break;  // (cannot happen: the run was cut at the end of range j - 1)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c0 = b * 512 + lane * 8;
            if (c0 < d) {
                const float* src = part + ((long long)j * 2) * d + c0;
                const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                a[b][0] += lo.x; a[b][1] += lo.y; a[b][2] += lo.z; a[b][3] += lo.w;
                a[b][4] += hi.x; a[b][5] += hi.y; a[b][6] += hi.z; a[b][7] += hi.w;
            }
        }
        if (!(f & FLAG_SPAN)) break;
    }
    if ((long long)id == pad_idx) return;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c0 = b * 512 + lane * 8;
        if (c0 >= d) continue;
        T* dst = dE + (long long)id * d + c0;
        if constexpr (ACC) add_existing<T>(a[b], dst);
        if constexpr (sizeof(T) == 2) {
            Vec16<T> o;
#pragma unroll
            for (int x = 0; x < 8; ++x) o.set(x, a[b][x]);
            store16<T>(dst, o);
        } else {
            *reinterpret_cast<float4*>(dst) = float4{a[b][0], a[b][1], a[b][2], a[b][3]};
            *reinterpret_cast<float4*>(dst + 4) = float4{a[b][4], a[b][5], a[b][6], a[b][7]};
        }
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
extern "C" size_t pk_embed_bwd_workspace(long long ntok, long long V, int d) {
    if (ntok <= 0) return 256;
    const size_t nwg = (size_t)((ntok + POS_PER_WG - 1) / POS_PER_WG);
    return 4 * align_up((size_t)ntok * 4) + align_up(sort_temp_bytes(ntok, V)) + align_up(nwg * 2 * (size_t)d * 4) +
           align_up(nwg * 4) + 256;
}

namespace {
// dE[V,d] = sum over tokens of dout[tok] * keep/(1-p) * scale into row ids[tok]; row pad_idx and rows without tokens 0 —
// or, `acc`: the same sums added INTO dE (rows without tokens and row pad_idx untouched, every touched row = fp32(existing)
// + its fixed-order sum, rounded once)
int embed_bwd_impl(const long long* ids, const void* dout, void* dE, void* workspace, size_t ws_bytes,
                   long long ntok, int d, long long V, long long pad_idx, float scale, float drop_p,
                   unsigned long long seed, unsigned long long offset, int dtype, void* stream, bool acc) {
    PK_CHECK_ARG(dE && (ntok == 0 || (ids && dout)), "pk_embed_bwd: null tensor");
    PK_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 4096, "pk_embed_bwd: embedding width %d (needs a multiple of 8, <= 4096)", d);
    PK_CHECK_ARG(V > 0 && V < (1ll << 31) && ntok < (1ll << 31), "pk_embed_bwd: sizes beyond 2^31");
    PK_CHECK_ARG(((uintptr_t)dout % 16) == 0 && ((uintptr_t)dE % 16) == 0, "pk_embed_bwd: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == PK_F32 ? 4 : 2;
    hipError_t e = acc ? hipSuccess : hipMemsetAsync(dE, 0, (size_t)V * d * esz, s);
    if (e != hipSuccess) { pk_set_error("pk_embed_bwd: memset: %s", hipGetErrorString(e)); return (int)e; }
    if (ntok == 0) return 0;
    PK_CHECK_ARG(workspace && ws_bytes >= pk_embed_bwd_workspace(ntok, V, d), "pk_embed_bwd: workspace too small");
    const size_t seg = align_up((size_t)ntok * 4);
    char* w = (char*)workspace;
    unsigned *k_in = (unsigned*)w, *k_out = (unsigned*)(w + seg), *v_in = (unsigned*)(w + 2 * seg), *v_out = (unsigned*)(w + 3 * seg);
    void* temp = w + 4 * seg;
    size_t temp_bytes = sort_temp_bytes(ntok, V);
    const int nwg = (int)((ntok + POS_PER_WG - 1) / POS_PER_WG);
    float* part = (float*)(w + 4 * seg + align_up(temp_bytes));            // [nwg][2][d] partial sums of cut runs
    int* flags = (int*)((char*)part + align_up((size_t)nwg * 2 * d * 4));  // [nwg]
    hipLaunchKernelGGL(embed_keys_kernel, dim3((unsigned)std::min<long long>(1024, (ntok + 255) / 256)), dim3(256), 0, s, ids,
                       k_in, v_in, ntok, V);
    PK_LAUNCH_CHECK();
    e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (size_t)ntok, 0u, (unsigned)key_bits(V), s);
    if (e != hipSuccess) { pk_set_error("pk_embed_bwd: radix sort: %s", hipGetErrorString(e)); return (int)e; }
    const unsigned thr = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const dim3 grid((unsigned)nwg);
    const int nb = (d + 511) / 512;
#define PK_SEG2(TT, NBV, AC)                                                                                          \
    do {                                                                                                              \
        hipLaunchKernelGGL((embed_segsum_kernel<TT, NBV, AC>), grid, dim3(256), 0, s, k_out, v_out, (const TT*)dout,  \
                           (TT*)dE, part, flags, ntok, d, pad_idx, scale, thr, ds, seed, offset);                     \
        hipLaunchKernelGGL((embed_fixup_kernel<TT, NBV, AC>), grid, dim3(64), 0, s, k_out, part, flags, (TT*)dE, nwg, \
                           ntok, d, pad_idx);                                                                         \
    } while (0)
#define PK_SEG(TT, NBV)                       \
    do {                                      \
        if (acc) PK_SEG2(TT, NBV, true);      \
        else PK_SEG2(TT, NBV, false);         \
    } while (0)
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
#undef PK_SEG2
    PK_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int pk_embed_bwd(const long long* ids, const void* dout, void* dE, void* workspace, size_t ws_bytes,
                            long long ntok, int d, long long V, long long pad_idx, float scale, float drop_p,
                            unsigned long long seed, unsigned long long offset, int dtype, void* stream) {
    return embed_bwd_impl(ids, dout, dE, workspace, ws_bytes, ntok, d, V, pad_idx, scale, drop_p, seed, offset, dtype, stream,
                          false);
}

extern "C" int pk_embed_bwd_acc(const long long* ids, const void* dout, void* dE, void* workspace, size_t ws_bytes,
                                long long ntok, int d, long long V, long long pad_idx, float scale, float drop_p,
                                unsigned long long seed, unsigned long long offset, int dtype, void* stream) {
    return embed_bwd_impl(ids, dout, dE, workspace, ws_bytes, ntok, d, V, pad_idx, scale, drop_p, seed, offset, dtype, stream,
                          true);
}
