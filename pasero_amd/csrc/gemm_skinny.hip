// bf16 GEMM for a handful of rows (M <= 256: one decoding step, modules.py Linear on (B, 1, d) inputs; up to 1024 rows for
// outputs of <= 512 columns: a short decoder batch): y = act(x W^T + b) (+ aux).
//
// The 128x128 / 256x256 kernels are built for throughput: with M = 64 they launch 4-16 workgroups that each walk the
// whole K loop through LDS, ~18 us per call at any size, and a decoding step is 37 such calls.  Here the problem is
// latency, not bandwidth (the whole weight matrix is 0.5-2 MB), so the shape is different:
//   * one workgroup = a 64 x 32 output tile; its 4 waves split K in four and each keeps two 32x32 accumulators
//     (v_mfma_f32_32x32x16_bf16), so N = 512 already gives 16 workgroups x 4 waves with K/4 each;
//   * no LDS staging: every operand element is used once per workgroup, fragments are loaded straight from global
//     memory in MFMA layout (16 B per lane, k contiguous: lane (r, h) holds row r, k = 8h..8h+7 of the k-step) with a
//     whole unrolled batch of loads in flight before the first MFMA;
//   * the four partial tiles meet in LDS (fp32), then bias / activation / residual and 16-byte stores.
#include <cstdlib>
#include "common.h"
#include "gemm_epi.h"


namespace {

constexpr int SK_BM = 64, SK_BN = 32, SK_WAVES = 4, SK_PITCH = SK_BN + 1;

template <typename T>
__device__ __forceinline__ typename H16<T>::vec ldfrag(const T* __restrict__ p, bool ok) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) v = *reinterpret_cast<const uint4*>(p);
    return __builtin_bit_cast(typename H16<T>::vec, v);
}

template <typename T, int ACT, int MODE>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          T* __restrict__ C, long long M, long long N, long long K,
                                                          long long lda, long long ldb, EpiParams ep) {
    __shared__ float red[SK_WAVES][SK_BM][SK_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long n0 = (long long)blockIdx.x * SK_BN, m0 = (long long)blockIdx.y * SK_BM;
    const long long kq = K / SK_WAVES, kb = wave * kq;  // the launcher guarantees K % 64 == 0
    const bool okb = n0 + r < N, oka0 = m0 + r < M, oka1 = m0 + 32 + r < M;
    const T* pa0 = A + (m0 + r) * lda + kb + 8 * h;
    const T* pa1 = pa0 + 32 * lda;
    const T* pb = W + (n0 + r) * ldb + kb + 8 * h;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    constexpr int U = 8;  // k-steps per batch: 24 x 16-byte loads per lane in flight
    for (long long k = 0; k < kq; k += 16 * U) {
        typename H16<T>::vec fa0[U], fa1[U], fb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = k + 16 * u < kq;
            fa0[u] = ldfrag(pa0 + k + 16 * u, in && oka0);
            fa1[u] = ldfrag(pa1 + k + 16 * u, in && oka1);
            fb[u] = ldfrag(pb + k + 16 * u, in && okb);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc0 = H16<T>::mfma(fa0[u], fb[u], acc0);
            acc1 = H16<T>::mfma(fa1[u], fb[u], acc1);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        red[wave][row][r] = acc0[i];
        red[wave][32 + row][r] = acc1[i];
    }
    __syncthreads();
    // thread -> row tid/4, 8 consecutive columns
    const int row = tid >> 2, c0 = (tid & 3) * 8;
    const long long gm = m0 + row, gn = n0 + c0;
    if (gm >= M || gn >= N) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        v[e] = (red[0][row][c0 + e] + red[1][row][c0 + e]) + (red[2][row][c0 + e] + red[3][row][c0 + e]);
    Vec16<T> bv, av;
    const bool full = gn + 8 <= N;
    if (ep.bias && full) bv = load16<T>(reinterpret_cast<const T*>(ep.bias) + gn);
    if (MODE != 0 && full) av = load16<T>(reinterpret_cast<const T*>(ep.aux) + gm * ep.ldaux + gn);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float y = v[e] * ep.alpha;
        if (MODE == 2) {  // the act'-mask epilogue (pk_gemm mode 2: C = alpha * A B^T * act'(aux), no bias) — an adapter's dA
            const float a = full ? av.get(e) : (gn + e < N ? to_f32(reinterpret_cast<const T*>(ep.aux)[gm * ep.ldaux + gn + e]) : 0.f);
            y *= act_bwd_t<T>(ACT, a);
        } else {
            if (ep.bias) y += full ? bv.get(e) : (gn + e < N ? to_f32(reinterpret_cast<const T*>(ep.bias)[gn + e]) : 0.f);
            y = act_fwd_t<T>(ACT, y);
            if (MODE == 1) y += full ? av.get(e) : (gn + e < N ? to_f32(reinterpret_cast<const T*>(ep.aux)[gm * ep.ldaux + gn + e]) : 0.f);
        }
        v[e] = y;
    }
    o = vec16_pack<T>(v);
    if (full) {
        store16<T>(C + gm * ep.ldc + gn, o);
    } else {
        for (int e = 0; e < 8 && gn + e < N; ++e) C[gm * ep.ldc + gn + e] = from_f32<T>(o.get(e));
    }
}

}  // namespace

// Returns 1 if launched, 0 if the call is not eligible (the caller falls through to the tiled kernels).
extern "C" int pk_gemm_skinny_launch(const void* A, const void* B, void* C, long long M, long long N, long long K,
                                     long long lda, long long ldb, EpiParams ep, int dtype, void* stream) {
    const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && lda % 8 == 0 &&
                         ldb % 8 == 0 && ep.ldc % 8 == 0 && (!ep.bias || (uintptr_t)ep.bias % 16 == 0) &&
                         (ep.mode == 0 || ((uintptr_t)ep.aux % 16 == 0 && ep.ldaux % 8 == 0));
    static const long long sk_m2 = [] { const char* e = getenv("PK_SKINNY_M2"); return e ? atoll(e) : 1024LL; }();
    if (!aligned || (M > sk_m2 && N > 64) || M > 64LL * 65535 || K % 64 != 0 || K <= 0 || ep.preact || ep.mode > 2) return 0;
    if (ep.mode == 2 && ep.bias) return 0;
    dim3 grid((unsigned)((N + SK_BN - 1) / SK_BN), (unsigned)((M + SK_BM - 1) / SK_BM)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define SK_T(TT, ACT, MD) \
    hipLaunchKernelGGL((gemm_skinny_kernel<TT, ACT, MD>), grid, block, 0, s, (const TT*)A, (const TT*)B, (TT*)C, M, N, K, lda, ldb, ep)
#define SK_L(ACT)                                                       \
    do {                                                                \
        if (dtype == PK_F16) {                                          \
            if (ep.mode == 0) SK_T(f16, ACT, 0); else if (ep.mode == 1) SK_T(f16, ACT, 1); else SK_T(f16, ACT, 2); \
        } else {                                                        \
            if (ep.mode == 0) SK_T(bf16, ACT, 0); else if (ep.mode == 1) SK_T(bf16, ACT, 1); else SK_T(bf16, ACT, 2); \
        }                                                               \
    } while (0)
    switch (ep.act) {
        case PK_ACT_NONE: SK_L(PK_ACT_NONE); break;
        case PK_ACT_RELU: SK_L(PK_ACT_RELU); break;
        case PK_ACT_GELU: SK_L(PK_ACT_GELU); break;
        case PK_ACT_GELU_TANH: SK_L(PK_ACT_GELU_TANH); break;
        case PK_ACT_SILU: SK_L(PK_ACT_SILU); break;
        default: return 0;
    }
#undef SK_L
#undef SK_T
    PK_LAUNCH_CHECK();
    return 1;
}
