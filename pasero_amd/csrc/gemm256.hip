// 256x256-tile bf16 MFMA GEMM — the large-tile variant of gemm.hip for operands that are 16-byte addressable and whose
// contraction length is a multiple of 64 (everything else, and fp32, stays on the 128x128 kernel).
//
// Why a second tile size: the 128x128 kernel is bound by the L2 -> LDS path, not by the matrix cores — it moves
// (128+128)*2 B per 2*128*128 FLOP per k = 1/64 B/FLOP, and the LDS-DMA path delivers ~12 TB/s chip-wide from L2, i.e.
// ~800 TFLOP/s.  A 256x256 tile halves the bytes per FLOP (1/128 B/FLOP).
//
// Structure: 512 threads = 8 waves as 2 (M) x 4 (N), each wave a 128x64 slab = 4x2 v_mfma_f32_32x32x16_bf16 tiles
// (128 fp32 accumulators per lane); BK = 64; A and B K-tiles (32 KiB each) are staged by LDS-DMA
// (global_load_lds_dwordx4, source-side swizzle, see gemm.hip) into a double buffer = 128 KiB, one workgroup per CU.
// The K loop is software-pipelined across tiles exactly like the 128 kernel: behind the one barrier per K-tile the
// fragments of the next tile's first k-step are read and the DMA of the tile after it is issued before the current tile's
// last MFMAs.  Epilogue: four 64-row passes through a 66 KiB fp32 staging buffer, 16-byte coalesced stores with fused
// bias / ReLU / residual / ReLU' (or raw fp32 split-K slabs); fused bias gradient (column sums of A) as in gemm.hip.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "gemm_epi.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NKK = BK / 16;
constexpr int OP_BYTES = 32768, STAGE = 2 * OP_BYTES, SMEM = 2 * STAGE;
constexpr int CP = BN + 4;  // floats, epilogue staging pitch

template <bool COL, int KB = BK> struct G2 {
    static constexpr int ROWS = COL ? KB : 256, ROWB = COL ? 512 : KB * 2, CPR = ROWB / 16;
    static constexpr int BYTES = 256 * KB * 2, PIECES = BYTES / 1024;
    // row form: 16 consecutive rows x one 16-B chunk (a ds_read_b128 lane group) must cover all 64 banks
    __device__ static __forceinline__ int swz(int row) {
        return COL ? ((row & 3) << 2) : (KB == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3));
    }
    __device__ static __forceinline__ int offset(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};

template <bool COL, int NW, int KB = BK>
__device__ __forceinline__ void tile_glds(char* lds, const bf16* __restrict__ base, long long ld, long long row0,
                                          long long col0, long long row_lim, long long col_lim, int wave, int lane) {
    using G = G2<COL, KB>;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void g_void;
#pragma unroll
    for (int i = 0; i < G::PIECES / NW; ++i) {
        const int piece = i * NW + wave;  // pieces of 1 KiB (one wave-instruction) per operand tile
        const int o = piece * 1024 + lane * 16;
        const int row = o / G::ROWB;
        const int chunk = ((o % G::ROWB) >> 4) ^ G::swz(row);
        long long gr = row0 + row, gc = col0 + chunk * 8;
        if (gr >= row_lim) gr = row_lim - 1;  // past the edge: re-read a valid row (feeds only unstored outputs)
        if (gc + 8 > col_lim) gc = col0;
        __builtin_amdgcn_global_load_lds((g_void*)(base + gr * ld + gc), (lds_void*)(lds + piece * 1024), 16, 0, 0);
    }
}

template <bool COL, int KB = BK>
__device__ __forceinline__ bf16x8_t frag(const char* lds, int r0, int kk, int lane) {
    using G = G2<COL, KB>;
    if constexpr (!COL) {
        return *reinterpret_cast<const bf16x8_t*>(lds + G::offset(r0 + (lane & 31), kk * 2 + (lane >> 5)));
    } else {
        int q = (lane & 15) >> 2, p4 = lane & 3;
        int col = r0 + 16 * ((lane >> 4) & 1) + 4 * p4;
        int krow = kk * 16 + 8 * (lane >> 5) + q;
        const char* p = lds + G::offset(krow, col >> 3) + (col & 7) * 2;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 4 * G::ROWB));
        s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, f);
    }
}

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// one 64-row pass of the epilogue: thread t owns 16-byte chunk t % 32 of rows t / 32 + 16*it
template <typename T, int ACT, int MODE, int NW>
__device__ __forceinline__ void epilogue_pass(const float* __restrict__ cs, T* __restrict__ C, const EpiParams& ep,
                                              long long mh, long long n0, long long M, long long N, int tid) {
    const int col = (tid & 31) * 8, r0 = tid >> 5;
    const long long gn = n0 + col;
    if (gn + 8 > N) return;
    float b[8];
    if (MODE != 2 && ep.bias) {
        Vec16<T> bv = load16<T>(reinterpret_cast<const T*>(ep.bias) + gn);
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = bv.get(e);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = 0.f;
    }
    const float alpha = ep.alpha;
    constexpr int RS = NW * 2, NIT = 64 / RS;  // rows per sweep of the workgroup, sweeps per 64-row pass
    Vec16<T> av[NIT];
    if (MODE != 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const long long gm = mh + r0 + RS * it;
            if (gm < M) av[it] = load16<T>(reinterpret_cast<const T*>(ep.aux) + gm * ep.ldaux + gn);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const long long gm = mh + r0 + RS * it;
        if (gm >= M) continue;
        const float* src = cs + (r0 + RS * it) * CP + col;
        const float4 a4 = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
        float x[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = x[e] * alpha;
            if (MODE == 2) {
                if (ACT == PK_ACT_RELU) y = av[it].get(e) > 0.f ? y : 0.f;
            } else {
                y += b[e];
                if (ACT == PK_ACT_RELU) y = fmaxf(y, 0.f);
                if (MODE == 1) y += av[it].get(e);
            }
            x[e] = y;
        }
        typedef __attribute__((ext_vector_type(8))) float f32x8;
        f32x8 f = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
        Vec16<T> o;
        o.raw = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename H16<T>::vec));
        store16_nt<T>(C + gm * ep.ldc + gn, o);
    }
}

// everything after the K loop (all LDS stages are free): fused bias-gradient column sums, then four 64-row passes of the
// accumulators through the fp32 staging buffer with the fused epilogue (or raw split-K slabs)
template <typename T, bool A_COL, int NW>
__device__ __forceinline__ void finish_tile(char* smem, f32x16 (&acc)[4][16 / NW], float (&asum)[8], bool do_asum,
                                            T* __restrict__ C, float* __restrict__ ws, float* __restrict__ asum_ws,
                                            T* __restrict__ asum_out, long long M, long long N, long long m0,
                                            long long n0, int kslab, const EpiParams& ep) {
    constexpr int WAVES_N = NW / 2, TJ = 16 / NW, RS = NW * 2, NIT = 64 / RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = (wave % WAVES_N) * (32 * TJ);
    if constexpr (A_COL) {
        if (do_asum) {
            float* red = reinterpret_cast<float*>(smem);  // [RS][256]
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(tid >> 5) * BM + (tid & 31) * 8 + e] = asum[e];
            __syncthreads();
            if (tid < BM && m0 + tid < M) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < RS; ++r) s += red[r * BM + tid];
                if (asum_ws) asum_ws[(long long)kslab * M + m0 + tid] = s;
                else asum_out[m0 + tid] = from_f32<T>(s);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: four 64-row passes through the fp32 staging buffer ----
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if ((wave / WAVES_N) == (p >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * ii + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        cs[row * CP + wn + 32 * j + (lane & 31)] = acc[(p & 1) * 2 + ii][j][r];
                    }
        }
        __syncthreads();
        const long long mh = m0 + p * 64;
        if (ws) {  // split-K partial: raw fp32 slab [splitk][M][N]
            float* slab = ws + (long long)kslab * M * N;
            const int col = (tid & 31) * 8, r0 = tid >> 5;
            const long long gn = n0 + col;
            if (gn + 8 <= N) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const long long gm = mh + r0 + RS * it;
                    if (gm >= M) continue;
                    const float* src = cs + (r0 + RS * it) * CP + col;
                    float* dst = slab + gm * N + gn;
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
                    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
                }
            }
        } else if (ep.mode == 0) {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 0, NW>(cs, C, ep, mh, n0, M, N, tid);
            else epilogue_pass<T, PK_ACT_NONE, 0, NW>(cs, C, ep, mh, n0, M, N, tid);
        } else if (ep.mode == 1) {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 1, NW>(cs, C, ep, mh, n0, M, N, tid);
            else epilogue_pass<T, PK_ACT_NONE, 1, NW>(cs, C, ep, mh, n0, M, N, tid);
        } else {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 2, NW>(cs, C, ep, mh, n0, M, N, tid);
            else epilogue_pass<T, PK_ACT_NONE, 2, NW>(cs, C, ep, mh, n0, M, N, tid);
        }
        if (p < 3) __syncthreads();
    }
}

template <typename T, bool A_COL, bool B_COL, int NW>
__global__ __launch_bounds__(NW * 64) void gemm256_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                        T* __restrict__ C, float* __restrict__ ws,
                                                        float* __restrict__ asum_ws, T* __restrict__ asum_out,
                                                        long long M, long long N, long long K, long long lda,
                                                        long long ldb, int kchunk, EpiParams ep) {
    using GA = G2<A_COL>;
    using GB = G2<B_COL>;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = NW * 64, WAVES_N = NW / 2, TJ = 16 / NW;  // wave slab: 128 x (32*TJ)
    static_assert(NW == 8 || NW == 4, "8 waves of 128x64 or 4 waves of 128x128");
    constexpr int RS = NW * 2, NIT = 64 / RS;
    const int wm = (wave / WAVES_N) * 128, wn = (wave % WAVES_N) * (32 * TJ);

    const int nt_m = (int)((M + BM - 1) / BM), nt_n = (int)((N + BN - 1) / BN);
    const int lin = xcd_remap(blockIdx.x, gridDim.x);  // slab-major (K-slab, tile) walk: an XCD owns whole K-slabs
    const int kslab = lin / (nt_m * nt_n);
    int t = lin % (nt_m * nt_n);
    const int GROUP_M = nt_n <= 2 ? 8 : 4;
    int group_size = GROUP_M * nt_n, gid = t / group_size, first_m = gid * GROUP_M;
    int gsz = min(nt_m - first_m, GROUP_M);
    int tile_m = first_m + (t % group_size) % gsz, tile_n = (t % group_size) / gsz;
    const long long m0 = (long long)tile_m * BM, n0 = (long long)tile_n * BN;
    const long long kbeg = (long long)kslab * kchunk;
    const long long kend = min(K, kbeg + (long long)kchunk);
    const int nk = (int)((kend - kbeg) / BK);  // the launcher guarantees full K-tiles

    f32x16 acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_asum = A_COL && (asum_ws || asum_out) && tile_n == 0;
    float asum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) asum[e] = 0.f;

    auto dma = [&](int kt) {
        long long k0 = kbeg + (long long)kt * BK;
        char* s = smem + (kt & 1) * STAGE;
        // (col form with M % 8 != 0: the host only sends it here when the rows are padded, lda >= M rounded up to 8 —
        // the chunk straddling the edge is then read whole; its pad columns feed output rows >= M, which are not stored)
        if constexpr (A_COL) tile_glds<true, NW>(s, (const bf16*)A, lda, k0, m0, kend, (M + 7) & ~7LL, wave, lane);
        else tile_glds<false, NW>(s, (const bf16*)A, lda, m0, k0, M, kend, wave, lane);
        if constexpr (B_COL) tile_glds<true, NW>(s + OP_BYTES, (const bf16*)B, ldb, k0, n0, kend, N, wave, lane);
        else tile_glds<false, NW>(s + OP_BYTES, (const bf16*)B, ldb, n0, k0, N, kend, wave, lane);
    };
    if (nk > 0) {
        typename H16<T>::vec fa[2][4], fb[2][TJ];
        dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = __builtin_bit_cast(typename H16<T>::vec, frag<A_COL>(smem, wm + 32 * i, 0, lane));
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[0][j] = __builtin_bit_cast(typename H16<T>::vec, frag<B_COL>(smem + OP_BYTES, wn + 32 * j, 0, lane));
        if (nk > 1) dma(1);
        // One K-tile.  NEXT: another tile follows (barrier + its first fragments behind this tile's k-step NKK-2);
        // DMA: the tile after that is still to be issued.  Compile-time flags keep every k-step in one basic block,
        // so the pinned instruction order below also covers the step with the barrier.
        auto tile = [&](int kt, auto next_c, auto dma_c) {
            constexpr bool NEXT = decltype(next_c)::value, DMA = decltype(dma_c)::value;
            const char* sa = smem + (kt & 1) * STAGE;
            const char* sb = sa + OP_BYTES;
            if constexpr (A_COL) {
                if (do_asum) {  // thread owns column chunk tid % 32 of the [64][256] A tile, rows tid / 32 + RS i
#pragma unroll
                    for (int i = 0; i < NIT; ++i) {
                        Vec16<T> v;
                        v.raw = *reinterpret_cast<const uint4*>(sa + GA::offset((tid >> 5) + RS * i, tid & 31));
#pragma unroll
                        for (int e = 0; e < 8; ++e) asum[e] += v.get(e);
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < NKK) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[nxt][i] = __builtin_bit_cast(typename H16<T>::vec, frag<A_COL>(sa, wm + 32 * i, kk + 1, lane));
#pragma unroll
                    for (int j = 0; j < TJ; ++j) fb[nxt][j] = __builtin_bit_cast(typename H16<T>::vec, frag<B_COL>(sb, wn + 32 * j, kk + 1, lane));
                } else if constexpr (NEXT) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile kt+1 landed, LDS reads returned
                    __builtin_amdgcn_s_barrier();
                    const char* na = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[nxt][i] = __builtin_bit_cast(typename H16<T>::vec, frag<A_COL>(na, wm + 32 * i, 0, lane));
#pragma unroll
                    for (int j = 0; j < TJ; ++j) fb[nxt][j] = __builtin_bit_cast(typename H16<T>::vec, frag<B_COL>(na + OP_BYTES, wn + 32 * j, 0, lane));
#if !defined(PK_ABLATE256) || PK_ABLATE256 != 1
                    if constexpr (DMA) dma(kt + 2);
#endif
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = H16<T>::mfma(fa[cur][i], fb[cur][j], acc[i][j]);
#ifndef PK_NO_SCHED
                // Instruction order of this k-step, pinned: hipcc otherwise sinks the next step's fragment reads to
                // just in front of their first use (lgkmcnt(0) then stalls every wave for an LDS round trip, several
                // times per K-tile).  One LDS read (two for transposed operands) and one DMA issue per MFMA gap.
                if constexpr (NW == 8) {
                    constexpr int RA = A_COL ? 2 : 1, RB = B_COL ? 2 : 1;  // DS reads per fragment
#define PK_GAP(R) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
#define PK_GAPV(R) PK_GAP(R) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (kk + 1 < NKK || (NEXT && !DMA)) {
                        PK_GAP(RA) PK_GAP(RA) PK_GAP(RA) PK_GAP(RA) PK_GAP(RB) PK_GAP(RB)
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    } else if constexpr (NEXT && DMA) {
                        PK_GAPV(RA) PK_GAPV(RA) PK_GAPV(RA) PK_GAPV(RA) PK_GAPV(RB) PK_GAPV(RB)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
#undef PK_GAPV
#undef PK_GAP
                }
#endif
            }
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) tile(kt++, std::true_type{}, std::false_type{});
        tile(kt, std::false_type{}, std::false_type{});
    }
    __syncthreads();

    finish_tile<T, A_COL, NW>(smem, acc, asum, do_asum, C, ws, asum_ws, asum_out, M, N, m0, n0, kslab, ep);
}

}  // namespace

// Returns 1 if the GEMM was launched on the 256x256 kernel, 0 if the shape / epilogue is not eligible (the caller then
// uses the 128x128 kernel), or a negative / hip error code.
extern "C" int pk_gemm256_launch(const void* A, const void* B, void* C, float* ws, float* asum_ws, void* asum_out,
                                 long long M, long long N, long long K, long long lda, long long ldb, int a_col,
                                 int b_col, int kchunk, int splitk, EpiParams ep, int dtype, void* stream) {
    // 8 waves of 128x64 (default), or 4 waves of 128x128 with the accumulators in AGPRs (PK_GEMM256_NW=4): half the LDS
    // fragment traffic per MFMA but half the waves to hide prologue / epilogue.  Measured: K = 2048 row,row 887 vs 811
    // TFLOP/s in isolation, K = 512 shapes and col-form operands 5-10 % slower, no difference in the training step.
    static const int nw = getenv("PK_GEMM256_NW") ? atoi(getenv("PK_GEMM256_NW")) : 8;
    dim3 grid((unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN) * splitk)), block(nw * 64);
    hipStream_t s = (hipStream_t)stream;
#define PK_K(TT, AC, BC, W)                                                                                        \
    hipLaunchKernelGGL((gemm256_kernel<TT, AC, BC, W>), grid, block, 0, s, (const TT*)A, (const TT*)B, (TT*)C, ws,  \
                       asum_ws, (TT*)asum_out, M, N, K, lda, ldb, kchunk, ep)
#define PK_L(AC, BC)                                                       \
    do {                                                                   \
        if (dtype == PK_F16) {                                             \
            if (nw == 4) PK_K(f16, AC, BC, 4); else PK_K(f16, AC, BC, 8);  \
        } else {                                                           \
            if (nw == 4) PK_K(bf16, AC, BC, 4); else PK_K(bf16, AC, BC, 8); \
        }                                                                  \
    } while (0)
    if (!a_col && !b_col) PK_L(false, false);
    else if (!a_col && b_col) PK_L(false, true);
    else if (a_col && !b_col) PK_L(true, false);
    else PK_L(true, true);
#undef PK_L
#undef PK_K
    PK_LAUNCH_CHECK();
    return 1;
}
