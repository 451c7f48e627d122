// MFMA GEMM for the linear layers of the Transformer hot path (K2/K4/K5/K6 of SURVEY §2c):
//   forward   Y[M,N]  = X[M,K] · W[N,K]ᵀ (+bias, activation, residual)      pasero/models/modules.py:92-96
//   backward  dX[M,K] = dY[M,N] · W[N,K]                                      (B operand "col" form)
//             dW[N,K] = dYᵀ · X  (contraction over the M rows, split-K)       (A and B operands "col" form)
// One kernel template covers all three: C[m,n] = sum_k A(m,k) · B(n,k) where each operand is stored either with
// k contiguous ("row" form) or with its m/n index contiguous ("col" form).
//
// gfx950 design: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each 64x64 = 2x2 MFMA 32x32 tiles),
// bf16: v_mfma_f32_32x32x16_bf16, BK = 64; f32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), BK = 16.
// Tiles are staged global -> VGPR (16 B per lane, coalesced) -> LDS with the next tile's global loads issued before
// the MFMAs of the current one (double-buffered LDS, one barrier per K-step).  Row-form tiles are read back with
// ds_read_b128 (row pitch padded by 16 B: conflict-free), col-form tiles with ds_read_b64_tr_b16 (hardware
// transpose; row pitch = 256 + 64 B so the 4 k-rows of a read land on distinct bank quarters).  The f32 accumulators
// are staged through LDS in the epilogue so that bias / activation / residual are applied on row-contiguous
// 16-byte chunks and C is written fully coalesced.  Workgroup ids are remapped so that each XCD (blockIdx % 8)
// owns a contiguous range of tiles, and tiles are walked in 8-row-panel groups, so the A and B panels a tile
// shares with its neighbours are hits in that XCD's private L2.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int C_PITCH = 132;  // floats; epilogue staging [128][132]

template <typename T> struct Traits;
template <> struct Traits<bf16> {
    static constexpr int BK = 64, EPV = 8, KSTEP = 16;
    static constexpr int ROW_PITCH = BK * 2 + 16;   // bytes, [128][BK] tile
    static constexpr int COL_PITCH = 128 * 2 + 64;  // bytes, [BK][128] tile
};
template <> struct Traits<float> {
    static constexpr int BK = 16, EPV = 4, KSTEP = 2;
    static constexpr int ROW_PITCH = BK * 4 + 4;  // 17 words: ds_read_b32 column reads are conflict-free
    static constexpr int COL_PITCH = 128 * 4;
};

template <typename T, bool COL> struct TileGeom {
    using TR = Traits<T>;
    static constexpr int ROWS = COL ? TR::BK : 128;   // memory rows of the tile
    static constexpr int COLS = COL ? 128 : TR::BK;   // contiguous elements per row
    static constexpr int PITCH = COL ? TR::COL_PITCH : TR::ROW_PITCH;
    static constexpr int BYTES = ROWS * PITCH;
    static constexpr int CPR = COLS / TR::EPV;              // 16-B chunks per row
    static constexpr int NCH = ROWS * CPR / NTHREADS;       // chunks per thread
};

struct EpiParams {
    const void* bias;   // [N] or null
    const void* aux;    // [M, ldaux] or null
    void* preact;       // optional second output: value before the activation
    long long ldaux, ldc, ldpre;
    int act;            // PK_ACT_*
    int mode;           // 0: act(v+bias)   1: act(v+bias) + aux   2: v * act'(aux)
    float alpha;
};

// global -> registers.  (row0, col0) origin inside the matrix; rows >= row_lim / cols >= col_lim read as zero.
template <typename T, bool COL, int NCH>
__device__ __forceinline__ void tile_g2r(Vec16<T> (&v)[NCH], const T* __restrict__ base,
                                         long long ld, long long row0, long long col0, long long row_lim,
                                         long long col_lim, bool vec_ok, int tid) {
    using G = TileGeom<T, COL>;
    static_assert(NCH == G::NCH, "staging register count");
    constexpr int EPV = Traits<T>::EPV;
#pragma unroll
    for (int i = 0; i < G::NCH; ++i) {
        int c = tid + i * NTHREADS;
        long long gr = row0 + c / G::CPR;
        long long gc = col0 + (c % G::CPR) * EPV;
        if (gr < row_lim && gc + EPV <= col_lim && vec_ok) {
            v[i] = load16<T>(base + gr * ld + gc);
        } else {
            v[i].raw = {0, 0, 0, 0};
            if (gr < row_lim) {
#pragma unroll
                for (int e = 0; e < EPV; ++e)
                    if (gc + e < col_lim) v[i].set(e, to_f32<T>(base[gr * ld + gc + e]));
            }
        }
    }
}

template <typename T, bool COL, int NCH>
__device__ __forceinline__ void tile_r2s(const Vec16<T> (&v)[NCH], char* lds, int tid) {
    using G = TileGeom<T, COL>;
    static_assert(NCH == G::NCH, "staging register count");
    constexpr int EPV = Traits<T>::EPV;
#pragma unroll
    for (int i = 0; i < G::NCH; ++i) {
        int c = tid + i * NTHREADS;
        char* p = lds + (c / G::CPR) * G::PITCH + (c % G::CPR) * EPV * (int)sizeof(T);
        if constexpr (G::PITCH % 16 == 0) {
            *reinterpret_cast<decltype(v[i].raw)*>(p) = v[i].raw;
        } else {
            const float* s = reinterpret_cast<const float*>(&v[i].raw);
#pragma unroll
            for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(p)[e] = s[e];
        }
    }
}

// MFMA operand fragment of a 32-row block starting at tile row/col `r0`, k-step `kk` (KSTEP wide), from LDS.
// bf16, 32x32x16: lane l (r = l&31, h = l>>5) holds elements k = 16*kk + 8*h + j, j = 0..7, of row r0 + r.
template <bool COL>
__device__ __forceinline__ bf16x8_t frag_bf16(const char* lds, int r0, int kk, int lane) {
    using G = TileGeom<bf16, COL>;
    if constexpr (!COL) {
        const char* p = lds + (r0 + (lane & 31)) * G::PITCH + (kk * 16 + 8 * (lane >> 5)) * 2;
        return *reinterpret_cast<const bf16x8_t*>(p);
    } else {
        // tile is [k][m]; ds_read_b64_tr_b16: lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a
        // 4x16 block and receives column (lane & 15), rows 0..3 -> 4 consecutive k of one m.
        int q = (lane & 15) >> 2, p4 = lane & 3;
        int col = r0 + 16 * ((lane >> 4) & 1) + 4 * p4;
        int krow = kk * 16 + 8 * (lane >> 5) + q;
        const char* p = lds + krow * G::PITCH + col * 2;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 4 * G::PITCH));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, f);
    }
}
// f32, 32x32x2: lane l holds element k = 2*kk + (l>>5) of row r0 + (l&31)
template <bool COL>
__device__ __forceinline__ float frag_f32(const char* lds, int r0, int kk, int lane) {
    using G = TileGeom<float, COL>;
    int k = 2 * kk + (lane >> 5), r = r0 + (lane & 31);
    if constexpr (!COL) return *reinterpret_cast<const float*>(lds + r * G::PITCH + k * 4);
    else return *reinterpret_cast<const float*>(lds + k * G::PITCH + r * 4);
}

// bijective "each XCD gets a contiguous chunk" remap (blocks b and b+8 share an XCD)
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <typename TO>
__device__ __forceinline__ void epilogue_chunk(float (&v)[8], int n_valid, long long gm, long long gn,
                                               const EpiParams& ep, int dtype_is_bf16, TO* C, bool c_vec_ok,
                                               bool aux_vec_ok) {
    constexpr int EPV = 16 / sizeof(TO);
    // aux/bias/preact share the compute dtype T == TO except for f32 outputs of bf16 GEMMs, handled by caller flag
    (void)dtype_is_bf16;
    float a[8];
    if (ep.aux) {
        const TO* ap = reinterpret_cast<const TO*>(ep.aux) + gm * ep.ldaux + gn;
        if (aux_vec_ok && n_valid == EPV) {
            Vec16<TO> t = load16<TO>(ap);
#pragma unroll
            for (int e = 0; e < EPV; ++e) a[e] = t.get(e);
        } else {
#pragma unroll
            for (int e = 0; e < EPV; ++e) a[e] = e < n_valid ? to_f32<TO>(ap[e]) : 0.f;
        }
    }
    Vec16<TO> pre;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        float x = v[e] * ep.alpha;
        if (ep.mode == 2) {
            x *= act_bwd(ep.act, a[e]);
        } else {
            if (ep.bias && e < n_valid) x += to_f32<TO>(reinterpret_cast<const TO*>(ep.bias)[gn + e]);
            pre.set(e, x);
            x = act_fwd(ep.act, x);
            if (ep.mode == 1) x += a[e];
        }
        v[e] = x;
    }
    TO* cp = C + gm * ep.ldc + gn;
    TO* pp = ep.preact ? reinterpret_cast<TO*>(ep.preact) + gm * ep.ldpre + gn : nullptr;
    if (c_vec_ok && n_valid == EPV) {
        Vec16<TO> o;
#pragma unroll
        for (int e = 0; e < EPV; ++e) o.set(e, v[e]);
        store16<TO>(cp, o);
        if (pp) store16<TO>(pp, pre);
    } else {
#pragma unroll
        for (int e = 0; e < EPV; ++e)
            if (e < n_valid) {
                cp[e] = from_f32<TO>(v[e]);
                if (pp) pp[e] = from_f32<TO>(pre.get(e));
            }
    }
}

template <typename T, bool A_COL, bool B_COL>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(
    const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, float* __restrict__ ws,
    float* __restrict__ asum_ws, T* __restrict__ asum_out, long long M, long long N, long long K, long long lda,
    long long ldb, int kchunk, EpiParams ep, int flags) {
    using TR = Traits<T>;
    using GA = TileGeom<T, A_COL>;
    using GB = TileGeom<T, B_COL>;
    constexpr int STAGE = GA::BYTES + GB::BYTES;
    constexpr int SMEM = (2 * STAGE > BM * C_PITCH * 4) ? 2 * STAGE : BM * C_PITCH * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const bool a_vec = flags & 1, b_vec = flags & 2, c_vec = flags & 4, aux_vec = flags & 8;

    // tile walk: XCD-contiguous, 8-row-panel groups, n fastest inside a group
    const int nt_m = (int)((M + BM - 1) / BM), nt_n = (int)((N + BN - 1) / BN);
    int t = xcd_remap(blockIdx.x, nt_m * nt_n);
    const int GROUP_M = 8;
    int group_size = GROUP_M * nt_n, gid = t / group_size, first_m = gid * GROUP_M;
    int gsz = min(nt_m - first_m, GROUP_M);
    int tile_m = first_m + (t % group_size) % gsz, tile_n = (t % group_size) / gsz;
    const long long m0 = (long long)tile_m * BM, n0 = (long long)tile_n * BN;

    const long long kbeg = (long long)blockIdx.y * kchunk;
    const long long kend = min(K, kbeg + (long long)kchunk);
    const int nk = (int)((kend - kbeg + TR::BK - 1) / TR::BK);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Vec16<T> ra[GA::NCH], rb[GB::NCH];
    // fused bias gradient (weight-gradient GEMMs, A = dY in col form): the sum over k of A(m, k) for this tile's 128
    // m-columns.  Every staging chunk of a thread covers the same EPV columns (NTHREADS % CPR == 0), so each thread
    // keeps EPV running sums; only the tile_n == 0 workgroups do it.
    const bool do_asum = A_COL && (asum_ws || asum_out) && tile_n == 0;
    float asum[TR::EPV];
#pragma unroll
    for (int e = 0; e < TR::EPV; ++e) asum[e] = 0.f;
    auto g2r = [&](int kt) {
        long long k0 = kbeg + (long long)kt * TR::BK;
        if constexpr (A_COL) tile_g2r<T, true>(ra, A, lda, k0, m0, kend, M, a_vec, tid);
        else tile_g2r<T, false>(ra, A, lda, m0, k0, M, kend, a_vec, tid);
        if constexpr (B_COL) tile_g2r<T, true>(rb, B, ldb, k0, n0, kend, N, b_vec, tid);
        else tile_g2r<T, false>(rb, B, ldb, n0, k0, N, kend, b_vec, tid);
    };
    auto r2s = [&](int buf) {
        char* s = smem + buf * STAGE;
        if constexpr (A_COL) {
            if (do_asum) {
#pragma unroll
                for (int i = 0; i < GA::NCH; ++i)
#pragma unroll
                    for (int e = 0; e < TR::EPV; ++e) asum[e] += ra[i].get(e);
            }
        }
        tile_r2s<T, A_COL>(ra, s, tid);
        tile_r2s<T, B_COL>(rb, s + GA::BYTES, tid);
    };

    if (nk > 0) {
        g2r(0);
        r2s(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) g2r(kt + 1);  // next tile's HBM/L2 loads fly under this tile's MFMAs
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sb = sa + GA::BYTES;
#pragma unroll
        for (int kk = 0; kk < TR::BK / TR::KSTEP; ++kk) {
            if constexpr (sizeof(T) == 2) {
                bf16x8_t fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = frag_bf16<A_COL>(sa, wm + 32 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = frag_bf16<B_COL>(sb, wn + 32 * j, kk, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            } else {
                float fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = frag_f32<A_COL>(sa, wm + 32 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = frag_f32<B_COL>(sb, wn + 32 * j, kk, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) r2s((kt + 1) & 1);
        __syncthreads();
    }

    if constexpr (A_COL) {
        if (do_asum) {  // (block-uniform) smem is free here: the k-loop ended with a barrier
            constexpr int CPR = GA::CPR, RL = NTHREADS / CPR;  // threads tid, tid + CPR, ... share a column chunk
            float* red = reinterpret_cast<float*>(smem);       // [RL][128]
#pragma unroll
            for (int e = 0; e < TR::EPV; ++e) red[(tid / CPR) * BM + (tid % CPR) * TR::EPV + e] = asum[e];
            __syncthreads();
            if (tid < BM && m0 + tid < M) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < RL; ++r) t += red[r * BM + tid];
                if (asum_ws) asum_ws[(long long)blockIdx.y * M + m0 + tid] = t;
                else asum_out[m0 + tid] = from_f32<T>(t);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: accumulators -> LDS (f32) -> row-contiguous chunks -> global ----
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int col = wn + 32 * j + (lane & 31);
                cs[row * C_PITCH + col] = acc[i][j][r];
            }
    __syncthreads();

    if (ws) {  // split-K partial: raw f32 slab [gridDim.y][M][N]
        float* slab = ws + (long long)blockIdx.y * M * N;
        const bool ws_vec = (N % 4) == 0;
#pragma unroll 4
        for (int c = tid; c < BM * (BN / 4); c += NTHREADS) {
            int row = c / (BN / 4), col = (c % (BN / 4)) * 4;
            long long gm = m0 + row, gn = n0 + col;
            if (gm >= M || gn >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(cs + row * C_PITCH + col);
            float* p = slab + gm * N + gn;
            if (ws_vec && gn + 4 <= N) *reinterpret_cast<float4*>(p) = v;
            else
                for (int e = 0; e < 4 && gn + e < N; ++e) p[e] = (&v.x)[e];
        }
        return;
    }
    constexpr int EPV = TR::EPV;
#pragma unroll 2
    for (int c = tid; c < BM * (BN / EPV); c += NTHREADS) {
        int row = c / (BN / EPV), col = (c % (BN / EPV)) * EPV;
        long long gm = m0 + row, gn = n0 + col;
        if (gm >= M || gn >= N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < EPV; e += 4) {
            float4 t4 = *reinterpret_cast<const float4*>(cs + row * C_PITCH + col + e);
            v[e] = t4.x; v[e + 1] = t4.y; v[e + 2] = t4.z; v[e + 3] = t4.w;
        }
        int n_valid = (int)min((long long)EPV, N - gn);
        epilogue_chunk<T>(v, n_valid, gm, gn, ep, sizeof(T) == 2, C, c_vec, aux_vec);
    }
}

// split-K reduction + epilogue: C = epi(sum_z ws[z])
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, T* __restrict__ C, long long M, long long N,
                                     int splitk, EpiParams ep, int flags, const float* __restrict__ asum_ws,
                                     T* __restrict__ asum_out) {
    constexpr int EPV = Traits<T>::EPV;
    if (asum_ws) {
        for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M;
             m += (long long)gridDim.x * blockDim.x) {
            float t = 0.f;
            for (int z = 0; z < splitk; ++z) t += asum_ws[(long long)z * M + m];
            asum_out[m] = from_f32<T>(t);
        }
    }
    const bool c_vec = flags & 4, aux_vec = flags & 8;
    long long nchunks_row = (N + EPV - 1) / EPV;
    long long total = M * nchunks_row;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
         c += (long long)gridDim.x * blockDim.x) {
        long long gm = c / nchunks_row, gn = (c % nchunks_row) * EPV;
        int n_valid = (int)min((long long)EPV, N - gn);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int z = 0; z < splitk; ++z) {
            const float* p = ws + ((long long)z * M + gm) * N + gn;
            if ((N % 4) == 0 && n_valid == EPV) {
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    float4 t4 = *reinterpret_cast<const float4*>(p + e);
                    v[e] += t4.x; v[e + 1] += t4.y; v[e + 2] += t4.z; v[e + 3] += t4.w;
                }
            } else {
                for (int e = 0; e < n_valid; ++e) v[e] += p[e];
            }
        }
        epilogue_chunk<T>(v, n_valid, gm, gn, ep, sizeof(T) == 2, C, c_vec, aux_vec);
    }
}

template <typename T>
int launch_gemm(const void* A, const void* B, void* C, long long M, long long N, long long K, long long lda,
                long long ldb, int a_col, int b_col, EpiParams ep, int splitk, void* workspace,
                size_t ws_bytes, void* asum_out, hipStream_t stream) {
    using TR = Traits<T>;
    constexpr int EPV = TR::EPV;
    auto aligned = [&](const void* p, long long ld) { return ((uintptr_t)p % 16) == 0 && (ld % EPV) == 0; };
    int flags = 0;
    if (aligned(A, lda)) flags |= 1;
    if (aligned(B, ldb)) flags |= 2;
    if (aligned(C, ep.ldc) && (!ep.preact || aligned(ep.preact, ep.ldpre))) flags |= 4;
    if (ep.aux && aligned(ep.aux, ep.ldaux)) flags |= 8;
    int nt = (int)(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
    int kchunk = (int)K;
    float* ws = nullptr;
    if (splitk > 1) {
        long long per = (K + splitk - 1) / splitk;
        per = (per + TR::BK - 1) / TR::BK * TR::BK;
        splitk = (int)((K + per - 1) / per);
        kchunk = (int)per;
    }
    float* asum_ws = nullptr;
    if (splitk > 1) {
        size_t need = (size_t)splitk * M * (N + (asum_out ? 1 : 0)) * sizeof(float);
        PK_CHECK_ARG(workspace && ws_bytes >= need, "pk_gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
        ws = (float*)workspace;
        if (asum_out) asum_ws = ws + (size_t)splitk * M * N;
    } else {
        splitk = 1;
    }
    dim3 grid(nt, splitk), block(NTHREADS);
    const T* a = (const T*)A;
    const T* b = (const T*)B;
    T* c = (T*)C;
#define PK_LAUNCH(AC, BC) \
    hipLaunchKernelGGL((gemm_kernel<T, AC, BC>), grid, block, 0, stream, a, b, c, ws, asum_ws, (T*)asum_out, M, N, K, \
                       lda, ldb, kchunk, ep, flags)
    if (!a_col && !b_col) PK_LAUNCH(false, false);
    else if (!a_col && b_col) PK_LAUNCH(false, true);
    else if (a_col && !b_col) PK_LAUNCH(true, false);
    else PK_LAUNCH(true, true);
#undef PK_LAUNCH
    PK_LAUNCH_CHECK();
    if (ws) {
        long long chunks = M * ((N + EPV - 1) / EPV);
        int blocks = (int)min((long long)2048, (chunks + 255) / 256);
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, stream, ws, c, M, N, splitk, ep,
                           flags, (const float*)asum_ws, (T*)asum_out);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" int pk_gemm(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact,
                       long long M, long long N, long long K, long long lda, long long ldb, long long ldc,
                       long long ldaux, long long ldpre, int a_col, int b_col, int act, int mode, float alpha,
                       int dtype, int splitk, void* workspace, size_t ws_bytes, void* asum_out, void* stream) {
    PK_CHECK_ARG(A && B && C, "pk_gemm: null operand");
    PK_CHECK_ARG(!asum_out || a_col, "pk_gemm: asum_out (fused bias gradient) needs A in col form");
    PK_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "pk_gemm: negative size");
    PK_CHECK_ARG(dtype == PK_F32 || dtype == PK_BF16, "pk_gemm: dtype %d not supported", dtype);
    PK_CHECK_ARG(mode >= 0 && mode <= 2, "pk_gemm: bad epilogue mode %d", mode);
    PK_CHECK_ARG(mode == 0 || aux, "pk_gemm: epilogue mode %d needs aux", mode);
    PK_CHECK_ARG(((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), "pk_gemm: too many tiles");
    if (M == 0 || N == 0) return 0;
    EpiParams ep;
    ep.bias = bias; ep.aux = aux; ep.preact = preact;
    ep.ldaux = ldaux; ep.ldc = ldc; ep.ldpre = ldpre;
    ep.act = act; ep.mode = mode; ep.alpha = alpha;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PK_BF16)
        return launch_gemm<bf16>(A, B, C, M, N, K, lda, ldb, a_col, b_col, ep, splitk, workspace, ws_bytes, asum_out, s);
    return launch_gemm<float>(A, B, C, M, N, K, lda, ldb, a_col, b_col, ep, splitk, workspace, ws_bytes, asum_out, s);
}
