// MFMA GEMM for the linear layers of the Transformer hot path (K2/K4/K5/K6 of SURVEY §2c):
//   forward   Y[M,N]  = X[M,K] · W[N,K]ᵀ (+bias, activation, residual)      pasero/models/modules.py:92-96
//   backward  dX[M,K] = dY[M,N] · W[N,K]                                      (B operand "col" form)
//             dW[N,K] = dYᵀ · X  (contraction over the M rows, split-K)       (A and B operands "col" form)
// One kernel template covers all three: C[m,n] = sum_k A(m,k) · B(n,k) where each operand is stored either with
// k contiguous ("row" form) or with its m/n index contiguous ("col" form).
//
// gfx950 design: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each 64x64 = 2x2 MFMA 32x32 tiles),
// bf16 / fp16: v_mfma_f32_32x32x16_{bf16,f16}, BK = 64; f32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), BK = 16.
// 16-bit staging is LDS-DMA: `global_load_lds_dwordx4` writes each wave-instruction's 64 x 16 B straight into a linear
// 1-KiB piece of the LDS tile (no VGPR round trip, no ds_write), the next K-tile's DMA is in flight under the current
// tile's MFMAs (double-buffered LDS, one vmcnt(0)+barrier per K-step).  Because the DMA destination is lane-linear, the
// bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the read: row-form tiles
// ([128][128 B]) XOR the 16-B chunk index with (row>>1)&7 so ds_read_b128 is conflict-free; col-form tiles
// ([64][256 B], read with the hardware-transposing ds_read_b64_tr_b16) XOR it with (row&3)<<2 so the 4 k-rows of a
// transposed read land on distinct bank quarters.  K tails / unaligned operands take a register-staged path into the
// same LDS image.  fp32 keeps register staging with padded rows.  The f32 accumulators are staged through LDS in the
// epilogue so that bias / activation / residual are applied on row-contiguous 16-byte chunks and C is written fully
// coalesced.  Workgroup ids are remapped so that each XCD (blockIdx % 8) owns a contiguous range of tiles, and tiles
// are walked in 8-row-panel groups, so the A and B panels a tile shares with its neighbours are hits in that XCD's
// private L2.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm_epi.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

extern "C" int pk_gemm256_launch(const void* A, const void* B, void* C, float* ws, float* asum_ws, void* asum_out,
                                 long long M, long long N, long long K, long long lda, long long ldb, int a_col,
                                 int b_col, int kchunk, int splitk, EpiParams ep, int dtype, void* stream);
extern "C" int pk_gemm8p_is_pw(long long M, long long N, long long K, long long lda, long long ldb, int a_col, int b_col,
                               int splitk, int has_ws, int has_asum, const EpiParams* ep);
extern "C" int pk_gemm8p_is_pt(long long M, long long N, long long K, long long lda, long long ldb, int a_col, int b_col,
                               int splitk, int has_ws, int has_asum, const EpiParams* ep);
extern "C" int pk_gemm8p_launch(const void* A, const void* B, void* C, float* ws, float* asum_ws, void* asum_out,
                                long long M, long long N, long long K, long long lda, long long ldb, int a_col,
                                int b_col, int kchunk, int splitk, EpiParams ep, int dtype, void* stream);
extern "C" int pk_gemm8p_eligible(long long M, long long N, long long K, long long lda, long long ldb, int a_col,
                                  int b_col, int want_asum);
extern "C" int pk_gemm8p_group_eligible(const PkWgradProblem* q);
extern "C" int pk_gemm8p_group_plan(const PkWgradProblem* p, int n, size_t* ws_bytes, int* workgroups, int* slabs);
extern "C" int pk_gemm8p_group_map(const PkWgradProblem* p, int n, int* out, int cap);
extern "C" int pk_gemm8p_group_launch(const PkWgradProblem* p, int n, int dtype, float* workspace, void* stream);
extern "C" int pk_gemm8p_group_reduce(const PkWgradProblem* p, int n, int dtype, float* workspace, void* stream);
extern "C" int pk_gemmln_spec(int has_residual, float drop_p, long long K);
extern "C" int pk_gemmln_launch(const void* A, const void* W, const void* bias, const void* residual, const void* gamma,
                                const void* beta, void* z_out, void* y_out, float* mean, float* rstd, long long M,
                                long long N, long long K, long long lda, long long ldb, long long ldr, float eps,
                                float drop_p, unsigned long long seed, unsigned long long offset, int dtype,
                                void* stream);
extern "C" int pk_gemm_skinny_launch(const void* A, const void* B, void* C, long long M, long long N, long long K,
                                     long long lda, long long ldb, EpiParams ep, int dtype, void* stream);

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int C_PITCH = 132;  // floats; epilogue staging [128][132]
constexpr int EPI_PASSES = 1;

template <typename T> struct Traits;
#ifndef PK_BK
#define PK_BK 64
#endif
#ifndef PK_NS
#define PK_NS 2
#endif
template <> struct Traits<bf16> {
    static constexpr int BK = PK_BK, EPV = 8, KSTEP = 16;
    static constexpr bool GLDS = true;
    static constexpr int NSTAGE = PK_NS;  // LDS stages (the pipelined loop below is written for 2)
};
template <> struct Traits<f16> : Traits<bf16> {};  // same bytes, same tiles; only the MFMA instruction differs
template <> struct Traits<float> {
    static constexpr int BK = 16, EPV = 4, KSTEP = 2;
    static constexpr bool GLDS = false;
    static constexpr int NSTAGE = 2;
};

// LDS image of one operand tile.  ROWS x COLS elements as they lie in memory (COLS contiguous):
//   row form: 128 x BK (k contiguous)      col form: BK x 128 (m/n contiguous)
// offset(row, chunk) = byte offset of 16-byte chunk `chunk` of row `row`.
template <typename T, bool COL> struct TileGeom;
template <bool COL> struct TileGeom<bf16, COL> {
    static constexpr int BKB = Traits<bf16>::BK;
    static constexpr int ROWS = COL ? BKB : 128, COLS = COL ? 128 : BKB;
    static constexpr int ROWB = COLS * 2;  // 64/128 or 256 bytes, linear (LDS-DMA pieces are 1 KiB of whole rows)
    static constexpr int BYTES = ROWS * ROWB;
    static constexpr int PIECES = BYTES / 1024;
    static constexpr int CPR = COLS / 8, NCH = ROWS * CPR / NTHREADS;
    static constexpr bool VEC_WRITE = true;
    // row form (64-B rows, ds_read_b128 of 16 rows x one chunk): 4 rows per 256-B bank row, chunk ^= (row>>2)&3
    // col form (256-B rows, transposed reads of 4 k-rows x 64 B): 64-B group ^= row&3
    __device__ static __forceinline__ int swz(int row) {
        return COL ? ((row & 3) << 2) : (BKB == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7));
    }
    __device__ static __forceinline__ int offset(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};
template <bool COL> struct TileGeom<f16, COL> : TileGeom<bf16, COL> {};
template <bool COL> struct TileGeom<float, COL> {
    static constexpr int ROWS = COL ? 16 : 128, COLS = COL ? 128 : 16;
    static constexpr int PITCH = COL ? 512 : 68;  // row form: 17 words, ds_read_b32 column reads conflict-free
    static constexpr int BYTES = ROWS * PITCH;
    static constexpr int CPR = COLS / 4, NCH = ROWS * CPR / NTHREADS;
    static constexpr bool VEC_WRITE = COL;
    __device__ static __forceinline__ int offset(int row, int chunk) { return row * PITCH + (chunk << 4); }
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
#pragma unroll
    for (int i = 0; i < G::NCH; ++i) {
        int c = tid + i * NTHREADS;
        char* p = lds + G::offset(c / G::CPR, c % G::CPR);
        if constexpr (G::VEC_WRITE) {
            *reinterpret_cast<decltype(v[i].raw)*>(p) = v[i].raw;
        } else {
            const float* s = reinterpret_cast<const float*>(&v[i].raw);
#pragma unroll
            for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(p)[e] = s[e];
        }
    }
}

// LDS-DMA staging of one bf16 operand tile: 8 pieces of 1 KiB, 2 per wave; lane l of piece b lands at byte
// b*1024 + l*16 and fetches the (row, logical chunk) that the swizzle maps there.  Rows / column chunks past the
// matrix edge re-read a valid address (their products only reach outputs that are never stored); the caller
// guarantees that every k of the tile is in range.
template <bool COL>
__device__ __forceinline__ void tile_glds(char* lds, const bf16* __restrict__ base, long long ld, long long row0,
                                          long long col0, long long row_lim, long long col_lim, int wave, int lane) {
    using G = TileGeom<bf16, COL>;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void g_void;
#pragma unroll
    for (int i = 0; i < G::PIECES / 4; ++i) {
        const int piece = i * 4 + wave;
        const int o = piece * 1024 + lane * 16;
        const int row = o / G::ROWB;
        const int chunk = ((o % G::ROWB) >> 4) ^ G::swz(row);
        long long gr = row0 + row, gc = col0 + chunk * 8;
        if (gr >= row_lim) gr = row_lim - 1;
        if (gc + 8 > col_lim) gc = col0;
        __builtin_amdgcn_global_load_lds((g_void*)(base + gr * ld + gc), (lds_void*)(lds + piece * 1024), 16, 0, 0);
    }
}

// MFMA operand fragment of a 32-row block starting at tile row/col `r0`, k-step `kk` (KSTEP wide), from LDS.
// bf16, 32x32x16: lane l (r = l&31, h = l>>5) holds elements k = 16*kk + 8*h + j, j = 0..7, of row r0 + r.
template <bool COL>
__device__ __forceinline__ bf16x8_t frag_bf16(const char* lds, int r0, int kk, int lane) {
    using G = TileGeom<bf16, COL>;
    if constexpr (!COL) {
        return *reinterpret_cast<const bf16x8_t*>(lds + G::offset(r0 + (lane & 31), kk * 2 + (lane >> 5)));
    } else {
        // tile is [k][m]; ds_read_b64_tr_b16: lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a
        // 4x16 block and receives column (lane & 15), rows 0..3 -> 4 consecutive k of one m.
        int q = (lane & 15) >> 2, p4 = lane & 3;
        int col = r0 + 16 * ((lane >> 4) & 1) + 4 * p4;
        int krow = kk * 16 + 8 * (lane >> 5) + q;
        const char* p = lds + G::offset(krow, col >> 3) + (col & 7) * 2;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 4 * G::ROWB));  // (row+4)&3 == row&3
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
            x *= act_bwd_t<TO>(ep.act, a[e]);
        } else {
            if (ep.bias && e < n_valid) x += to_f32<TO>(reinterpret_cast<const TO*>(ep.bias)[gn + e]);
            pre.set(e, x);
            x = act_fwd_t<TO>(ep.act, x);
            if (ep.mode == 1) x += a[e];
            else if (ep.mode == 3) x *= a[e];
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

// Lean epilogue for interior tiles (full 128x128, 16-byte addressable C / aux / bias) and the common epilogues
// (no activation or ReLU; plain, +residual, or x ReLU'): thread t owns the 8-column (bf16) / 4-column (fp32) chunk
// t % CPR of rows t / CPR + i * RPT, so its bias chunk is loaded once, all LDS reads / aux loads / stores are
// straight-line 16-byte accesses with no per-element branching (the generic path below costs ~2500 instructions per
// wave; this one ~300).
// (round 4: also GELU — forward with or without the pre-activation output, and GELU' as the mask factor of mode 2: the
// generic path made the 1024-row Whisper decoder's fc1 25.4 us against 9.2 without an activation, its dH GEMM 36.6 against 25.6)
template <typename T, int ACT, int MODE, bool PRE = false>
__device__ __forceinline__ void fast_epilogue(const float* __restrict__ cs, T* __restrict__ C, const EpiParams& ep,
                                              long long m0, long long n0, int tid) {
    constexpr int EPV = 16 / sizeof(T), CPR = BN / EPV, RPT = NTHREADS / CPR, NIT = (BM / EPI_PASSES) / RPT;
    const int col = (tid % CPR) * EPV, r0 = tid / CPR;
    float b[EPV];
    if (MODE != 2 && ep.bias) {
        Vec16<T> bv = load16<T>(reinterpret_cast<const T*>(ep.bias) + n0 + col);
#pragma unroll
        for (int e = 0; e < EPV; ++e) b[e] = bv.get(e);
    } else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) b[e] = 0.f;
    }
    T* cp = C + (m0 + r0) * ep.ldc + n0 + col;
    const T* ap = MODE != 0 ? reinterpret_cast<const T*>(ep.aux) + (m0 + r0) * ep.ldaux + n0 + col : nullptr;
    const float alpha = ep.alpha;
    Vec16<T> av[NIT];
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) av[i] = load16<T>(ap + (long long)i * RPT * ep.ldaux);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const float* src = cs + (r0 + i * RPT) * C_PITCH + col;
        float x[EPV];
#pragma unroll
        for (int e = 0; e < EPV; e += 4) {
            float4 t4 = *reinterpret_cast<const float4*>(src + e);
            x[e] = t4.x; x[e + 1] = t4.y; x[e + 2] = t4.z; x[e + 3] = t4.w;
        }
        Vec16<T> o;
        float pre[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            float y = x[e] * alpha;
            if (MODE == 2) {
                if (ACT == PK_ACT_RELU) y = av[i].get(e) > 0.f ? y : 0.f;
                else if (ACT != PK_ACT_NONE) y *= act_bwd_t<T>(ACT, av[i].get(e));
            } else {
                y += b[e];
                pre[e] = y;
                if (ACT == PK_ACT_RELU) y = fmaxf(y, 0.f);
                else if (ACT != PK_ACT_NONE) y = act_fwd_t<T>(ACT, y);
                if (MODE == 1) y += av[i].get(e);
            }
            x[e] = y;
        }
        if constexpr (PRE) {
            store16<T>(reinterpret_cast<T*>(ep.preact) + (m0 + r0 + (long long)i * RPT) * ep.ldpre + n0 + col, vec16_pack<T>(pre));
        }
        if constexpr (sizeof(T) == 2) {
            typedef __attribute__((ext_vector_type(8))) float f32x8;
            f32x8 f = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
            o.raw = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename H16<T>::vec));
        } else {
            o.raw = make_float4(x[0], x[1], x[2], x[3]);
        }
        store16_nt<T>(cp + (long long)i * RPT * ep.ldc, o);
    }
}

#ifndef PK_OCC
#define PK_OCC 2
#endif
template <typename T, bool A_COL, bool B_COL>
__global__ __launch_bounds__(NTHREADS, PK_OCC) void gemm_kernel(
    const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, float* __restrict__ ws,
    float* __restrict__ asum_ws, T* __restrict__ asum_out, long long M, long long N, long long K, long long lda,
    long long ldb, int kchunk, EpiParams ep, int flags) {
    using TR = Traits<T>;
    using GA = TileGeom<T, A_COL>;
    using GB = TileGeom<T, B_COL>;
    constexpr int STAGE = GA::BYTES + GB::BYTES;
    constexpr int NS = TR::NSTAGE;
    constexpr int CST = (BM / EPI_PASSES) * C_PITCH * 4;
    constexpr int SMEM = (NS * STAGE > CST) ? NS * STAGE : CST;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const bool a_vec = flags & 1, b_vec = flags & 2, c_vec = flags & 4, aux_vec = flags & 8;

    // tile walk: XCD-contiguous, 8-row-panel groups, n fastest inside a group
    const int nt_m = (int)((M + BM - 1) / BM), nt_n = (int)((N + BN - 1) / BN);
    // split-K: the (K-slab, tile) list is walked slab-major, so an XCD owns whole slabs — the tiles that re-read the
    // same rows of A and B run on the XCD whose L2 already holds them, and no other XCD fetches those rows
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int kslab = lin / (nt_m * nt_n);
    int t = lin % (nt_m * nt_n);
    // panel-group height: the GROUP_M A panels + the B panels touched by the ~64 workgroups resident on an XCD should stay
    // inside its 4 MiB L2; narrow outputs (nt_n <= 4) afford 16 A panels, which quarters the re-reads of B
    const int GROUP_M = nt_n <= 4 ? 16 : 8;
    int group_size = GROUP_M * nt_n, gid = t / group_size, first_m = gid * GROUP_M;
    int gsz = min(nt_m - first_m, GROUP_M);
    int tile_m = first_m + (t % group_size) % gsz, tile_n = (t % group_size) / gsz;
    const long long m0 = (long long)tile_m * BM, n0 = (long long)tile_n * BN;

    const long long kbeg = (long long)kslab * kchunk;
    const long long kend = min(K, kbeg + (long long)kchunk);
#if !defined(PK_ABLATE) || PK_ABLATE != 5
    const int nk = (int)((kend - kbeg + TR::BK - 1) / TR::BK);
#endif

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fused bias gradient (weight-gradient GEMMs, A = dY in col form): the sum over k of A(m, k) for this tile's 128
    // m-columns, accumulated from the staged LDS tile: thread t owns column chunk t % CPR, rows t / CPR + i*RL.
    // Only the tile_n == 0 workgroups do it.
    const bool do_asum = A_COL && (asum_ws || asum_out) && tile_n == 0;
    float asum[TR::EPV];
#pragma unroll
    for (int e = 0; e < TR::EPV; ++e) asum[e] = 0.f;

    // register staging (fp32 always; bf16 only for K tails / unaligned operands)
    Vec16<T> ra[GA::NCH], rb[GB::NCH];
    auto g2r = [&](int kt) {
        long long k0 = kbeg + (long long)kt * TR::BK;
        if constexpr (A_COL) tile_g2r<T, true>(ra, A, lda, k0, m0, kend, M, a_vec, tid);
        else tile_g2r<T, false>(ra, A, lda, m0, k0, M, kend, a_vec, tid);
        if constexpr (B_COL) tile_g2r<T, true>(rb, B, ldb, k0, n0, kend, N, b_vec, tid);
        else tile_g2r<T, false>(rb, B, ldb, n0, k0, N, kend, b_vec, tid);
    };
    auto r2s = [&](int buf) {
        char* s = smem + buf * STAGE;
        tile_r2s<T, A_COL>(ra, s, tid);
        tile_r2s<T, B_COL>(rb, s + GA::BYTES, tid);
    };
    // one K-tile of MFMAs from LDS stage `sa`/`sb`
    auto compute = [&](const char* sa, const char* sb) {
        if constexpr (A_COL) {
            if (do_asum) {
                constexpr int RL = NTHREADS / GA::CPR;
#pragma unroll
                for (int i = 0; i < GA::ROWS / RL; ++i) {
                    Vec16<T> v;
                    v.raw = *reinterpret_cast<const decltype(v.raw)*>(sa + GA::offset(tid / GA::CPR + i * RL, tid % GA::CPR));
#pragma unroll
                    for (int e = 0; e < TR::EPV; ++e) asum[e] += v.get(e);
                }
            }
        }
        if constexpr (sizeof(T) == 2) {
            // fragment reads of k-step kk+1 are issued before the MFMAs of k-step kk (register double buffer)
            constexpr int NKK = TR::BK / TR::KSTEP;
            typename H16<T>::vec fa[2][2], fb[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[0][i] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<A_COL>(sa, wm + 32 * i, 0, lane));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<B_COL>(sb, wn + 32 * j, 0, lane));
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < NKK) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[nxt][i] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<A_COL>(sa, wm + 32 * i, kk + 1, lane));
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[nxt][j] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<B_COL>(sb, wn + 32 * j, kk + 1, lane));
                }
#if defined(PK_ABLATE) && PK_ABLATE == 2
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(fa[cur][i]), "v"(fb[cur][i]));
#else
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = H16<T>::mfma(fa[cur][i], fb[cur][j], acc[i][j]);
#endif
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < TR::BK / TR::KSTEP; ++kk) {
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
    };

    // ---- main loop ----
    // 16-bit types, 16-byte addressable operands: the K-tiles that are full in k are staged by LDS-DMA into a double
    // buffer, with an explicit vmcnt(0) and a raw barrier per tile (a __syncthreads() would drain the DMA queue too early)
    const bool dma_ok = TR::GLDS && a_vec && b_vec && (M % TR::EPV == 0 || !A_COL) && (N % TR::EPV == 0 || !B_COL);
#if defined(PK_ABLATE) && PK_ABLATE == 5
    const int nk_dma = 0;
    const int nk = 0;
#else
    const int nk_dma = dma_ok ? (int)((kend - kbeg) / TR::BK) : 0;
#endif
    if constexpr (TR::GLDS) {
        static_assert(NS == 2, "the software-pipelined DMA loop below is written for a double buffer");
        constexpr int NKK = TR::BK / TR::KSTEP;
        static_assert(NKK % 2 == 0, "fragment register parity must be the same at every tile start");
        auto dma = [&](int kt) {
            long long k0 = kbeg + (long long)kt * TR::BK;
            char* s = smem + (kt & 1) * STAGE;
            if constexpr (A_COL) tile_glds<true>(s, (const bf16*)A, lda, k0, m0, kend, M, wave, lane);
            else tile_glds<false>(s, (const bf16*)A, lda, m0, k0, M, kend, wave, lane);
            if constexpr (B_COL) tile_glds<true>(s + GA::BYTES, (const bf16*)B, ldb, k0, n0, kend, N, wave, lane);
            else tile_glds<false>(s + GA::BYTES, (const bf16*)B, ldb, n0, k0, N, kend, wave, lane);
        };
        if (nk_dma > 0) {
            // Software pipeline across K-tiles: the fragments of (tile t+1, k-step 0) are read from LDS — and the DMA of
            // tile t+2 is issued — BEFORE the last 4 MFMAs of tile t, right behind the one barrier per tile, so neither the
            // LDS read latency nor the DMA issue sits in front of a tile's first MFMA.
            typename H16<T>::vec fa[2][2], fb[2][2];
            dma(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[0][i] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<A_COL>(smem, wm + 32 * i, 0, lane));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<B_COL>(smem + GA::BYTES, wn + 32 * j, 0, lane));
            if (nk_dma > 1) dma(1);
            for (int kt = 0; kt < nk_dma; ++kt) {
                const char* sa = smem + (kt & 1) * STAGE;
                const char* sb = sa + GA::BYTES;
                if constexpr (A_COL) {
                    if (do_asum) {
                        constexpr int RL = NTHREADS / GA::CPR;
#pragma unroll
                        for (int i = 0; i < GA::ROWS / RL; ++i) {
                            Vec16<T> v;
                            v.raw = *reinterpret_cast<const decltype(v.raw)*>(
                                sa + GA::offset(tid / GA::CPR + i * RL, tid % GA::CPR));
#pragma unroll
                            for (int e = 0; e < TR::EPV; ++e) asum[e] += v.get(e);
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const int cur = kk & 1, nxt = cur ^ 1;
                    if (kk + 1 < NKK) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) fa[nxt][i] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<A_COL>(sa, wm + 32 * i, kk + 1, lane));
#pragma unroll
                        for (int j = 0; j < 2; ++j) fb[nxt][j] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<B_COL>(sb, wn + 32 * j, kk + 1, lane));
                    } else if (kt + 1 < nk_dma) {
                        // every LDS read of tile kt has returned, my pieces of tile kt+1 have landed
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        const char* na = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
                        for (int i = 0; i < 2; ++i) fa[nxt][i] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<A_COL>(na, wm + 32 * i, 0, lane));
#pragma unroll
                        for (int j = 0; j < 2; ++j) fb[nxt][j] = __builtin_bit_cast(typename H16<T>::vec, frag_bf16<B_COL>(na + GA::BYTES, wn + 32 * j, 0, lane));
#if !defined(PK_ABLATE) || PK_ABLATE != 1
                        if (kt + 2 < nk_dma) dma(kt + 2);  // into the stage tile kt just vacated
#endif
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = H16<T>::mfma(fa[cur][i], fb[cur][j], acc[i][j]);
                }
            }
            __syncthreads();
        }
    }
    // register-staged path: fp32, unaligned operands, and the K tail (rows/cols past kend are zero-filled)
    if (nk_dma < nk) {
        g2r(nk_dma);
        r2s(0);
        __syncthreads();
        for (int kt = nk_dma; kt < nk; ++kt) {
            const int buf = (kt - nk_dma) & 1;
            if (kt + 1 < nk) g2r(kt + 1);  // next tile's loads fly under this tile's MFMAs
            const char* sa = smem + buf * STAGE;
            compute(sa, sa + GA::BYTES);
            if (kt + 1 < nk) r2s(buf ^ 1);
            __syncthreads();
        }
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
                if (asum_ws) asum_ws[(long long)kslab * M + m0 + tid] = t;
                else asum_out[m0 + tid] = from_f32<T>(t);
            }
            __syncthreads();
        }
    }

#if defined(PK_ABLATE) && PK_ABLATE == 4
    if (acc[0][0][0] != 12345.f) return;
#endif
    // ---- epilogue: accumulators -> LDS (f32) -> row-contiguous chunks -> global, in EPI_PASSES row slabs (1: the
    // whole 128x128 tile through a 66 KiB staging buffer — measured faster than 2 x 33 KiB at 2 workgroups/CU) ----
    float* cs = reinterpret_cast<float*>(smem);
    constexpr int EPV = TR::EPV;
    constexpr int HM = BM / EPI_PASSES;
    const bool interior = m0 + BM <= M && n0 + BN <= N;
    const bool bias_ok = !ep.bias || (((uintptr_t)ep.bias % 16) == 0);
    const bool gelu16 = sizeof(T) == 2 && ep.act == PK_ACT_GELU && ep.mode != 1;  // (16-bit: the shared fast evaluation)
    const bool simple = (!ep.preact || (gelu16 && ep.mode == 0)) && c_vec && bias_ok && (ep.mode == 0 || aux_vec) && ep.mode < 3 &&
                        (ep.act == PK_ACT_NONE || ep.act == PK_ACT_RELU || gelu16);
#pragma unroll 1
    for (int half = 0; half < EPI_PASSES; ++half) {
        if (EPI_PASSES == 1 || (wave >> 1) == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = (EPI_PASSES == 1 ? wm : 0) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        int col = wn + 32 * j + (lane & 31);
                        cs[row * C_PITCH + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        const long long mh = m0 + half * HM;
        if (ws) {  // split-K partial: raw f32 slab [splitk][M][N]
            float* slab = ws + (long long)kslab * M * N;
            const bool ws_vec = (N % 4) == 0;
#pragma unroll 4
            for (int c = tid; c < HM * (BN / 4); c += NTHREADS) {
                int row = c / (BN / 4), col = (c % (BN / 4)) * 4;
                long long gm = mh + row, gn = n0 + col;
                if (gm >= M || gn >= N) continue;
                float4 v = *reinterpret_cast<const float4*>(cs + row * C_PITCH + col);
                float* p = slab + gm * N + gn;
                if (ws_vec && gn + 4 <= N) *reinterpret_cast<float4*>(p) = v;
                else
                    for (int e = 0; e < 4 && gn + e < N; ++e) p[e] = (&v.x)[e];
            }
        } else if (interior && simple) {  // block-uniform: lean path
            if (ep.mode == 0) {
                if (ep.act == PK_ACT_RELU) fast_epilogue<T, PK_ACT_RELU, 0>(cs, C, ep, mh, n0, tid);
                else if (ep.act == PK_ACT_GELU) {
                    if (ep.preact) fast_epilogue<T, PK_ACT_GELU, 0, true>(cs, C, ep, mh, n0, tid);
                    else fast_epilogue<T, PK_ACT_GELU, 0>(cs, C, ep, mh, n0, tid);
                } else fast_epilogue<T, PK_ACT_NONE, 0>(cs, C, ep, mh, n0, tid);
            } else if (ep.mode == 1) {
                if (ep.act == PK_ACT_RELU) fast_epilogue<T, PK_ACT_RELU, 1>(cs, C, ep, mh, n0, tid);
                else fast_epilogue<T, PK_ACT_NONE, 1>(cs, C, ep, mh, n0, tid);
            } else {
                if (ep.act == PK_ACT_RELU) fast_epilogue<T, PK_ACT_RELU, 2>(cs, C, ep, mh, n0, tid);
                else if (ep.act == PK_ACT_GELU) fast_epilogue<T, PK_ACT_GELU, 2>(cs, C, ep, mh, n0, tid);
                else fast_epilogue<T, PK_ACT_NONE, 2>(cs, C, ep, mh, n0, tid);
            }
        } else {
#pragma unroll 2
            for (int c = tid; c < HM * (BN / EPV); c += NTHREADS) {
                int row = c / (BN / EPV), col = (c % (BN / EPV)) * EPV;
                long long gm = mh + row, gn = n0 + col;
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
        if (half + 1 < EPI_PASSES) __syncthreads();
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
    // lean path (weight gradients: no bias / activation, optional accumulate): straight-line 16-byte accesses, the
    // splitk slab reads of a chunk are independent loads in flight together
    const bool lean = c_vec && (N % EPV) == 0 && !ep.bias && !ep.preact && ep.act == PK_ACT_NONE &&
                      (ep.mode == 0 || (ep.mode == 1 && aux_vec));  // (mode 3 takes the generic path)
    if (lean) {
        const long long slab = M * N;
        for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
             c += (long long)gridDim.x * blockDim.x) {
            const long long gm = c / nchunks_row, gn = (c % nchunks_row) * EPV;
            const float* p = ws + gm * N + gn;
            float v[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) v[e] = 0.f;
#pragma unroll 4
            for (int z = 0; z < splitk; ++z) {
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    float4 t4 = *reinterpret_cast<const float4*>(p + (long long)z * slab + e);
                    v[e] += t4.x; v[e + 1] += t4.y; v[e + 2] += t4.z; v[e + 3] += t4.w;
                }
            }
            if (ep.mode == 1) {
                Vec16<T> a = load16<T>(reinterpret_cast<const T*>(ep.aux) + gm * ep.ldaux + gn);
#pragma unroll
                for (int e = 0; e < EPV; ++e) v[e] = v[e] * ep.alpha + a.get(e);
            } else {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v[e] = v[e] * ep.alpha;
            }
            store16_nt<T>(C + gm * ep.ldc + gn, vec16_pack<T>(v));
        }
        return;
    }
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

// PK_GEMM_8P=0 / pk_gemm_use_8p(0): keep the 256-tile GEMMs on gemm256.hip (A/B of the two K loops inside one process)
// B-stationary kernel for K = 512 / 256 (gemmbs.hip)
extern "C" int pk_gemmbs_eligible(const void* A, const void* B, const void* C, long long M, long long N, long long K,
                                  long long lda, long long ldb, int a_col, int b_col, const EpiParams* ep);
extern "C" int pk_gemmbs_launch(const void* A, const void* B, void* C, long long M, long long N, long long K,
                                long long lda, long long ldb, int b_col, EpiParams ep, int dtype, void* stream,
                                unsigned char* bits, long long ldbits);
extern "C" int pk_gemmbs_use(int on);

int g_use_8p = [] { const char* e = getenv("PK_GEMM_8P"); return (!e || atoi(e) != 0) ? 1 : 0; }();

// ---- optional launch timing (bench.py's roofline leg): HIP events around exactly the main GEMM kernel, on its stream ----
struct GemmSample {
    hipEvent_t start, stop;
    int kernel, a_col, b_col, splitk, dtype;
    double flops;
    long long M, N, K;
};
struct GemmTiming {
    bool on = false;
    int stride = 1, count = 0, used = 0;
    std::vector<GemmSample> pool;
} g_timing;

// returns the sample slot to fill (events already recorded `start`) or nullptr
inline GemmSample* timing_begin(int kernel, int a_col, int b_col, int splitk, int dtype, long long M, long long N,
                                long long K, hipStream_t stream) {
    if (!g_timing.on) return nullptr;
    if (++g_timing.count % g_timing.stride != 0 || g_timing.used >= (int)g_timing.pool.size()) return nullptr;
    GemmSample* sm = &g_timing.pool[g_timing.used++];
    sm->kernel = kernel; sm->a_col = a_col; sm->b_col = b_col; sm->splitk = splitk; sm->dtype = dtype;
    sm->flops = 2.0 * (double)M * (double)N * (double)K;
    sm->M = M; sm->N = N; sm->K = K;
    (void)hipEventRecord(sm->start, stream);
    return sm;
}
inline void timing_end(GemmSample* sm, hipStream_t stream) {
    if (sm) (void)hipEventRecord(sm->stop, stream);
}

template <typename T>
int launch_gemm(const void* A, const void* B, void* C, long long M, long long N, long long K, long long lda,
                long long ldb, int a_col, int b_col, EpiParams ep, int splitk, void* workspace,
                size_t ws_bytes, void* asum_out, hipStream_t stream, int pad_flags = 0) {
    using TR = Traits<T>;
    constexpr int EPV = TR::EPV;
    auto aligned = [&](const void* p, long long ld) { return ((uintptr_t)p % 16) == 0 && (ld % EPV) == 0; };
    // pk_gemm_ex: a vocabulary-sized dimension that is no multiple of 8 inside buffers whose rows are padded to one.  The
    // promises only matter to the 256-tile kernel (16-byte chunks everywhere); the 128-tile kernel below ignores them.
    const bool pad_n = (pad_flags & PK_GEMM_PAD_N) && sizeof(T) == 2 && N % 8 && ep.mode == 0 && !ep.bias && !ep.preact &&
                       ep.act == PK_ACT_NONE && splitk <= 1 && ep.ldc >= ((N + 7) & ~7LL);
    const bool pad_k = (pad_flags & PK_GEMM_PAD_K) && sizeof(T) == 2 && K % 8 && !a_col && b_col && lda >= ((K + 7) & ~7LL);
    int flags = 0;
    if (aligned(A, lda)) flags |= 1;
    if (aligned(B, ldb)) flags |= 2;
    if (aligned(C, ep.ldc) && (!ep.preact || aligned(ep.preact, ep.ldpre))) flags |= 4;
    if (ep.aux && aligned(ep.aux, ep.ldaux)) flags |= 8;
    constexpr int dtype16 = std::is_same<T, f16>::value ? PK_F16 : PK_BF16;  // (meaningful for the 16-bit types)
    int nt = (int)(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
    int kchunk = (int)K;
    float* ws = nullptr;
    if constexpr (sizeof(T) == 2) {
        // a handful of rows (one decoding step): latency-shaped kernel without LDS staging (gemm_skinny.hip)
        static const bool no_skinny = getenv("PK_GEMM_NO_SKINNY") != nullptr;
        // (every 64-row block re-reads its weight columns: only while that is cheaper than a tiled kernel's latency)
        // (up to 1024 rows for N <= 512 — the out / q / fc2 projections of a 1024-row decoder batch, C4: 256 workgroups
        // instead of the 32 of the 128-tile kernel: 22.5 -> 15.4 us at K = 2048, 8.3 -> 7.8 at K = 512; wider outputs re-read
        // too much of A: N = 1536 / 2048 at K = 512 8.8 -> 12.2 / 9.0 -> 15.4 us, they stay on the tiles.  C4 step -0.25 ms.)
        // (round 4: any number of rows for outputs of <= 64 columns — an adapter's down-projection, 16 000 x 64 with K = 1024: the
        // weight is 128 KB, re-reading it per 64-row block costs nothing, while the 128-tile kernel ran it at 1.2 TB/s: 28 -> 9 us)
        // (PK_SKINNY_M2 / PK_SKINNY_N2: the second row / column bound, 1024 / 512 by default — the experiment knob of round 5)
        static const long long sk_m2 = [] { const char* e = getenv("PK_SKINNY_M2"); return e ? atoll(e) : 1024LL; }();
        static const long long sk_n2 = [] { const char* e = getenv("PK_SKINNY_N2"); return e ? atoll(e) : 512LL; }();
        if (!a_col && !b_col && (M <= 64 || (M <= 256 && N <= 8192) || (M <= sk_m2 && N <= sk_n2) || N <= 64) && splitk <= 1 && !asum_out &&
            !no_skinny) {
            GemmSample* sm = timing_begin(64, a_col, b_col, 1, dtype16, M, N, K, stream);  // (sample tag 64: the few-rows kernel)
            int rc = pk_gemm_skinny_launch(A, B, C, M, N, K, lda, ldb, ep, dtype16, stream);
            if (rc == 1) timing_end(sm, stream);
            else if (sm) --g_timing.used;  // not eligible (or an error): the sample slot goes back
            if (rc != 0) return rc == 1 ? 0 : rc;  // 1 = launched, 0 = not eligible, anything else = error
        }
        // short contraction (K = 512 / 256), many rows, lean epilogue: the B-stationary walk (gemmbs.hip) — no epilogue phase
        // (sample tag 0x200 | K-tiles | activation << 4 | 0x40 act'-mask epilogue of mode 2 | 0x80 mask as bits | 0x100 preact)
        if (splitk <= 1 && !asum_out && pk_gemmbs_eligible(A, B, C, M, N, K, lda, ldb, a_col, b_col, &ep)) {
            const int tagbs = 0x200 | (int)(K / 64) | ((ep.act & 3) << 4) | (ep.mode == 2 ? 0x40 : 0) | (ep.preact ? 0x100 : 0);
            GemmSample* sm = timing_begin(tagbs, a_col, b_col, 1, dtype16, M, N, K, stream);
            const int rc = pk_gemmbs_launch(A, B, C, M, N, K, lda, ldb, b_col, ep, dtype16, stream, nullptr, 0);
            timing_end(sm, stream);
            return rc == 1 ? 0 : rc;
        }
        // 256x256-tile kernel (gemm256.hip): LDS-DMA only, so it needs 16-byte addressable operands, K in whole 64-tiles
        // and one of the lean epilogues; it pays when its (4x fewer) tiles still fill the chip.
        static const int tile_pref = [] { const char* e = getenv("PK_GEMM_TILE"); return e ? atoi(e) : 0; }();
        // epilogues of the 256-tile kernels: everything 16-byte addressable; gemm256.hip only knows none / ReLU without
        // a pre-activation output, gemm8p.hip every activation, `preact` and the gate product (mode 3)
        const bool epi_ok = (flags & 4) && (!ep.bias || ((uintptr_t)ep.bias % 16) == 0) && (ep.mode == 0 || (flags & 8));
        const bool lean_epi = !ep.preact && (ep.act == PK_ACT_NONE || ep.act == PK_ACT_RELU) && ep.mode < 3;
        const bool addr_ok = (flags & 1) && (flags & 2) && (!a_col || M % 8 == 0 || lda >= ((M + 7) & ~7LL)) &&
                             (!b_col || N % 8 == 0) &&
                             (N % 8 == 0 || pad_n) && K > 0;
        // gemm8p.hip zero-fills a partial last K-tile (K % 8 == 0: the vocabulary dX GEMMs, K = V); gemm256.hip needs
        // whole 64-deep tiles
        const long long K8 = pad_k ? ((K + 7) & ~7LL) : K;  // (what the phase-interleaved kernel contracts over)
        const bool e8 = g_use_8p && pk_gemm8p_eligible(M, N, K8, lda, ldb, a_col, b_col, asum_out != nullptr);
        const bool k_ok = e8 || K % 64 == 0;
        const bool simple = epi_ok && (lean_epi || (e8 && splitk <= 1)) &&  // (the split-K reduce kernel has its own epilogue)
                            (e8 || (N % 8 == 0 && !pad_k));  // (only gemm8p.hip knows the padded forms)
        const long long t256 = ((M + 255) / 256) * ((N + 255) / 256);
        // (round 5: a general epilogue — GELU with its pre-activation output — behind a contraction of a few K-tiles is an
        // elementwise pass, not a GEMM: Whisper's first conv, 48 032 x 512 x 240, took 99 us on the 256-tile kernel's general
        // epilogue call; the 128-tile kernel's straight-line GELU path with two workgroups per CU takes it.  PK_GEMM_ANY_SHORTK=0: off)
        static const bool any_shortk = [] { const char* e = getenv("PK_GEMM_ANY_SHORTK"); return !e || atoi(e) != 0; }();
        const bool general_epi = ep.preact || ep.mode == 3 || (ep.act != PK_ACT_NONE && ep.act != PK_ACT_RELU);
        const bool short_any = any_shortk && general_epi && K8 <= 256 && ep.act == PK_ACT_GELU && ep.mode != 1 && ep.mode != 3;
        if (simple && addr_ok && k_ok && tile_pref != 128 && M >= 256 && N >= 256 && !short_any) {
            int sk = 1;
            long long per = K8;
            // split-K (weight-gradient) GEMMs: the 256 kernel re-derives its own split factor (~1 workgroup per CU).
            // Measured in the training step after the slab-major walk: C2 19.95 vs 20.16 ms, transformer_big 74.5 vs
            // 79.0 ms in favour of the 256 kernel.  PK_GEMM_SK256=2 keeps split-K on the 128 kernel, =1 caps the factor.
            static const int sk_mode = [] { const char* e = getenv("PK_GEMM_SK256"); return e ? atoi(e) : 0; }();
            static const int sk_min_tiles = [] { const char* e = getenv("PK_GEMM_SK256_MIN_TILES"); return e ? atoi(e) : 8; }();
            // fewer than 8 output tiles (the d x d weight gradients of the base model: 4) would be cut into 64 K = 512
            // slices, all prologue / epilogue, and 64 fp32 slabs to reduce: on 128-tiles with 32 slices the step gains
            // 1 % (19.99 -> 19.76 ms); from 12 tiles up (qkv, fc1/fc2 gradients) the 256 kernel wins (threshold 16: 14.27
            // vs 14.00 ms of GEMM time per step)
            // an output the 128 x 256-tile form fills the chip with (80..159 256-tiles: NLLB-1.3B's N = d GEMMs at 8192 rows)
            // is not split even where the caller allows it: two K slices of 256-tiles + the reduction launch took 103 + 17 us
            // for the fc2 dX (8192 x 1024 x 8192) where the unsplit 128 x 256 tiles take 111 (round 5; PK_GEMM_HM_NOSPLIT=0: A/B)
            static const bool hm_nosplit = [] { const char* e = getenv("PK_GEMM_HM_NOSPLIT"); return !e || atoi(e) != 0; }();
            static const bool halfm_env = [] { const char* e = getenv("PK_GEMM_HALFM"); return !e || atoi(e) != 0; }();
            // (round 5: also a few thousand rows x d with a d-long contraction — the frozen 2048-row decoder projections of the
            // IWSLT recipe and their dX: 64 tiles of 128 x 256 instead of 128 tiles of 128 x 128, with two K slices for the dX —
            // 19 -> 16 us per launch, 142 launches per step: 64.26 -> 63.92 ms same box.  Longer contractions stay split (fc2,
            // 8192 deep: 40 us as K-slabs, ~90 unsplit on 64 workgroups).  PK_GEMM_HM_SMALL=0: off)
            static const bool hm_small = [] { const char* e = getenv("PK_GEMM_HM_SMALL"); return !e || atoi(e) != 0; }();
            const long long t_half_pre = ((M + 127) / 128) * ((N + 255) / 256);
            const bool small_hm = hm_small && M <= 4096 && t_half_pre >= 48 && t_half_pre < 160 && K8 >= 1024 && K8 <= 1536 && N % 256 == 0;
            const bool hm_takes_it = hm_nosplit && halfm_env && e8 && t256 < 160 && !a_col && !asum_out && lean_epi &&
                                     (t_half_pre >= 160 || small_hm) && tile_pref != 256 && g_use_8p != 2;
            if (splitk > 1 && hm_takes_it) splitk = 1;
            if (splitk > 1 && (sk_mode == 2 || t256 < sk_min_tiles)) sk = 0;
            else if (splitk > 1) {  // the caller allows split-K: re-derive the factor for 256-tiles (~1 workgroup per CU)
                sk = (int)std::max(1LL, std::min((long long)(256 / std::max(1LL, t256)), K8 / 512));
                sk = std::min(sk, 2 * splitk);  // the caller sized the workspace for twice its own factor
                if (sk_mode == 1) sk = std::min(sk, splitk);
                per = ((K8 + sk - 1) / sk + 63) / 64 * 64;
                sk = (int)((K8 + per - 1) / per);
                if ((size_t)sk * M * (N + (asum_out ? 1 : 0)) * sizeof(float) > ws_bytes) sk = 0;  // does not fit
            }
            const bool fills = t256 * std::max(sk, 1) >= 160;
            // an output of 80..159 256-tiles (8192 x 1024: NLLB-1.3B's out-proj / cross-q / fc2 and their dX at C5) fills half
            // the chip with 256-tiles and runs as one or two low-rate rounds of 128-tiles (560-950 TFLOP/s): gemm8p's half-M
            // form (128 x 256 tiles) gives every CU one.  Lean epilogues, row-form A, no split-K.  PK_GEMM_HALFM=0: off (A/B).
            static const bool halfm_on = [] { const char* e = getenv("PK_GEMM_HALFM"); return !e || atoi(e) != 0; }();
            const long long t_half = ((M + 127) / 128) * ((N + 255) / 256);
            // (round 5, PK_GEMM_HM_ROUNDS: an output whose 256-tiles leave the last round of the chip half empty while its
            // 128 x 256 tiles make whole rounds — NLLB-1.3B's q|k|v projection at 8192 rows: 384 tiles = 1.5 rounds against 768 =
            // 3 — by the per-tile costs of the two schedules, K / 64 x 1.45 + 6.5 us against K / 64 x 0.72 + 5.5)
            // (C5's q|k|v forward 64.5 -> 59.2 us, the step 72.93 -> 72.75 ms same box; =0: off)
            static const bool hm_rounds = [] { const char* e = getenv("PK_GEMM_HM_ROUNDS"); return !e || atoi(e) != 0; }();
            bool by_rounds = false;
            if (hm_rounds && fills && sk == 1 && t256 < 1024) {
                const double kt = (double)((K8 + 63) / 64);
                const double c256 = (double)((t256 + 255) / 256) * (kt * 1.45 + 6.5), chalf = (double)((t_half + 255) / 256) * (kt * 0.72 + 5.5);
                by_rounds = chalf < 0.92 * c256;
            }
            const bool half_m = halfm_on && e8 && (!fills || by_rounds) && sk == 1 && !a_col && !asum_out && lean_epi && (t_half >= 160 || small_hm) && tile_pref != 256 && g_use_8p != 2;
            if (sk > 0 && (tile_pref == 256 || fills || half_m || (g_use_8p == 2 && e8))) {
                float* w2 = sk > 1 ? (float*)workspace : nullptr;
                float* asw = (sk > 1 && asum_out) ? w2 + (size_t)sk * M * N : nullptr;
                // the phase-interleaved kernel (gemm8p.hip) takes what it can; gemm256.hip the rest (fused bias gradient,
                // operands beyond 4 GiB).  PK_GEMM_8P=0 switches it off (A/B inside one process: tools/gemm_bench.py)
                // (sample tag of the gemm8p instantiation: 8 | 0x10 general epilogue | 0x20 partial last K-tile)
                const bool any_epi = ep.preact || ep.mode == 3 || (ep.act != PK_ACT_NONE && ep.act != PK_ACT_RELU);
                EpiParams ep8 = ep;
                if (e8 && pad_n) ep8.nstore = (N + 7) & ~7LL;
                if (e8 && pad_k) ep8.kb_rows = K;
                ep8.half_m = half_m ? 1 : 0;
                float* w2pre = sk > 1 ? (float*)workspace : nullptr;
                // (0x800: the persistent 128 x 256-tile kernel, gemmpw.hip — whatever tile the rules above chose)
                const bool pw = e8 && pk_gemm8p_is_pw(M, N, K8, lda, ldb, a_col, b_col, std::max(sk, 1), w2pre != nullptr,
                                                      asum_out != nullptr, &ep8);
                // (0x4000: the persistent walk of 256 x 256 tiles, gemm8p_pt_kernel)
                const bool pt = e8 && !pw && pk_gemm8p_is_pt(M, N, K8, lda, ldb, a_col, b_col, std::max(sk, 1), w2pre != nullptr,
                                                             asum_out != nullptr, &ep8);
                const int tag8 = pw ? (8 | 0x800) : pt ? (8 | 0x4000) : (8 | (any_epi ? 0x10 : 0) | (K8 % 64 ? 0x20 : 0) | (half_m ? 0x400 : 0));
                GemmSample* sm = timing_begin(e8 ? tag8 : 256, a_col, b_col, std::max(sk, 1), dtype16, M, N, K, stream);
                int rc = (e8 ? pk_gemm8p_launch : pk_gemm256_launch)(A, B, C, w2, asw, asum_out, M, N, e8 ? K8 : K, lda, ldb,
                                                                    a_col, b_col, (int)per, std::max(sk, 1), ep8, dtype16, stream);
                timing_end(sm, stream);
                if (rc != 1) return rc;
                if (w2) {
                    long long chunks = M * ((N + EPV - 1) / EPV);
                    int blocks = (int)min((long long)2048, (chunks + 255) / 256);
                    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, stream, w2, (T*)C, M, N,
                                       sk, ep, flags, (const float*)asw, (T*)asum_out);
                    PK_LAUNCH_CHECK();
                }
                return 0;
            }
        }
    }
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
    dim3 grid(nt * splitk), block(NTHREADS);
    const T* a = (const T*)A;
    const T* b = (const T*)B;
    T* c = (T*)C;
#define PK_LAUNCH(AC, BC) \
    hipLaunchKernelGGL((gemm_kernel<T, AC, BC>), grid, block, 0, stream, a, b, c, ws, asum_ws, (T*)asum_out, M, N, K, \
                       lda, ldb, kchunk, ep, flags)
    GemmSample* sm = timing_begin(128, a_col, b_col, splitk, sizeof(T) == 4 ? PK_F32 : dtype16, M, N, K, stream);
    if (!a_col && !b_col) PK_LAUNCH(false, false);
    else if (!a_col && b_col) PK_LAUNCH(false, true);
    else if (a_col && !b_col) PK_LAUNCH(true, false);
    else PK_LAUNCH(true, true);
#undef PK_LAUNCH
    timing_end(sm, stream);
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

extern "C" int pk_gemm_ex(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact,
                          long long M, long long N, long long K, long long lda, long long ldb, long long ldc,
                          long long ldaux, long long ldpre, int a_col, int b_col, int act, int mode, float alpha,
                          int dtype, int splitk, void* workspace, size_t ws_bytes, void* asum_out, int pad_flags,
                          void* stream);

extern "C" int pk_gemm(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact,
                       long long M, long long N, long long K, long long lda, long long ldb, long long ldc,
                       long long ldaux, long long ldpre, int a_col, int b_col, int act, int mode, float alpha,
                       int dtype, int splitk, void* workspace, size_t ws_bytes, void* asum_out, void* stream) {
    return pk_gemm_ex(A, B, C, bias, aux, preact, M, N, K, lda, ldb, ldc, ldaux, ldpre, a_col, b_col, act, mode, alpha, dtype,
                      splitk, workspace, ws_bytes, asum_out, 0, stream);
}

extern "C" int pk_gemm_ex(const void* A, const void* B, void* C, const void* bias, const void* aux, void* preact,
                          long long M, long long N, long long K, long long lda, long long ldb, long long ldc,
                          long long ldaux, long long ldpre, int a_col, int b_col, int act, int mode, float alpha,
                          int dtype, int splitk, void* workspace, size_t ws_bytes, void* asum_out, int pad_flags,
                          void* stream) {
    if (M == 0 || N == 0) return 0;  // an empty output: nothing to read, nothing to write (operands may be NULL)
    PK_CHECK_ARG((pad_flags & ~(PK_GEMM_PAD_N | PK_GEMM_PAD_K)) == 0, "pk_gemm_ex: unknown flags %d", pad_flags);
    PK_CHECK_ARG(A && B && C, "pk_gemm: null operand");
    PK_CHECK_ARG(!asum_out || a_col, "pk_gemm: asum_out (fused bias gradient) needs A in col form");
    PK_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "pk_gemm: negative size");
    PK_CHECK_ARG(dtype == PK_F32 || dtype == PK_BF16 || dtype == PK_F16, "pk_gemm: dtype %d not supported", dtype);
    PK_CHECK_ARG(mode >= 0 && mode <= 3, "pk_gemm: bad epilogue mode %d", mode);
    PK_CHECK_ARG(mode == 0 || aux, "pk_gemm: epilogue mode %d needs aux", mode);
    PK_CHECK_ARG(((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), "pk_gemm: too many tiles");
    if (M == 0 || N == 0) return 0;
    EpiParams ep;
    ep.bias = bias; ep.aux = aux; ep.preact = preact;
    ep.ldaux = ldaux; ep.ldc = ldc; ep.ldpre = ldpre;
    ep.act = act; ep.mode = mode; ep.alpha = alpha;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PK_BF16)
        return launch_gemm<bf16>(A, B, C, M, N, K, lda, ldb, a_col, b_col, ep, splitk, workspace, ws_bytes, asum_out, s, pad_flags);
    if (dtype == PK_F16)
        return launch_gemm<f16>(A, B, C, M, N, K, lda, ldb, a_col, b_col, ep, splitk, workspace, ws_bytes, asum_out, s, pad_flags);
    return launch_gemm<float>(A, B, C, M, N, K, lda, ldb, a_col, b_col, ep, splitk, workspace, ws_bytes, asum_out, s);
}

// ---- grouped weight gradients (see include/pasero_hip.h) ----
extern "C" int pk_gemm_wgrad_group_eligible(const PkWgradProblem* q, int dtype) {
    if (!q || (dtype != PK_BF16 && dtype != PK_F16) || !g_use_8p) return 0;
    return pk_gemm8p_group_eligible(q);
}

extern "C" size_t pk_gemm_wgrad_group_workspace(const PkWgradProblem* p, int n) {
    size_t b = 0;
    if (!p || pk_gemm8p_group_plan(p, n, &b, nullptr, nullptr) != 0) return 0;
    return b;
}

extern "C" int pk_gemm_wgrad_group_map(const PkWgradProblem* p, int n, int* out, int cap) {
    if (!p || !out) return -1;
    return pk_gemm8p_group_map(p, n, out, cap);
}

extern "C" int pk_gemm_wgrad_group(const PkWgradProblem* p, int n, int dtype, void* workspace, size_t ws_bytes,
                                   void* stream) {
    if (n == 0) return 0;
    PK_CHECK_ARG(p && n > 0 && n <= PK_WGRAD_MAX, "pk_gemm_wgrad_group: 1..%d problems per launch, got %d", PK_WGRAD_MAX, n);
    PK_CHECK_ARG(dtype == PK_BF16 || dtype == PK_F16, "pk_gemm_wgrad_group: 16-bit operands only (dtype %d)", dtype);
    for (int i = 0; i < n; ++i)
        PK_CHECK_ARG(pk_gemm_wgrad_group_eligible(&p[i], dtype),
                     "pk_gemm_wgrad_group: problem %d (M=%lld N=%lld K=%lld) is not eligible; ask "
                     "pk_gemm_wgrad_group_eligible first and send it through pk_gemm", i, p[i].M, p[i].N, p[i].K);
    size_t need = 0;
    int wgs = 0;
    PK_CHECK_ARG(pk_gemm8p_group_plan(p, n, &need, &wgs, nullptr) == 0, "pk_gemm_wgrad_group: plan failed");
    PK_CHECK_ARG(need == 0 || (workspace && ws_bytes >= need && ((uintptr_t)workspace % 16) == 0),
                 "pk_gemm_wgrad_group: workspace too small or misaligned (%zu < %zu)", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    double flops = 0;
    for (int i = 0; i < n; ++i) flops += 2.0 * (double)p[i].M * (double)p[i].N * (double)p[i].K;
    // (sample tag 8 | 0x40: the grouped instantiation of the phase-interleaved kernel; flops = the sum over the group)
    GemmSample* sm = timing_begin(8 | 0x40, 1, 1, n, dtype, 1, 1, 1, s);
    if (sm) sm->flops = flops;
    int rc = pk_gemm8p_group_launch(p, n, dtype, (float*)workspace, stream);
    timing_end(sm, s);
    if (rc != 1) return rc;
    rc = pk_gemm8p_group_reduce(p, n, dtype, (float*)workspace, stream);
    return rc == 1 ? 0 : rc;
}

// ---- Linear + residual + dropout + LayerNorm in one kernel (gemmln.hip; see include/pasero_hip.h) ----
extern "C" int pk_gemm_ln_fwd(const void* A, const void* W, const void* bias, const void* residual, const void* gamma,
                              const void* beta, void* z_out, void* y_out, float* mean, float* rstd, long long M,
                              long long N, long long K, long long lda, long long ldb, long long ldr, float eps,
                              float drop_p, unsigned long long seed, unsigned long long offset, int dtype,
                              void* stream) {
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // (sample tag 8 | 0x80: the 128 x 512-tile instantiation with the LayerNorm epilogue; the sample's `a_col` field carries
    // the epilogue specialisation gemmln.hip picks: 0 generic, 1 residual, 2 residual + dropout)
    GemmSample* sm = timing_begin(8 | 0x80, pk_gemmln_spec(residual != nullptr, drop_p, K), 0, 1, dtype, M, N, K, s);
    const int rc = pk_gemmln_launch(A, W, bias, residual, gamma, beta, z_out, y_out, mean, rstd, M, N, K, lda, ldb, ldr,
                                    eps, drop_p, seed, offset, dtype, stream);
    timing_end(sm, s);
    return rc;
}

extern "C" int pk_gemm_use_bs(int on) { return pk_gemmbs_use(on); }

// ---- ReLU feed-forward with the mask as one bit per element (see include/pasero_hip.h) ----
namespace {
EpiParams relu_bits_epi(const void* bias, void* C, long long ldc, int mode, float alpha) {
    EpiParams ep;
    ep.bias = mode == 0 ? bias : nullptr;
    ep.aux = mode == 2 ? C : nullptr;  // (stands in for the mask operand in the eligibility check: the bits replace it)
    ep.preact = nullptr;
    ep.ldaux = mode == 2 ? ldc : 0; ep.ldc = ldc; ep.ldpre = 0;
    ep.act = PK_ACT_RELU; ep.mode = mode; ep.alpha = alpha;
    return ep;
}
}  // namespace

namespace {
// round 5: the same mask bits through the phase-interleaved kernel's lean epilogue (gemm8p.hip: epilogue_pass_bits) — the
// d = 1024 feed-forward (K = 1024), whose dH GEMM read 134-268 MB of activations per launch for their signs.  Taken where
// launch_gemm sends the plain GEMM of this shape to gemm8p's 256 x 256 or 128 x 256 tiles WITHOUT a split (the conditions of
// its `fills` / `half_m` branches, restated); PK_GEMM_RELU_BITS8P=0: off (A/B)
bool relu_bits_8p_ok(const void* A, const void* B, const void* C, const void* bias, long long M, long long N, long long K,
                     long long lda, long long ldb, long long ldc, int b_col) {
    static const bool on = [] { const char* e = getenv("PK_GEMM_RELU_BITS8P"); return !e || atoi(e) != 0; }();
    if (!on || !g_use_8p || M < 256 || N < 256 || K < 64 || K % 64 || N % 32) return false;
    auto al = [](const void* p, long long ld) { return ((uintptr_t)p % 16) == 0 && (ld % 8) == 0; };
    if (!al(A, lda) || !al(B, ldb) || !al(C, ldc) || (bias && (uintptr_t)bias % 16)) return false;
    if (!pk_gemm8p_eligible(M, N, K, lda, ldb, 0, b_col, 0)) return false;
    const long long t256 = ((M + 255) / 256) * ((N + 255) / 256), t_half = ((M + 127) / 128) * ((N + 255) / 256);
    return t256 >= 160 || t_half >= 160;
}
}  // namespace

extern "C" int pk_gemm_relu_bits_eligible(const void* A, const void* B, const void* C, const void* bias, long long M,
                                          long long N, long long K, long long lda, long long ldb, long long ldc,
                                          long long ldbits, int b_col, int mode, int dtype) {
    static const int on = [] { const char* e = getenv("PK_GEMM_RELU_BITS"); return (!e || atoi(e) != 0) ? 1 : 0; }();  // (diagnostic switch)
    if (!on) return 0;
    if ((dtype != PK_BF16 && dtype != PK_F16) || (mode != 0 && mode != 2) || N % 32 || ldbits < N / 8 || ldbits % 4) return 0;
    const EpiParams ep = relu_bits_epi(bias, const_cast<void*>(C), ldc, mode, 1.f);
    if (pk_gemmbs_eligible(A, B, C, M, N, K, lda, ldb, 0, b_col, &ep)) return 1;
    return relu_bits_8p_ok(A, B, C, bias, M, N, K, lda, ldb, ldc, b_col) ? 1 : 0;
}

extern "C" int pk_gemm_relu_bits(const void* A, const void* B, void* C, const void* bias, unsigned char* bits, long long M,
                                 long long N, long long K, long long lda, long long ldb, long long ldc, long long ldbits,
                                 int b_col, int mode, float alpha, int dtype, void* stream) {
    PK_CHECK_ARG(A && B && C && bits, "pk_gemm_relu_bits: null operand");
    PK_CHECK_ARG(pk_gemm_relu_bits_eligible(A, B, C, bias, M, N, K, lda, ldb, ldc, ldbits, b_col, mode, dtype),
                 "pk_gemm_relu_bits: M=%lld N=%lld K=%lld mode %d is not a shape of the B-stationary kernel; ask "
                 "pk_gemm_relu_bits_eligible and use pk_gemm with the activations as the mask operand", M, N, K, mode);
    EpiParams ep = relu_bits_epi(bias, C, ldc, mode, alpha);
    hipStream_t s = (hipStream_t)stream;
    if (!pk_gemmbs_eligible(A, B, C, M, N, K, lda, ldb, 0, b_col, &ep)) {  // the phase-interleaved kernel (relu_bits_8p_ok held)
        PK_CHECK_ARG(((uintptr_t)bits % 4) == 0, "pk_gemm_relu_bits: the mask rows must be 4-byte aligned");
        ep.aux = nullptr; ep.ldaux = 0;
        ep.bits = bits; ep.ldbits = ldbits;
        const long long t256 = ((M + 255) / 256) * ((N + 255) / 256);
        ep.half_m = t256 >= 160 ? 0 : 1;
        // (sample tag: the gemm8p instantiation | 0x1000 mask as bits)
        const bool pw = pk_gemm8p_is_pw(M, N, K, lda, ldb, 0, b_col, 1, 0, 0, &ep) != 0;  // (0x800: gemmpw.hip's persistent kernel)
        const bool pt = !pw && pk_gemm8p_is_pt(M, N, K, lda, ldb, 0, b_col, 1, 0, 0, &ep) != 0;  // (0x4000: the persistent walk of 256 x 256 tiles)
        GemmSample* sm = timing_begin(8 | 0x1000 | (pw ? 0x800 : pt ? 0x4000 : (ep.half_m ? 0x400 : 0)) | (mode == 2 ? 0x2000 : 0), 0, b_col, 1, dtype, M, N, K, s);
        const int rc = pk_gemm8p_launch(A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, 0, b_col, (int)K, 1, ep, dtype, stream);
        timing_end(sm, s);
        return rc == 1 ? 0 : (rc == 0 ? (pk_set_error("pk_gemm_relu_bits: the phase-interleaved kernel refused the shape"), -1) : rc);
    }
    ep.aux = nullptr;
    GemmSample* sm = timing_begin(0x200 | (int)(K / 64) | (PK_ACT_RELU << 4) | (mode == 2 ? 0x40 : 0) | 0x80, 0, b_col, 1, dtype, M, N, K, s);
    const int rc = pk_gemmbs_launch(A, B, C, M, N, K, lda, ldb, b_col, ep, dtype, stream, bits, ldbits);
    timing_end(sm, s);
    return rc == 1 ? 0 : rc;
}

extern "C" int pk_gemm_use_8p(int on) {
    const int old = g_use_8p;
    if (on >= 0) g_use_8p = on > 2 ? 2 : on;
    return old;
}

// ---- launch timing API (see include/pasero_hip.h) ----
extern "C" int pk_gemm_timing_start(int max_samples, int stride) {
    PK_CHECK_ARG(max_samples > 0 && stride > 0, "pk_gemm_timing_start: bad arguments");
    for (auto& sm : g_timing.pool) { (void)hipEventDestroy(sm.start); (void)hipEventDestroy(sm.stop); }
    g_timing.pool.assign((size_t)max_samples, GemmSample{});
    for (auto& sm : g_timing.pool) {
        hipError_t e = hipEventCreate(&sm.start);
        if (e == hipSuccess) e = hipEventCreate(&sm.stop);
        if (e != hipSuccess) { pk_set_error("pk_gemm_timing_start: %s", hipGetErrorString(e)); return (int)e; }
    }
    g_timing.stride = stride; g_timing.count = 0; g_timing.used = 0; g_timing.on = true;
    return 0;
}
extern "C" int pk_gemm_timing_stop(void) {
    g_timing.on = false;
    return g_timing.used;
}
extern "C" int pk_gemm_timing_read(int i, int* kernel, int* a_col, int* b_col, int* splitk, int* dtype, double* flops,
                                   float* ms) {
    PK_CHECK_ARG(i >= 0 && i < g_timing.used, "pk_gemm_timing_read: sample %d of %d", i, g_timing.used);
    const GemmSample& sm = g_timing.pool[i];
    hipError_t e = hipEventSynchronize(sm.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, sm.start, sm.stop);
    if (e != hipSuccess) { pk_set_error("pk_gemm_timing_read: %s", hipGetErrorString(e)); return (int)e; }
    *kernel = sm.kernel; *a_col = sm.a_col; *b_col = sm.b_col; *splitk = sm.splitk; *dtype = sm.dtype; *flops = sm.flops;
    return 0;
}
extern "C" int pk_gemm_timing_shape(int i, long long* M, long long* N, long long* K) {
    PK_CHECK_ARG(i >= 0 && i < g_timing.used && M && N && K, "pk_gemm_timing_shape: sample %d of %d", i, g_timing.used);
    const GemmSample& sm = g_timing.pool[i];
    *M = sm.M; *N = sm.N; *K = sm.K;
    return 0;
}
