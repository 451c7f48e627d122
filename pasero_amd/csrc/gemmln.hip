// y = LayerNorm(residual + dropout(x W^T + b)) in ONE kernel: the output projection of an attention block / the second
// feed-forward GEMM with the post-norm block end fused into its epilogue
//   reference: x = self.out_proj(x) / self.fc2(x); x = residual + dropout(x); x = LayerNorm(x)
//              pasero/models/modules.py:739, pasero/models/transformer.py:1018,1043-1048,1076-1086 (encoder),
//              :1322-1339,1389-1407 (decoder)
// Why: as separate launches the GEMM writes its (rows, d) output, and the LayerNorm pass reads it back together with the
// residual and writes z and y — five passes over a (rows, d) tensor of which the first two exist only to hand the GEMM
// result to the next kernel (11.6 % of the C2 step were LayerNorm passes at 4.3-4.5 TB/s).  LayerNorm needs whole rows,
// so the tile is 128 x 512 (d = 512: Transformer-base, the BASELINE C1 / C2 / C4 configurations): the same 65 536
// accumulators per workgroup as gemm8p.hip's 256 x 256 tile, 1.25x its L2->LDS bytes per FLOP (80 KiB per K-tile).
//
// K loop: gemm8p.hip's phase-interleaved schedule restated for a 1 x 4 arrangement of half-tiles.  8 waves = 2 (M) x 4 (N),
// wave (wr, wc) owns rows 64 wr + [0, 64) and columns {128 h + 32 wc + [0, 32)}, h = 0..3: four 64 x 32 quadrants, one
// per phase, 128 fp32 accumulators, v_mfma_f32_16x16x32 with swapped operands.  A K-tile (BK = 64) is FIVE 16-KiB
// half-tile images — A (128 rows) and B0..B3 (128 columns of W each) — in two stages = 160 KiB, all of the CU's LDS.
// The A fragments are read once per K-tile and serve all four phases.  Waves 4..7 run one barrier behind waves 0..3.
//   phase   fragment reads              MFMA quadrant   DMA issued (after the wait)          wait (vmcnt)
//   0       A(t)                 [8]    h = 0           B1(t+1)                              8
//   1       B1(t)                [4]    h = 1           B2(t+1)                              8
//   2       B2(t)                [4]    h = 2           B3(t+1), B0(t+2)                     6
//   3       B3(t), B0(t+1)       [8]    h = 3           A(t+2)                               8
// Every half-tile is issued 5 phases before its first read and overwrites a slot 3 phases after its last read; the
// counted waits retire exactly what the NEXT phase reads (in-order vmcnt: the number of DMA instructions issued after
// it, two per half-tile per wave).  Past the last K-tile the same instructions run against an empty descriptor.
// Epilogue: two passes of 64 rows through an fp32 staging image [64][516]; then one wave per row (lane = 8 columns):
// + bias, dropout (the same Philox (seed, offset, element) function as layernorm.hip: the stand-alone backward kernel
// regenerates the mask), + residual, z stored, statistics on the rounded z (what the backward re-reads), y stored.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "gemm8p_common.h"

namespace {

constexpr int LBM = 128, LBN = 512, LBK = 64;
constexpr int HALF = 16384, NSLOT = 5, STAGE = NSLOT * HALF, SMEM = 2 * STAGE;  // slots of a stage: A B0 B1 B2 B3
constexpr int SLOT_A = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_B2 = 3, SLOT_B3 = 4;
constexpr int CPL = LBN + 4;  // floats, pitch of the epilogue staging image
#ifndef PKLN_EROWS
#define PKLN_EROWS 32
#endif
constexpr int EROWS = PKLN_EROWS;  // rows per epilogue pass (32: 66 KiB of staging; 64: 129 KiB)

struct LnArgs {
    const void* bias;      // [512] or null
    const void* residual;  // [M][ldr] or null
    const void* gamma;     // [512]
    const void* beta;      // [512] or null
    void* z_out;           // [M][512] or null
    void* y_out;           // [M][512]
    float* mean_out;       // [M] or null (RMSNorm)
    float* rstd_out;       // [M]
    long long ldr;
    float eps, drop_scale;
    unsigned thr;
    unsigned long long seed, offset;
};

// SPEC: the epilogue's per-element decisions as template flags (decided at run time, "dropout?" and "residual?" become
// two selects per element: the epilogue is bound by its instruction count — 8 waves per CU, 223 vector instructions per
// row of 512): 0 = at run time, 1 = residual without dropout, 2 = residual and dropout.
// (Measured and not kept: the keep bits drawn inside the K loop — one Philox draw per two phases of the first eight
// K-tiles, five rounds behind each phase's DMAs, 8 bits per draw into four registers — so that the epilogue, 47 % of
// whose issue time with dropout on is the draw, only tests bits: results identical, fused block end at K = 512
// 35.7 -> 38.2 us, at K = 2048 68.3 -> 71.0: the quarter-rate multiplies of the draw hold the SIMD's issue port while
// the partner wave's MFMAs wait for it; the K loop slows down by more than the epilogue gains.)
template <typename T, int SPEC>
__global__ __launch_bounds__(512, 2) void gemm8p_ln_kernel(const T* __restrict__ A, const T* __restrict__ B, long long M,
                                                          long long K, long long lda, long long ldb, unsigned a_bytes,
                                                          unsigned b_bytes, LnArgs ln) {
    typedef typename M16<T>::vec V;
    typedef __attribute__((address_space(3))) void lds_void;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const long long m0 = (long long)blockIdx.x * LBM;
    const int nk = (int)(K / LBK);  // (the host sends whole K-tiles only)

    // ---- operand streams: per-lane offsets of this wave's two DMA pieces of every half-tile ----
    unsigned offa[2], offb[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offa[i] = src_offset<false>(wave * 2 + i, lane, lda, m0, M);
#pragma unroll
        for (int h = 0; h < 4; ++h) offb[h][i] = src_offset<false>(wave * 2 + i, lane, ldb, 128 * h, LBN);
    }
    auto dma = [&](int kt, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        char* dst = smem + (kt & 1) * STAGE + SLOT * HALF + wave * 2048;
        const bool live = kt < nk;
        const unsigned so = (unsigned)kt * (unsigned)(LBK * 2);
        if constexpr (SLOT == SLOT_A) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, live ? (int)a_bytes : 0, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, offa[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, offa[1], so, 0, 0);
        } else {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, live ? (int)b_bytes : 0, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, offb[SLOT - 1][0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, offb[SLOT - 1][1], so, 0, 0);
        }
    };

    f32x4 acc[4][4][2];  // [h][m-tile][n-tile]: D'[n][m] of the swapped product — lane: m = l & 15, n = 4 (l >> 4) + r
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    V fa[4][2], fb0[2][2][2], fb[2][2];  // A of the K-tile [m-tile][kk]; B0 per stage (read a phase early); B1..B3

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    auto load_a = [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fa[i][kk] = frag<T, false>(smem + S * STAGE + SLOT_A * HALF, wr * 64 + 16 * i, kk, lane);
    };
    auto load_b = [&](auto s_c, auto slot_c, V (&dst)[2][2]) {
        constexpr int S = decltype(s_c)::value, SLOT = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                dst[j][kk] = frag<T, false>(smem + S * STAGE + SLOT * HALF, wc * 32 + 16 * j, kk, lane);
    };
    auto mma = [&](auto h_c, V (&b)[2][2]) {
        constexpr int H = decltype(h_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[H][i][j] = M16<T>::mfma(b[j][kk], fa[i][kk], acc[H][i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
#define PK_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    // One phase.  P = 0..3 (quadrant h), S = stage of K-tile kt.
    auto phase = [&](auto p_c, auto s_c, int kt) {
        constexpr int P = decltype(p_c)::value, S = decltype(s_c)::value;
        using SN = std::integral_constant<int, S ^ 1>;
        if constexpr (P == 0) load_a(s_c);
        if constexpr (P == 1) load_b(s_c, I2{}, fb);  // B1
        if constexpr (P == 2) load_b(s_c, I3{}, fb);  // B2
        if constexpr (P == 3) {
            load_b(s_c, I4{}, fb);                    // B3
            load_b(SN{}, I1{}, fb0[S ^ 1]);           // B0 of the NEXT K-tile (other stage)
        }
        if constexpr (P == 2) PK_WAIT(6); else PK_WAIT(8);  // what the NEXT phase reads has landed
        if constexpr (P == 0) dma(kt + 1, I2{});
        if constexpr (P == 1) dma(kt + 1, I3{});
        if constexpr (P == 2) { dma(kt + 1, I4{}); dma(kt + 2, I1{}); }
        if constexpr (P == 3) dma(kt + 2, I0{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) mma(p_c, fb0[S]); else mma(p_c, fb);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    static_assert(SLOT_A == 0 && SLOT_B0 == 1 && SLOT_B1 == 2 && SLOT_B2 == 3 && SLOT_B3 == 4, "slot constants used above");

    if (nk > 0) {
        // prologue, in the order the steady state would have issued them: B0(0) A(0) | B1(0) B2(0) B3(0) B0(1) A(1)
        dma(0, I1{}); dma(0, I0{}); dma(0, I2{}); dma(0, I3{}); dma(0, I4{}); dma(1, I1{}); dma(1, I0{});
        PK_WAIT(10);  // B0, A of K-tile 0
        asm volatile("; PK8P_LOOP_BEGIN" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_b(I0{}, I1{}, fb0[0]);
        if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave of every SIMD runs one barrier behind the first
        for (int kt = 0; kt < nk; kt += 2) {
            phase(I0{}, I0{}, kt);
            phase(I1{}, I0{}, kt);
            phase(I2{}, I0{}, kt);
            phase(I3{}, I0{}, kt);
            if (kt + 1 >= nk) break;  // odd number of K-tiles
            phase(I0{}, I1{}, kt + 1);
            phase(I1{}, I1{}, kt + 1);
            phase(I2{}, I1{}, kt + 1);
            phase(I3{}, I1{}, kt + 1);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();  // the first half catches the barrier count up
        asm volatile("; PK8P_LOOP_END" ::: "memory");
        PK_WAIT(0);  // (the tail's DMAs are empty but still target LDS: retire them before it is reused)
    }
#undef PK_WAIT
    __syncthreads();

    // ---- epilogue: bias, dropout, residual, LayerNorm — one wave per row, lane = 8 consecutive columns ----
    // Passes of EROWS rows through the staging image; every wave then owns EROWS / 8 rows of the pass.
    float* cs = reinterpret_cast<float*>(smem);
    const int col = lane * 8;
    typedef typename H16<T>::vec HV;
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    float bia[8], gam[8], bet[8];
    {
        Vec16<T> gv = load16<T>(reinterpret_cast<const T*>(ln.gamma) + col), bv, cv;
        if (ln.beta) bv = load16<T>(reinterpret_cast<const T*>(ln.beta) + col);
        if (ln.bias) cv = load16<T>(reinterpret_cast<const T*>(ln.bias) + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gam[e] = gv.get(e);
            bet[e] = ln.beta ? bv.get(e) : 0.f;
            bia[e] = ln.bias ? cv.get(e) : 0.f;
        }
    }
    const T* res = reinterpret_cast<const T*>(ln.residual);
    T* zo = reinterpret_cast<T*>(ln.z_out);
    T* yo = reinterpret_cast<T*>(ln.y_out);
    const bool rms = ln.mean_out == nullptr;
    const bool use_res = SPEC == 0 ? res != nullptr : true, use_thr = SPEC == 0 ? ln.thr != 0 : SPEC == 2;
    constexpr int RPW = EROWS / 8;  // rows per wave per pass
    // Every residual row this wave will need is requested NOW, before the first store: vmcnt retires loads and stores in
    // one in-order queue, so a load issued behind the z / y stores of an earlier row would not return before those
    // stores are acknowledged — with every CU of the chip writing at once that is microseconds per row.
    Vec16<T> rv[LBM / 8];
    if (use_res) {
#pragma unroll
        for (int p = 0; p < LBM / EROWS; ++p)
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const long long gm = min(m0 + p * EROWS + wave * RPW + rr, M - 1);
                rv[p * RPW + rr] = load16<T>(res + gm * ln.ldr + col);
            }
    }
#pragma unroll
    for (int p = 0; p < LBM / EROWS; ++p) {
        constexpr int PPH = 64 / EROWS, MT = EROWS / 16;  // passes per row half, m-tiles per pass
        if (wr == p / PPH) {  // rows [EROWS p, EROWS (p + 1)): m-tiles MT (p % PPH) ... of the waves with wr == p / PPH
#pragma unroll
            for (int i2 = 0; i2 < MT; ++i2)
#pragma unroll
                for (int h = 0; h < 4; ++h)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 v = acc[h][MT * (p % PPH) + i2][j];
                        float* d = cs + (16 * i2 + (lane & 15)) * CPL + 128 * h + 32 * wc + 16 * j + 4 * (lane >> 4);
                        *reinterpret_cast<float4*>(d) = float4{v[0], v[1], v[2], v[3]};
                    }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int lr = wave * RPW + rr;
            const long long gm = m0 + p * EROWS + lr;
            if (gm >= M) continue;  // (uniform over the wave)
            const float* src = cs + lr * CPL + col;
            const float4 a4 = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
            float x[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
            const long long off = gm * LBN + col;
            bool keep[8];
            if (use_thr) {
                dropout_keep8(ln.seed, ln.offset, (unsigned long long)off >> 3, ln.thr, keep);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = x[e] + bia[e];
                if (use_thr) a = keep[e] ? a * ln.drop_scale : 0.f;
                if (use_res) a += rv[p * RPW + rr].get(e);
                x[e] = a;
            }
            // z is a tensor of the storage type in the reference (and what the backward pass re-reads): statistics on
            // the rounded values, as layernorm.hip takes them
            Vec16<T> zv;
            {
                f32x8 f = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
                zv.raw = __builtin_bit_cast(uint4, __builtin_convertvector(f, HV));
            }
            // z is only read again in the backward pass: a streaming store (C2 step 13.90 -> 13.85 ms); y feeds the next GEMM,
            // which re-reads it once per column strip — stored streaming, the step LOST 0.6 ms (14.49): it stays a plain store
            if (zo) store16_nt<T>(zo + off, zv);
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] = zv.get(e);
                sum += x[e];
            }
            const float mu = rms ? 0.f : wave_sum(sum) * (1.f / LBN);
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = x[e] - mu;
                sq += c * c;
            }
            const float rstd = rsqrtf(wave_sum(sq) * (1.f / LBN) + ln.eps);
            if (lane == 0) {
                if (!rms) ln.mean_out[gm] = mu;
                ln.rstd_out[gm] = rstd;
            }
            f32x8 yf;
#pragma unroll
            for (int e = 0; e < 8; ++e) yf[e] = (x[e] - mu) * rstd * gam[e] + bet[e];
            Vec16<T> yv;
            yv.raw = __builtin_bit_cast(uint4, __builtin_convertvector(yf, HV));
            store16<T>(yo + off, yv);
        }
        if (p + 1 < LBM / EROWS) __syncthreads();
    }
}

}  // namespace

// 1 if pk_gemm_ln_fwd takes the problem: d = 512 rows (whole rows in one tile), whole 64-deep K-tiles, 16-byte
// addressable 16-bit operands below 2 GiB
extern "C" int pk_gemm_ln_eligible(long long M, long long N, long long K, long long lda, long long ldb, int dtype) {
    if (dtype != PK_BF16 && dtype != PK_F16) return 0;
    if (N != LBN || M < 1 || K < LBK || K % LBK) return 0;
    if (lda % 8 || ldb % 8 || lda < K || ldb < K) return 0;
    const long long lim = 0x7FFFFFFFLL - 65536;
    return ((M - 1) * lda + K) * 2 <= lim && ((N - 1) * ldb + K) * 2 <= lim;
}

// the epilogue specialisation a problem runs on (gemm8p_ln_kernel's SPEC)
extern "C" int pk_gemmln_spec(int has_residual, float drop_p, long long K) {
    return !has_residual ? 0 : drop_p > 0.f ? 2 : 1;
}

// (the public entry pk_gemm_ln_fwd lives in gemm.hip, next to the launch-timing hooks of bench.py's roofline leg)
extern "C" int pk_gemmln_launch(const void* A, const void* W, const void* bias, const void* residual, const void* gamma,
                                const void* beta, void* z_out, void* y_out, float* mean, float* rstd, long long M,
                                long long N, long long K, long long lda, long long ldb, long long ldr, float eps,
                                float drop_p, unsigned long long seed, unsigned long long offset, int dtype,
                                void* stream) {
    if (M == 0) return 0;
    PK_CHECK_ARG(A && W && gamma && y_out && rstd, "pk_gemm_ln_fwd: null operand");
    PK_CHECK_ARG(pk_gemm_ln_eligible(M, N, K, lda, ldb, dtype),
                 "pk_gemm_ln_fwd: M=%lld N=%lld K=%lld lda=%lld ldb=%lld dtype %d is not eligible (N = 512, K %% 64 == 0, "
                 "16-bit): ask pk_gemm_ln_eligible and use pk_gemm + pk_residual_ln_fwd", M, N, K, lda, ldb, dtype);
    PK_CHECK_ARG(mean || !beta, "pk_gemm_ln_fwd: RMSNorm (mean == NULL) has no beta");
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_gemm_ln_fwd: bad dropout %f", drop_p);
    auto al = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
    PK_CHECK_ARG(al(A) && al(W) && al(bias) && al(residual) && al(gamma) && al(beta) && al(z_out) && al(y_out) &&
                 (!residual || ldr % 8 == 0), "pk_gemm_ln_fwd: operands must be 16-byte addressable");
    LnArgs ln;
    ln.bias = bias; ln.residual = residual; ln.gamma = gamma; ln.beta = beta;
    ln.z_out = z_out; ln.y_out = y_out; ln.mean_out = mean; ln.rstd_out = rstd;
    ln.ldr = ldr; ln.eps = eps;
    ln.thr = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    ln.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    ln.seed = seed; ln.offset = offset;
    const unsigned a_bytes = (unsigned)(((M - 1) * lda + K) * 2), b_bytes = (unsigned)(((N - 1) * ldb + K) * 2);
    dim3 grid((unsigned)((M + LBM - 1) / LBM)), block(512);
    hipStream_t s = (hipStream_t)stream;
    const int spec = pk_gemmln_spec(residual != nullptr, drop_p, K);
#define PK_LN_LAUNCH(TT, SP)                                                                                              \
    hipLaunchKernelGGL((gemm8p_ln_kernel<TT, SP>), grid, block, 0, s, (const TT*)A, (const TT*)W, M, K, lda, ldb, a_bytes, \
                       b_bytes, ln)
    if (dtype == PK_F16) {
        if (spec == 2) PK_LN_LAUNCH(f16, 2);
        else if (spec == 1) PK_LN_LAUNCH(f16, 1);
        else PK_LN_LAUNCH(f16, 0);
    } else {
        if (spec == 2) PK_LN_LAUNCH(bf16, 2);
        else if (spec == 1) PK_LN_LAUNCH(bf16, 1);
        else PK_LN_LAUNCH(bf16, 0);
    }
#undef PK_LN_LAUNCH
    PK_LAUNCH_CHECK();
    return 0;
}
