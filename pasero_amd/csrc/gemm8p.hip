// 256x256-tile bf16 / fp16 MFMA GEMM with a phase-interleaved ("ping-pong") K loop — the main GEMM of the training step.
// Same contract as gemm256.hip (C[m,n] = epi(alpha * sum_k A(m,k) B(n,k)), operands in row (k-contiguous) or col
// (m/n-contiguous) form, fused bias / activation / residual / act' epilogues, split-K slabs, fused bias gradient); what
// changes is how the K loop keeps the matrix cores fed.
//
// gemm256.hip runs all eight waves through "one barrier per K-tile": between two barriers every wave reads its
// fragments, issues the next tile's LDS-DMA and runs its 32 MFMAs, so the two waves of a SIMD contend for the same
// phases and every wave waits out the DMA's vmcnt(0).  Here (MI355X guide, "256^2 8-phase template"):
//   * 8 waves = 2 (M) x 4 (N); wave (wr, wc) owns rows {128h + 64wr + [0,64)} and cols {128h + 32wc + [0,32)}, h = 0, 1
//     of the tile: four 64x32 QUADRANTS (mh, nh), 128 fp32 accumulators per lane, v_mfma_f32_16x16x32 (the shape that
//     holds the higher clock under load);
//   * a K-tile (BK = 64) is FOUR half-tile images in LDS — A0, A1 (rows 0..127 / 128..255) and B0, B1, 16 KiB each,
//     two stages (even / odd K-tiles) = 128 KiB — and FOUR phases, one quadrant each: 16 MFMAs per wave per phase.
//     A phase = [load section: this phase's fragment reads + ONE half-tile of LDS-DMA for a later K-tile] s_barrier
//     [MFMA section] s_barrier.  Waves 4..7 run ONE barrier behind waves 0..3, so on every SIMD one wave is in its
//     MFMA section while its partner is in its load section: LDS reads, DMA issue and address arithmetic hide behind
//     the partner's MFMAs instead of competing with the wave's own;
//   * the DMA runs 5-6 phases (1.25-1.5 K-tiles) ahead and is retired by COUNTED waits — `s_waitcnt vmcnt(6)`: all but
//     the three youngest half-tiles have landed — never vmcnt(0) inside the loop; raw s_barrier only.
// Schedule of one iteration (two K-tiles t, t+1 in stages E, O); quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0):
//   phase   fragment reads            MFMA quadrant   DMA issued (after the wait)   last read of the slot it overwrites
//   1       A0(E)           [8]       (0,0)           B1(O) <- tile t+1             phase 6 of the previous iteration
//   2       B1(E)           [4]       (0,1)           A1(O) <- t+1                  phase 7 of the previous iteration
//   3       A1(E)           [8]       (1,1)           B0(E) <- t+2                  phase 8 of the previous iteration
//   4       B0(O) of t+1    [4]       (1,0)           A0(E) <- t+2                  phase 1
//   5       A0(O)           [8]       (0,0)           B1(E) <- t+2                  phase 2
//   6       B1(O)           [4]       (0,1)           A1(E) <- t+2                  phase 3
//   7       A1(O)           [8]       (1,1)           B0(O) <- t+3                  phase 4
//   8       B0(E) of t+2    [4]       (1,0)           A0(O) <- t+3                  phase 5
// (the B0 fragments of a K-tile are read one phase EARLY, into their own register set, in the otherwise read-free load
// section of the previous K-tile's last phase: 8 / 4 / 8 / 4 reads per phase instead of 12 / 4 / 8 / 0)
// Hazards.  RAW: a half-tile is read one phase (or more) after the wait that retires it — the wait sits in front of a
// barrier every wave passes before any wave's read.  WAR: a slot is overwritten >= 2 phases after its last read (the
// staggered half issues its reads one barrier late; their lgkmcnt(0) follows the next barrier).
// Tails: past the last K-tile the same DMA instructions are issued against an EMPTY descriptor (every lane out of
// range, nothing fetched) into slots that are already dead, so one loop body with one wait count serves every K; an
// odd number of K-tiles leaves the loop after phase 4.
// Operand staging: `buffer_load_dwordx4 ... lds` (one SRD per operand, per-lane 32-bit offsets, the K offset in an
// SGPR); rows / columns past the edge of the matrix re-read a valid one.  The LDS image is lane-linear, so the bank
// swizzle sits on the per-lane SOURCE offset and is undone on the read (row images: ds_read_b128, chunk ^ (row>>1)&7;
// col images: ds_read_b64_tr_b16, chunk ^ 2*((k&3) | ((k>>3)&1)<<2)).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <queue>
#include <vector>
#include <type_traits>
#include <vector>
#include "common.h"
#include "gemm_epi.h"

#include "gemm8p_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 16384, STAGE = 4 * HALF, SMEM = 2 * STAGE;  // slots of a stage: A0 A1 B0 B1
constexpr int SLOT_A0 = 0, SLOT_A1 = 1, SLOT_B0 = 2, SLOT_B1 = 3;
constexpr int CP = BN + 4;  // floats, epilogue staging pitch
// cache policy bits of the operand DMA (buffer_load aux: 0 default, 1 sc0, 2 nt, 16 sc1): experiment switches
#ifndef PK8P_AUX_A
#define PK8P_AUX_A 0
#endif
#ifndef PK8P_AUX_B
#define PK8P_AUX_B 0
#endif

// one 64-row pass of the epilogue: thread t owns 16-byte chunk t % 32 of rows t / 32 + 16*it   (as in gemm256.hip)
// (`av`: the pass's four aux chunks of this thread, loaded by the caller one pass AHEAD — loaded inside the pass, each of a
// tile's four passes waited out a memory round trip of its own: the mode 1 / 2 GEMMs ran 4-5 us per tile and round behind
// the same GEMM without aux)
template <typename T, int ACT, int MODE>
__device__ __forceinline__ void epilogue_pass(const float* __restrict__ cs, T* __restrict__ C, const EpiParams& ep,
                                              long long mh, long long n0, long long M, long long N, int tid,
                                              const Vec16<T> (&av)[4]) {
    const int col = (tid & 31) * 8, r0 = tid >> 5;
    const long long gn = n0 + col;
    if (gn + 8 > (ep.nstore ? ep.nstore : N)) return;  // (nstore: the rows of C are padded, the last chunk is stored whole)
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
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long long gm = mh + r0 + 16 * it;
        if (gm >= M) continue;
        const float* src = cs + (r0 + 16 * it) * CP + col;
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
#ifndef PK8P_PLAIN_MASK
#define PK8P_PLAIN_MASK 0 /* diagnostic: bit 0 = plain stores for mode 0 without activation, 1 = mode 1 (+ aux), 2 = ReLU / mode 2 */
#endif
        constexpr bool PLAIN = ((PK8P_PLAIN_MASK & 1) && MODE == 0 && ACT == PK_ACT_NONE) || ((PK8P_PLAIN_MASK & 2) && MODE == 1) ||
                               ((PK8P_PLAIN_MASK & 4) && (MODE == 2 || (MODE == 0 && ACT != PK_ACT_NONE)));
        if constexpr (PLAIN) store16<T>(C + gm * ep.ldc + gn, o);
        else store16_nt<T>(C + gm * ep.ldc + gn, o);
    }
}

// The ReLU feed-forward's mask as ONE BIT per element (EpiParams::bits; gemmbs.hip does the same for K = 512): the d = 1024
// dH = (dZ W2) * [h > 0] GEMM read the 134 MB of activations of a C5 layer (268 MB at C3) for their signs alone.
//   MODE 0: C = relu(alpha v + bias) and, beside it, the byte of this thread's eight columns — from the STORED 16-bit patterns
//           (what a later `h > 0` on the tensor would see); the four bytes of four neighbouring lanes leave as one dword store;
//   MODE 2: C = bit ? alpha v : 0, `bw[it]` = the dword that holds this thread's byte of row `it` (loaded one pass ahead by
//           the caller: four lanes read the same dword).
template <typename T, int MODE>
__device__ __forceinline__ void epilogue_pass_bits(const float* __restrict__ cs, T* __restrict__ C, const EpiParams& ep,
                                                   long long mh, long long n0, long long M, long long N, int tid,
                                                   const unsigned (&bw)[4]) {
    static_assert(MODE == 0 || MODE == 2, "fc1 forward / the masked dH GEMM");
    const int col = (tid & 31) * 8, r0 = tid >> 5;
    const long long gn = n0 + col;
    if (gn + 8 > N) return;  // (N % 32 == 0: the four lanes that share a mask dword leave together)
    float b[8];
    if (MODE == 0 && ep.bias) {
        Vec16<T> bv = load16<T>(reinterpret_cast<const T*>(ep.bias) + gn);
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = bv.get(e);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = 0.f;
    }
    const float alpha = ep.alpha;
    const int sh = 8 * (tid & 3);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long long gm = mh + r0 + 16 * it;
        if (gm >= M) continue;
        const float* src = cs + (r0 + 16 * it) * CP + col;
        const float4 a4 = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
        float x[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
        const unsigned mbyte = MODE == 2 ? (bw[it] >> sh) & 0xFFu : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = x[e] * alpha;
            if (MODE == 2) y = ((mbyte >> e) & 1u) ? y : 0.f;
            else y = fmaxf(y + b[e], 0.f);
            x[e] = y;
        }
        typedef __attribute__((ext_vector_type(8))) float f32x8;
        f32x8 f = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
        Vec16<T> o;
        o.raw = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename H16<T>::vec));
        store16_nt<T>(C + gm * ep.ldc + gn, o);
        if (MODE == 0) {
            // bit e = (the stored value > 0), without compares (as gemmbs.hip): after max(0, .) no 16-bit pattern has its sign
            // set, so adding 0x7FFF carries into bit 15 / 31 exactly where a half is not zero
            unsigned u = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // flags of dword w (elements 2 w, 2 w + 1) to bits 9 + 2 w and 25 + 2 w
                const unsigned ow = (&o.raw.x)[w];
                u |= ((ow + 0x7FFF7FFFu) >> (6 - 2 * w)) & (0x80008000u >> (6 - 2 * w));
            }
            const unsigned byte = ((u >> 9) | (u >> 24)) & 0xFFu;
            // the four lanes of a quad hold the bytes of 32 consecutive columns of one row: lane 0 of the quad collects them
            // by DPP (no LDS round trip) and stores ONE dword
            const unsigned b1 = (unsigned)__builtin_amdgcn_mov_dpp((int)byte, 0x55, 0xF, 0xF, true);   // quad_perm: lane 1 of the quad
            const unsigned b2 = (unsigned)__builtin_amdgcn_mov_dpp((int)byte, 0xAA, 0xF, 0xF, true);   // lane 2
            const unsigned b3 = (unsigned)__builtin_amdgcn_mov_dpp((int)byte, 0xFF, 0xF, 0xF, true);   // lane 3
            const unsigned wd = byte | (b1 << 8) | (b2 << 16) | (b3 << 24);
            if ((tid & 3) == 0) *reinterpret_cast<unsigned*>(ep.bits + gm * ep.ldbits + (gn >> 3)) = wd;
        }
    }
}

// The same 64-row pass for every other epilogue of pk_gemm (include/pasero_hip.h): any activation (GELU erf / tanh, SiLU:
// the Whisper / BLOOM / Llama feed-forward layers, pasero/models/modules.py:220-228), the pre-activation as a second output
// (`preact`: what act' needs in backward), mode 3 = act(v + bias) * aux (the SwiGLU / GEGLU gate, transformer.py:1011-1018)
// and mode 2 with a general act'.  Activation chosen at run time: these layers spend their time in erf / tanh anyway.
// (NOT inlined: erf / tanh / exp need registers the K loop cannot spare — inlined, this pass made hipcc spill inside the
// loop of every instantiation; as a call, its cost stays inside the call)
template <typename T>
__device__ __noinline__ void epilogue_pass_any(const float* __restrict__ cs, T* __restrict__ C, const EpiParams& ep,
                                                  long long mh, long long n0, long long M, long long N, int tid) {
    const int col = (tid & 31) * 8, r0 = tid >> 5;
    const long long gn = n0 + col;
    if (gn + 8 > N) return;
    float b[8];
    if (ep.mode != 2 && ep.bias) {
        Vec16<T> bv = load16<T>(reinterpret_cast<const T*>(ep.bias) + gn);
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = bv.get(e);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = 0.f;
    }
    const float alpha = ep.alpha;
    const int act = ep.act, mode = ep.mode;
    for (int it = 0; it < 4; ++it) {
        const long long gm = mh + r0 + 16 * it;
        if (gm >= M) continue;
        Vec16<T> av;
        if (mode != 0) av = load16<T>(reinterpret_cast<const T*>(ep.aux) + gm * ep.ldaux + gn);
        const float* src = cs + (r0 + 16 * it) * CP + col;
        const float4 a4 = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
        float x[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
        float pre[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = x[e] * alpha;
            pre[e] = 0.f;
            if (mode == 2) {
                y *= act_bwd_fast(act, av.get(e));
            } else {
                y += b[e];
                pre[e] = y;
                y = act_fwd_fast(act, y);
                if (mode == 1) y += av.get(e);
                else if (mode == 3) y *= av.get(e);
            }
            x[e] = y;
        }
        store16_nt<T>(C + gm * ep.ldc + gn, vec16_pack<T>(x));
        if (ep.preact && mode != 2) store16_nt<T>(reinterpret_cast<T*>(ep.preact) + gm * ep.ldpre + gn, vec16_pack<T>(pre));
    }
}

// ANY: the instantiation for the general epilogues (epilogue_pass_any); the lean one never calls a function, so it needs no
// scratch memory (a kernel with a call gets a stack, and the waves of a 1024-workgroup launch then start measurably slower)
// TAIL: the instantiation whose last K-tile may be partial (K % 64 != 0: the vocabulary dX GEMMs); the common one carries no
// per-lane masking in its DMA issue (VALU work in a load section is taken out of the partner wave's MFMA issue slots)
// `lin`: this workgroup's position in the slab-major (K-slab, tile) walk of ONE problem (the kernels below derive it from
// blockIdx through the XCD-contiguous remap; the grouped kernel subtracts the first workgroup of the problem)
// HM ("half M"): a 128 x 256 tile for outputs whose 256 x 256 tiles would fill half the chip (8192 x 1024: NLLB-1.3B's
// out-proj / cross-q / fc2 at C5).  The SAME eight-phase schedule with the second row half switched off: its DMA pieces are
// issued against an empty descriptor (so every counted wait still counts the same instructions), its fragment reads, its
// two quadrant MFMA sections and its two epilogue passes are compiled out.  Not the schedule one would design for this tile
// (two of four phases only synchronise), but every hazard distance of the full schedule holds a fortiori.
// pair mode's control words (grouped weight gradients; see "pair mode" in gemm8p_tile): where a lost hand-off is reported,
// how long the second workgroup waits, and the diagnostic that makes it wait in vain (tests)
struct PairCtl {
    unsigned* err;        // sticky error word in host-visible memory (the host reads it before the next grouped launch)
    unsigned spin_limit;  // polls of the partner's flag before the second workgroup gives up
    int drop_publish;     // diagnostic: the first workgroup does NOT raise its flag
};

// PW ("persistent walk", round 6): ONE workgroup per CU runs the tiles xcd_remap(blockIdx + j * gridDim) one after the other —
// row-form A, whole and an EVEN number of K-tiles, lean / mask-bit epilogues, no split — and the K-tile ring does not stop at a
// tile's end: the requests that the one-tile kernel issues against an empty descriptor behind its last K-tile fetch the NEXT
// tile's K-tile 0 (the same four half-tiles, the same stage: what the prologue would ask for), so a tile's epilogue runs with
// its successor's first operands on their way and the successor has no prologue; the epilogue stages through the OTHER stage
// (+ 1 KiB behind it).  Everything tile-dependent of the operand streams is an SGPR offset (the per-lane offsets are those of
// tile (0, 0); rows past M / N read as zeros through the descriptor's range check).
template <typename T, bool A_COL, bool B_COL, bool ANY, bool TAIL, bool HM = false, bool BITS = false, bool PW = false>
__device__ __forceinline__ void gemm8p_tile(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                            float* __restrict__ ws, float* __restrict__ asum_ws,
                                            T* __restrict__ asum_out, long long M, long long N, long long K,
                                            long long lda, long long ldb, int kchunk, unsigned a_bytes,
                                            unsigned b_bytes, int lin, unsigned long long* stamps,
                                            const EpiParams& ep, unsigned* pair_sync = nullptr,
                                            const PairCtl pctl = PairCtl{nullptr, 0u, 0}, int pw_total = 0) {
    typedef typename M16<T>::vec V;
    typedef __attribute__((address_space(3))) void lds_void;
    static_assert(!PW || (!A_COL && !ANY && !TAIL && !HM), "the persistent walk: row-form A, lean epilogues, whole K-tiles");
    constexpr int SMEM_T = PW ? SMEM + 1024 : SMEM;  // (PW: the epilogue's staging buffer = stage 1 + 1 KiB)
    __shared__ __attribute__((aligned(16))) char smem[SMEM_T];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    static_assert(!HM || !A_COL, "the half-M tile serves the row-form GEMMs (forward, dX)");
    constexpr int TM = HM ? BM / 2 : BM;  // rows of the tile
    const int nt_m = (int)((M + TM - 1) / TM), nt_n = (int)((N + BN - 1) / BN);
    // diagnostic build (-DPK8P_STAMPS, tools/gemm_phase_stamps.py): s_memrealtime at the seams of the tile, into a buffer
    // of their own (never into an output); the shipped build has no stamp
#ifdef PK8P_STAMPS
    int stamp_i = 0;
#define PK_STAMP() do { if (stamps && tid == 0) { stamps[(size_t)blockIdx.x * 64 + (stamp_i & 31)] = __builtin_amdgcn_s_memrealtime(); \
        stamps[(size_t)blockIdx.x * 64 + 32 + (stamp_i & 31)] = __builtin_amdgcn_s_memtime(); ++stamp_i; } } while (0)
#else
#define PK_STAMP() do { } while (0)
#endif
    // (tried: the workgroups of the odd XCDs started 4 / 8 / 12 us late, so that one half's epilogue stores meet the other
    // half's K loops — 8192 x 8192 x 1024 122.5 / 125.8 / 129.7 us against 122.6 without, the vocabulary logits 915-930 against
    // 925-931: nothing; the epilogue is not only the chip's write burst)
    int vb = lin;  // PW: the position in this workgroup's walk (blockIdx.x + j * gridDim.x); its tile: xcd_remap(vb, pw_total)
    bool pw_first = true;
    do {  // (one pass unless PW)
    const int lin_t = PW ? xcd_remap(vb, pw_total) : lin;
    PK_STAMP();  // tile start
    const int kslab = lin_t / (nt_m * nt_n);
    int t = lin_t % (nt_m * nt_n);
    const int tile_id = t;  // (the tile's position in the walk of its problem: what the two workgroups of a pair share)
    const int GROUP_M = nt_n <= 2 ? 8 : 4;
    int group_size = GROUP_M * nt_n, gid = t / group_size, first_m = gid * GROUP_M;
    int gsz = min(nt_m - first_m, GROUP_M);
    int tile_m = first_m + (t % group_size) % gsz, tile_n = (t % group_size) / gsz;
    const long long m0 = (long long)tile_m * TM, n0 = (long long)tile_n * BN;
    // PW: the next tile of the walk (its K-tile 0 is requested behind this tile's last K-tile)
    bool pw_has_next = false;
    unsigned pw_a_cur = 0u, pw_b_cur = 0u, pw_a_nxt = 0u, pw_b_nxt = 0u;  // byte offsets of the tiles' operand panels (SGPR operands)
    if constexpr (PW) {
        const int vn = vb + (int)gridDim.x;
        pw_has_next = vn < pw_total;
        const int ln = xcd_remap(pw_has_next ? vn : 0, pw_total);
        const int gid_n = ln / group_size, first_n = gid_n * GROUP_M, gsz_n = min(nt_m - first_n, GROUP_M);
        const int tm_n = first_n + (ln % group_size) % gsz_n, tn_n = (ln % group_size) / gsz_n;
        pw_a_cur = (unsigned)(m0 * lda * 2);
        pw_a_nxt = (unsigned)((long long)tm_n * TM * lda * 2);
        pw_b_cur = B_COL ? (unsigned)(n0 * 2) : (unsigned)(n0 * ldb * 2);
        pw_b_nxt = B_COL ? (unsigned)((long long)tn_n * BN * 2) : (unsigned)((long long)tn_n * BN * ldb * 2);
    }
    const long long kbeg = (long long)kslab * kchunk;
    const long long kend = min(K, kbeg + (long long)kchunk);
    // K-tiles of this slab; the last one may be partial (K % 8 == 0): col-form rows k >= K lie past the end of their
    // buffer and read as zeros, row-form chunks k >= K are masked per lane (below)
    const int nk = (int)((kend - kbeg + BK - 1) / BK);
    const int kvalid = (int)(kend - kbeg) - (nk - 1) * BK;  // depth of the last K-tile, 8..64

    // ---- operand streams: one buffer descriptor each, per-lane offsets of this wave's two DMA pieces per half-tile ----
    unsigned offa[2][2], offb[2][2];  // [half][piece of this wave]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (col form with M % 8 != 0: the host only sends it here when the rows are padded, lda >= M rounded up to 8)
            if constexpr (PW) {  // (those of tile (0, 0), no clamp: the tile is an SGPR offset, what lies past the edge reads as zeros)
                offa[h][i] = src_offset<A_COL>(wave * 2 + i, lane, lda, 128 * h, 1LL << 40);
                offb[h][i] = src_offset<B_COL>(wave * 2 + i, lane, ldb, 128 * h, 1LL << 40);
            } else {
                offa[h][i] = src_offset<A_COL>(wave * 2 + i, lane, lda, m0 + 128 * h, A_COL ? ((M + 7) & ~7LL) : M);
                offb[h][i] = src_offset<B_COL>(wave * 2 + i, lane, ldb, n0 + 128 * h, N);
            }
        }
    // row form, last K-tile: this lane's 16 bytes of piece i are columns k = 8 * chunk ... (the same chunk in both
    // halves and both operands: rows 8 * (2 wave + i) + (lane >> 3), swizzle (row >> 1) & 7)
    bool tail_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tail_ok[i] = (((lane & 7) ^ ((4 * i + (lane >> 4)) & 7)) * 8) < kvalid;
    constexpr unsigned DEAD_OFF = 0x80000000u;  // beyond every eligible operand (< 2 GiB), no 32-bit wrap with the K offset
    // byte step of one K-tile, and the K offset of tile 0 (an SGPR operand of the DMA)
    const unsigned kstep_a = A_COL ? (unsigned)(BK * lda * 2) : (unsigned)(BK * 2);
    const unsigned kstep_b = B_COL ? (unsigned)(BK * ldb * 2) : (unsigned)(BK * 2);
    const unsigned kbase_a = A_COL ? (unsigned)(kbeg * lda * 2) : (unsigned)(kbeg * 2);
    const unsigned kbase_b = B_COL ? (unsigned)(kbeg * ldb * 2) : (unsigned)(kbeg * 2);

    // half-tile `slot` of K-tile kt -> stage kt & 1.  Past the last K-tile the SAME two instructions are issued against
    // an empty descriptor (every lane out of range: nothing is fetched, zeros land), so the counted waits of the tail
    // are those of the steady state; their destination is a slot whose last reader has passed (the schedule's WAR
    // distance).
    auto dma = [&](int kt, int slot) {
        char* dst = smem + (kt & 1) * STAGE + slot * HALF + wave * 2048;
#ifdef PK8P_ABL_NODMA
        const bool live = kt < 0;  // ablation build: every DMA empty (the MFMAs chew on whatever LDS holds)
#else
        const bool pw_nx = PW && kt >= nk;  // (PW: behind the last K-tile — the next tile's K-tile 0, nothing beyond it)
        const bool live = pw_nx ? (pw_has_next && kt == nk) : kt < nk;
#endif
        const bool tail = TAIL && kt == nk - 1 && kvalid < BK;
        if (slot < 2) {
            const bool live_a = live && !(HM && slot == SLOT_A1);  // (half-M: the second row half is never fetched)
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, live_a ? (int)a_bytes : 0, 0x00020000);
            const unsigned so = PW ? (pw_nx ? pw_a_nxt : pw_a_cur + (unsigned)kt * kstep_a) : kbase_a + (unsigned)kt * kstep_a;
            unsigned v0 = offa[slot][0], v1 = offa[slot][1];
            if (TAIL && !A_COL && tail) { v0 = tail_ok[0] ? v0 : DEAD_OFF; v1 = tail_ok[1] ? v1 : DEAD_OFF; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, v0, so, 0, PK8P_AUX_A);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, v1, so, 0, PK8P_AUX_A);
        } else {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, live ? (int)b_bytes : 0, 0x00020000);
            const unsigned so = PW ? (pw_nx ? pw_b_nxt : pw_b_cur + (unsigned)kt * kstep_b) : kbase_b + (unsigned)kt * kstep_b;
            unsigned v0 = offb[slot - 2][0], v1 = offb[slot - 2][1];
            if (TAIL && !B_COL && tail) { v0 = tail_ok[0] ? v0 : DEAD_OFF; v1 = tail_ok[1] ? v1 : DEAD_OFF; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, v0, so, 0, PK8P_AUX_B);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, v1, so, 0, PK8P_AUX_B);
        }
    };

    f32x4 acc[2][4][2][2];  // [mh][m-tile][nh][n-tile]: D'[n][m] of the swapped product — lane: m = l & 15, n = 4 (l >> 4) + r
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragments of the current row half [m-tile][kk]; B fragments: B1 of the current K-tile, and B0 in TWO sets — the set
    // of stage S is filled one phase early, in the read-free load section of the previous K-tile's last phase, so the
    // fragment reads per phase are 8 / 4 / 8 / 4 instead of 12 / 4 / 8 / 0 (the 12-read section was the longest)
    V fa[4][2], fb0[2][2][2], fb1[2][2];  // fb0[stage][n-tile][kk], fb1[n-tile][kk]

    // col-form fragments: per-lane LDS byte address of tile i (A) / j (B) at k-step 0, slot 0, one set per stage; slot,
    // k-step and the k + 4 rows are instruction immediates (16-bit: the stage does not fit)
    unsigned ca[2][4], cb[2][2];
    {
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)smem;
        const int q = (lane & 15) >> 2, p = lane & 3, krow = 8 * (lane >> 4) + q;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = wr * 64 + 16 * i + 4 * p;
                ca[st][i] = base + st * STAGE + HT<true>::offset(krow, col >> 3) + (col & 7) * 2;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wc * 32 + 16 * j + 4 * p;
                cb[st][j] = base + st * STAGE + HT<true>::offset(krow, col >> 3) + (col & 7) * 2;
            }
        }
    }
    // inline asm, not the builtin: behind an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of the builtin's
    // transposed reads (it cannot tell which LDS bytes they touch), which would drain the prefetch every phase.  The
    // compiler does not wait for asm loads either: the phase's own `s_waitcnt lgkmcnt(0)` + sched_barrier stands
    // between these reads and the MFMAs (tools/check_asm_loads.py audits the .s for any other use of the destination
    // registers before that wait).
#define PK_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
    auto col_frag = [&](unsigned addr, auto off_c) -> V {
        constexpr int OFF = decltype(off_c)::value;
        static_assert(OFF >= 0 && OFF + 4 * HT<true>::ROWB < 65536, "ds offset field");
        s16x4 lo, hi;
        PK_TR(lo, addr, OFF);
        PK_TR(hi, addr, OFF + 4 * HT<true>::ROWB);  // rows k + 4: same swizzle
        s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(V, f);
    };

    // Fused bias gradient (col-form A = dY: the sum over k of every column), by the tile_n == 0 workgroups, ON THE MATRIX
    // CORES: one more MFMA per phase whose B operand is all ones — D'[n][m] = sum_k A(m, k) for every n.  Wave (wr, wc)
    // sums its m-tile i = wc (the four waves of a row half hold the same A fragments): quadrant phases 0, 1 take k-steps
    // 0, 1 of row half 0, phases 2, 3 those of row half 1, so every phase carries 17 MFMAs instead of 16.  (Round 3 summed
    // the staged A images with LDS reads + VALU adds in the load sections of phases 2 and 4: those workgroups ran their K
    // loop 25 % slower than the others — 189 against 146 us at the C2 encoder shapes, tools/gemm_phase_stamps.py — and a
    // launch lasts as long as its slowest workgroup.)
    const bool do_asum = A_COL && (asum_ws || asum_out) && tile_n == 0;
    f32x4 acc_s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // [row half]: lane l, any r: m = 16 wc + (l & 15)
    // (its A fragments are read a second time, into registers of their own, through an address computed once: choosing
    // among fa[0..3] by wc inside the loop made hipcc index the fragment array through scratch memory)
    V fs[2];
    unsigned cs_a[2] = {0u, 0u};
    if constexpr (A_COL) {
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)smem;
        const int q = (lane & 15) >> 2, p = lane & 3, krow = 8 * (lane >> 4) + q, col = wr * 64 + 16 * wc + 4 * p;
#pragma unroll
        for (int st = 0; st < 2; ++st) cs_a[st] = base + st * STAGE + HT<true>::offset(krow, col >> 3) + (col & 7) * 2;
    }
    V ones;
    {
        const short one = std::is_same<T, bf16>::value ? (short)0x3F80 : (short)0x3C00;
        s16x8 o = {one, one, one, one, one, one, one, one};
        ones = __builtin_bit_cast(V, o);
    }

    // One phase.  P = 0..3 (quadrant), S = 0 / 1 the K-tile's stage, (dma_kt, dma_slot) the half-tile staged here.
#define PK_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    auto load_a = [&](auto s_c, auto slot_c) {
        constexpr int S = decltype(s_c)::value, SLOT = decltype(slot_c)::value;
        if constexpr (A_COL) {
            if (do_asum) {
                fs[0] = col_frag(cs_a[S], std::integral_constant<int, SLOT * HALF>{});
                fs[1] = col_frag(cs_a[S], std::integral_constant<int, SLOT * HALF + 32 * HT<true>::ROWB>{});
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (A_COL) {
                fa[i][0] = col_frag(ca[S][i], std::integral_constant<int, SLOT * HALF>{});
                fa[i][1] = col_frag(ca[S][i], std::integral_constant<int, SLOT * HALF + 32 * HT<true>::ROWB>{});
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    fa[i][kk] = frag<T, false>(smem + S * STAGE + SLOT * HALF, wr * 64 + 16 * i, kk, lane);
            }
        }
    };
    auto load_b = [&](auto s_c, auto slot_c, V (&dst)[2][2]) {
        constexpr int S = decltype(s_c)::value, SLOT = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (B_COL) {
                dst[j][0] = col_frag(cb[S][j], std::integral_constant<int, SLOT * HALF>{});
                dst[j][1] = col_frag(cb[S][j], std::integral_constant<int, SLOT * HALF + 32 * HT<true>::ROWB>{});
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    dst[j][kk] = frag<T, false>(smem + S * STAGE + SLOT * HALF, wc * 32 + 16 * j, kk, lane);
            }
        }
    };
    auto mma = [&](int mh, int nh, V (&b)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#ifdef PK8P_ABL_NOMFMA
                    asm volatile("" :: "v"(b[j][kk]), "v"(fa[i][kk]));  // ablation build: fragments stay live, no MFMA
#else
                    acc[mh][i][nh][j] = M16<T>::mfma(b[j][kk], fa[i][kk], acc[mh][i][nh][j]);
#endif
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // RELAX (PW, the first two phases of a tile that is not the workgroup's first): everything these phases and the next ones
    // read landed before the previous tile's epilogue began; what sits in the queue in front of this tile's first requests are
    // that epilogue's stores (16 per thread or more), which the steady-state count would wait for
    bool pw_relax = false;  // (set at the start of a later tile, cleared behind its second phase: a scalar branch, not a second copy
    // of the loop body — that copy cost the persistent instantiations their last registers: 70-140 bytes of spills in the loop)
    auto phase = [&](auto p_c, auto s_c, int dma_kt, int dma_slot) {
        constexpr int P = decltype(p_c)::value, S = decltype(s_c)::value;
        using SN = std::integral_constant<int, S ^ 1>;
        if constexpr (P == 0) load_a(s_c, I0{});                  // A0 of this K-tile
        if constexpr (P == 1) load_b(s_c, I3{}, fb1);             // B1
        if constexpr (P == 2 && !HM) load_a(s_c, I1{});           // A1
        if constexpr (P == 3) load_b(SN{}, I2{}, fb0[S ^ 1]);     // B0 of the NEXT K-tile (other stage)
        if constexpr (PW && S == 0 && P < 2) {
            if (pw_relax) PK_WAIT(22);
            else PK_WAIT(6);
            if constexpr (P == 1) pw_relax = false;
        } else {
            PK_WAIT(6);  // all but the three youngest half-tiles have landed (what the NEXT phase reads is among them)
        }
        dma(dma_kt, dma_slot);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) mma(0, 0, fb0[S]);
        if constexpr (P == 1) mma(0, 1, fb1);
        if constexpr (P == 2 && !HM) mma(1, 1, fb1);
        if constexpr (P == 3 && !HM) mma(1, 0, fb0[S]);
        if constexpr (A_COL) {
            if (do_asum) {
                constexpr int MH = P >> 1, KK = P & 1;
                acc_s[MH] = M16<T>::mfma(ones, fs[KK], acc_s[MH]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    static_assert(SLOT_A0 == 0 && SLOT_A1 == 1 && SLOT_B0 == 2 && SLOT_B1 == 3, "slot constants used above");

    if (nk > 0) {
        // ---- prologue: K-tile 0 whole (PW, not the first tile: requested behind the previous tile's last K-tile and landed
        // before its epilogue began), K-tile 1's A0 B0 ----
        if (!PW || pw_first) { dma(0, SLOT_B0); dma(0, SLOT_A0); dma(0, SLOT_B1); dma(0, SLOT_A1); }
        dma(1, SLOT_B0); dma(1, SLOT_A0);
        if (!PW || pw_first) PK_WAIT(8);  // B0, A0 of tile 0 (PW, later tiles: landed long ago; the queue holds the last epilogue's stores)
        PK_STAMP();  // first K-tile landed
        asm volatile("; PK8P_LOOP_BEGIN" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_b(I0{}, I2{}, fb0[0]);  // B0 of K-tile 0 (every later one is read a phase ahead, inside the loop)
        if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave of every SIMD runs one barrier behind the first
        pw_relax = PW && !pw_first;
        for (int kt = 0; kt < nk; kt += 2) {
            phase(I0{}, I0{}, kt + 1, SLOT_B1);
            phase(I1{}, I0{}, kt + 1, SLOT_A1);
            phase(I2{}, I0{}, kt + 2, SLOT_B0);
            phase(I3{}, I0{}, kt + 2, SLOT_A0);
            if (kt + 1 >= nk) break;  // odd number of K-tiles
            phase(I0{}, I1{}, kt + 2, SLOT_B1);
            phase(I1{}, I1{}, kt + 2, SLOT_A1);
            phase(I2{}, I1{}, kt + 3, SLOT_B0);
            phase(I3{}, I1{}, kt + 3, SLOT_A0);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();  // the first half catches the barrier count up
        asm volatile("; PK8P_LOOP_END" ::: "memory");
        PK_WAIT(0);  // (trailing DMAs of the tail are empty, but they still target LDS: retire them before it is reused)
    }
#undef PK_WAIT
#undef PK_TR
    __syncthreads();
    PK_STAMP();  // K loop done

    // ---- pair mode (grouped weight gradients whose problem has exactly TWO K-slabs): no reduction launch.  The two workgroups
    // of a tile draw a ticket; the FIRST stores its fp32 partial write-through (`sc1`) and raises the tile's flag, the SECOND
    // waits for the flag, adds the first one's partial to its own accumulators as they pass through the staging buffer and
    // stores the finished 16-bit tile — one slab written and read instead of two, and own + other = other + own bit for bit,
    // so the result does not depend on who came first (and equals the reduction kernel's 0 + slab 0 + slab 1).  The hand-off
    // is MI355X_MICROARCH.md's valid form "sc1 payload stores -> every storing wave's vmcnt(0) -> workgroup barrier -> ONE
    // lane's relaxed agent-scope flag store" / "ONE relaxed poll -> ONE agent acquire -> vmcnt(0) -> workgroup barrier ->
    // plain loads", correct for any placement of the two workgroups.  The second workgroup can only wait for a workgroup that
    // is already running (it holds the even ticket), so the wait ends whatever the dispatch order; it is bounded all the same.
    // ticket and flag: pair_sync[2 tile], [2 tile + 1] in a buffer the library owns (zeroed once, when it is allocated).  No
    // per-launch reset: the ticket only ever counts up, every pair takes exactly two tickets — an even one (first) and the odd one
    // behind it (second) — and the first workgroup publishes by storing the SECOND's ticket value into the flag; the second
    // waits until the flag equals its own ticket.  Flag values of earlier launches are smaller tickets, never this one
    // (a memset node per launch cost 4.7 us of stream time each, more than the reduction launch it was to save).
    // A wait that ends WITHOUT the flag (a partner that died, tickets of another stream mixed in — neither can happen
    // while the buffer is used as the host code uses it) must not become a silently wrong weight gradient: the second
    // workgroup then writes the tile's id into a sticky error word the host reads before its next grouped launch
    // (pk_gemm8p_group_launch: RuntimeError through pk_last_error, tickets re-zeroed) and stores NaN instead of sums.
    int role = 0;  // 0: not a pair, 1: first to arrive (publishes its slab), 2: second (adds and finishes)
    bool lost = false;  // role 2 only: the partner's flag never came
    unsigned* sync_w = nullptr;
    unsigned ticket = 0;
    if constexpr (A_COL && B_COL && !ANY && !TAIL && !HM) {
        if (pair_sync) {
            sync_w = pair_sync + 2 * tile_id;
            unsigned* lw = reinterpret_cast<unsigned*>(smem + SMEM - 16);  // (the stages are dead; staging uses the first 66 KiB)
            if (tid == 0) *lw = __hip_atomic_fetch_add(sync_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)*lw);
            role = (ticket & 1u) ? 2 : 1;
            if (role == 2) {
                if (tid == 0) {
                    unsigned got = 0u;
                    for (unsigned spins = 0; spins < pctl.spin_limit; ++spins) {  // (2^22: ~0.5 s — a lost partner must not hang the GPU)
                        if (__hip_atomic_load(sync_w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ticket) { got = 1u; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lw[1] = got;
                    if (!got && pctl.err)
                        __hip_atomic_store(pctl.err, 0x80000000u | (unsigned)tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __syncthreads();
                lost = __builtin_amdgcn_readfirstlane((int)lw[1]) == 0;
            }
        }
    }

    if constexpr (A_COL) {
        if (do_asum) {  // lanes 0..15 of every wave hold the sums of rows 128 mh + 64 wr + 16 wc + lane
            float* red = reinterpret_cast<float*>(smem);  // [256]
            if (lane < 16) {
                red[64 * wr + 16 * wc + lane] = acc_s[0][0];
                red[128 + 64 * wr + 16 * wc + lane] = acc_s[1][0];
            }
            __syncthreads();
            if (tid < BM && m0 + tid < M) {
                const float sum = red[tid];
                if (role == 1) {  // (part of what the flag publishes)
                    __hip_atomic_store(asum_ws + (long long)kslab * M + m0 + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (role == 2) {
                    const float other = lost ? __builtin_nanf("") : asum_ws[(long long)(1 - kslab) * M + m0 + tid];
                    asum_out[m0 + tid] = from_f32<T>(sum + other);
                } else if (asum_ws) asum_ws[(long long)kslab * M + m0 + tid] = sum;
                else asum_out[m0 + tid] = from_f32<T>(sum);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: four 64-row passes of the accumulators through the fp32 staging buffer (all stages are free) ----
    float* cs = reinterpret_cast<float*>(smem + (PW ? STAGE : 0));  // (PW: stage 0 holds the next tile's K-tile 0)
    // the aux operand of the lean mode 1 / 2 epilogues, one pass ahead of its use (aux may alias C: a chunk is read by the
    // thread that stores it, and before that store)
    // (BITS: an instantiation of its own — mode 0 + ReLU writes the mask bits, mode 2 + ReLU reads them; as a run-time branch in
    // the common instantiations it cost the d = 512 dX GEMMs 2 us per launch)
    static_assert(!BITS || (!A_COL && !ANY && !TAIL), "the mask bits ride on the lean row-form instantiations");
    constexpr bool use_bits = BITS;
    const bool pre_aux = !PW && !ws && !ANY && ep.mode != 0 && !use_bits;  // (PW: mode 0 or the mask bits only)
    const bool pre_bits = use_bits && !ws && ep.mode == 2;
    unsigned bw_next[4] = {0u, 0u, 0u, 0u};
    auto bits_load = [&](int p) {  // this thread's mask dwords of pass p (four lanes share one)
        const int col = (tid & 31) * 8, r0 = tid >> 5;
        const long long gn = n0 + col;
        if (gn + 8 > N) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long long gm = m0 + p * 64 + r0 + 16 * it;
            if (gm < M) bw_next[it] = *reinterpret_cast<const unsigned*>(ep.bits + gm * ep.ldbits + ((gn >> 3) & ~3LL));
        }
    };
    if (pre_bits) bits_load(0);
    Vec16<T> av_next[4];
    auto aux_load = [&](int p) {
        const int col = (tid & 31) * 8, r0 = tid >> 5;
        const long long gn = n0 + col;
        if (gn + 8 > N) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long long gm = m0 + p * 64 + r0 + 16 * it;
            if (gm < M) av_next[it] = load16<T>(reinterpret_cast<const T*>(ep.aux) + gm * ep.ldaux + gn);
        }
    };
    if (pre_aux) aux_load(0);
    constexpr int NPASS = HM ? 2 : 4;
    // pair mode, second workgroup: this thread's chunks of the first workgroup's slab, one pass ahead of their use
    float4 oth[4][2];
    auto other_load = [&](int p) {
        const int col = (tid & 31) * 8, r0 = tid >> 5;
        const long long gn = n0 + col;
        if (gn + 8 > N) return;
        const float* oslab = ws + (long long)(1 - kslab) * M * N;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long long gm = m0 + p * 64 + r0 + 16 * it;
            if (gm < M) {
                oth[it][0] = *reinterpret_cast<const float4*>(oslab + gm * N + gn);
                oth[it][1] = *reinterpret_cast<const float4*>(oslab + gm * N + gn + 4);
            }
        }
    };
    if (role == 2) other_load(0);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (wr == (p & 1)) {  // tile rows [64p, 64p + 64) = row half p >> 1 of the waves with wr == p & 1
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 v = acc[p >> 1][i][nh][j];
                        float* d = cs + (16 * i + (lane & 15)) * CP + 128 * nh + 32 * wc + 16 * j + 4 * (lane >> 4);
                        *reinterpret_cast<float4*>(d) = float4{v[0], v[1], v[2], v[3]};
                    }
        }
        __syncthreads();
        const long long mh = m0 + p * 64;
        Vec16<T> av[4];
        if (pre_aux) {
#pragma unroll
            for (int it = 0; it < 4; ++it) av[it] = av_next[it];
            if (p + 1 < NPASS) aux_load(p + 1);
        }
        unsigned bw[4] = {bw_next[0], bw_next[1], bw_next[2], bw_next[3]};
        if constexpr (BITS) {
            if (pre_bits && p + 1 < NPASS) bits_load(p + 1);
        }
        if (ws && role == 2) {  // the pair's second workgroup: own partial (staging buffer) + the first one's slab -> C
            const int col = (tid & 31) * 8, r0 = tid >> 5;
            const long long gn = n0 + col;
            float4 cur[4][2];
#pragma unroll
            for (int it = 0; it < 4; ++it) { cur[it][0] = oth[it][0]; cur[it][1] = oth[it][1]; }
            if (lost) {  // (wave-uniform; never taken in a healthy launch)
                const float nan = __builtin_nanf("");
#pragma unroll
                for (int it = 0; it < 4; ++it) { cur[it][0] = float4{nan, nan, nan, nan}; cur[it][1] = cur[it][0]; }
            }
            if (p + 1 < NPASS) other_load(p + 1);  // (one pass ahead, as the aux operand of the single GEMMs)
            if (gn + 8 <= N) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const long long gm = mh + r0 + 16 * it;
                    if (gm >= M) continue;
                    const float* src = cs + (r0 + 16 * it) * CP + col;
                    const float4 a4 = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
                    float v[8] = {a4.x + cur[it][0].x, a4.y + cur[it][0].y, a4.z + cur[it][0].z, a4.w + cur[it][0].w,
                                  b4.x + cur[it][1].x, b4.y + cur[it][1].y, b4.z + cur[it][1].z, b4.w + cur[it][1].w};
                    store16_nt<T>(C + gm * ep.ldc + gn, vec16_pack<T>(v));
                }
            }
        } else if (ws && role == 1) {  // the pair's first workgroup: its slab, write-through
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, (int)(2 * M * N * 4), 0x00020000);
            const int col = (tid & 31) * 8, r0 = tid >> 5;
            const long long gn = n0 + col;
            if (gn + 8 <= N) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const long long gm = mh + r0 + 16 * it;
                    if (gm >= M) continue;
                    const float* src = cs + (r0 + 16 * it) * CP + col;
                    const unsigned off = (unsigned)((((long long)kslab * M + gm) * N + gn) * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(src), rs, (int)off, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(src + 4), rs, (int)(off + 16), 0, 16);
                }
            }
        } else if (ws) {  // split-K partial: raw fp32 slab [splitk][M][N]
            float* slab = ws + (long long)kslab * M * N;
            const int col = (tid & 31) * 8, r0 = tid >> 5;
            const long long gn = n0 + col;
            if (gn + 8 <= N) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const long long gm = mh + r0 + 16 * it;
                    if (gm >= M) continue;
                    const float* src = cs + (r0 + 16 * it) * CP + col;
                    float* dst = slab + gm * N + gn;
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
                    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
                }
            }
        } else if constexpr (ANY) {
            epilogue_pass_any<T>(cs, C, ep, mh, n0, M, N, tid);
        } else if constexpr (BITS) {
            if (ep.mode == 0) epilogue_pass_bits<T, 0>(cs, C, ep, mh, n0, M, N, tid, bw);
            else epilogue_pass_bits<T, 2>(cs, C, ep, mh, n0, M, N, tid, bw);
        } else if (PW || ep.mode == 0) {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 0>(cs, C, ep, mh, n0, M, N, tid, av);
            else epilogue_pass<T, PK_ACT_NONE, 0>(cs, C, ep, mh, n0, M, N, tid, av);
        } else if (ep.mode == 1) {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 1>(cs, C, ep, mh, n0, M, N, tid, av);
            else epilogue_pass<T, PK_ACT_NONE, 1>(cs, C, ep, mh, n0, M, N, tid, av);
        } else {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 2>(cs, C, ep, mh, n0, M, N, tid, av);
            else epilogue_pass<T, PK_ACT_NONE, 2>(cs, C, ep, mh, n0, M, N, tid, av);
        }
        if (p + 1 < NPASS) __syncthreads();
    }
    if (role == 1) {  // publish: every storing wave drains its write-through stores, then ONE lane raises the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && !pctl.drop_publish)
            __hip_atomic_store(sync_w + 1, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef PK8P_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PK_STAMP();  // epilogue stores acknowledged
#endif
    if constexpr (PW) {
        vb += (int)gridDim.x;
        pw_first = false;
        if (vb < pw_total) __syncthreads();  // the staging buffer (stage 1) has been read: the next tile's requests may overwrite it
    }
    } while (PW && vb < pw_total);
#undef PK_STAMP
}

// the persistent walk of the 256 x 256 tile (gemm8p_tile's PW form): lean and mask-bit epilogues, row-form A
template <typename T, bool B_COL, bool BITS>
__global__ __launch_bounds__(512, 2) void gemm8p_pt_kernel(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                                          long long M, long long N, long long K, long long lda, long long ldb,
                                                          unsigned a_bytes, unsigned b_bytes, int total,
                                                          unsigned long long* stamps, EpiParams ep) {
    gemm8p_tile<T, false, B_COL, false, false, false, BITS, true>(A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, (int)K, a_bytes,
                                                                  b_bytes, (int)blockIdx.x, stamps, ep, nullptr,
                                                                  PairCtl{nullptr, 0u, 0}, total);
}

template <typename T, bool A_COL, bool B_COL, bool ANY, bool TAIL, bool HM = false, bool BITS = false>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                       T* __restrict__ C, float* __restrict__ ws,
                                                       float* __restrict__ asum_ws, T* __restrict__ asum_out,
                                                       long long M, long long N, long long K, long long lda,
                                                       long long ldb, int kchunk, unsigned a_bytes, unsigned b_bytes,
                                                       int total, unsigned long long* stamps, EpiParams ep) {
    // slab-major (K-slab, tile) walk over the XCD-contiguous remap: an XCD owns whole K-slabs
    gemm8p_tile<T, A_COL, B_COL, ANY, TAIL, HM, BITS>(A, B, C, ws, asum_ws, asum_out, M, N, K, lda, ldb, kchunk, a_bytes, b_bytes,
                                                      xcd_remap(blockIdx.x, gridDim.x), stamps, ep);
}

// ---- 128 x 256 tiles with a schedule of their own (round 4) ----
// Outputs of 80..159 256-tiles (8192 x 1024: NLLB-1.3B's out-proj / cross-q / fc2 and their dX at C5) fill half the chip as
// 256 x 256 tiles.  The half-M switch of gemm8p_tile (HM) gives every CU a 128 x 256 tile but keeps four phases per K-tile, two
// of which only synchronise (1.0 us per K-tile).  This is the schedule made for that tile: row-form A only (forward, dX).
//   * 8 waves = 2 (M) x 4 (N) as above; wave (wr, wc) owns rows 64 wr + [0, 64) and columns 128 h + 32 wc + [0, 32), h = 0, 1:
//     TWO 64 x 32 quadrants, 64 fp32 accumulators per lane;
//   * a K-tile is THREE half-tile images — A0 (128 rows), B0, B1 — in a ring of THREE stages (144 KiB) and TWO phases of 16
//     MFMAs per wave; waves 4..7 run one barrier behind waves 0..3 as above;
//   * per-image rings, every DMA three phases (1.5 K-tiles) ahead of the wait that retires it, every wait `vmcnt(6)`:
//       phase   fragment reads                    MFMA quadrant   DMA issued (after the wait)          last read of the slot
//       A(t)    A0(t)                      [8]    (0, 0)          A0(t+2)                             A(t-1)
//       B(t)    B1(t), B0(t+1)             [8]    (0, 1)          B1(t+2), B0(t+3)                    B(t-1), B(t-1)
//     RAW: the wait at the end of a phase's load section leaves the three youngest half-tiles in flight; what the NEXT
//     phase reads is older (A(t): in flight A0(t+1) B1(t+1) B0(t+2), B(t) reads B1(t) B0(t+1); B(t): in flight B1(t+1)
//     B0(t+2) A0(t+2), A(t+1) reads A0(t+1)).  WAR: every slot is overwritten two phases after its last read.
//   * tails as above: past the last K-tile the same DMA instructions go against an empty descriptor.
namespace hm2 {
constexpr int HSTAGE = 3 * HALF, HSMEM = 3 * HSTAGE;
constexpr int S_A = 0, S_B0 = 1, S_B1 = 2;
}  // namespace hm2

template <typename T, bool B_COL, bool TAIL, bool BITS = false>
__device__ __forceinline__ void gemm8p_hm2_tile(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                                long long M, long long N, long long K, long long lda, long long ldb,
                                                unsigned a_bytes, unsigned b_bytes, int lin, const EpiParams& ep) {
    using namespace hm2;
    typedef typename M16<T>::vec V;
    typedef __attribute__((address_space(3))) void lds_void;
    __shared__ __attribute__((aligned(16))) char smem[HSMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    constexpr int TM = BM / 2;
    const int nt_m = (int)((M + TM - 1) / TM), nt_n = (int)((N + BN - 1) / BN);
    const int GROUP_M = nt_n <= 2 ? 8 : 4;
    const int group_size = GROUP_M * nt_n, gid = lin / group_size, first_m = gid * GROUP_M;
    const int gsz = min(nt_m - first_m, GROUP_M);
    const int tile_m = first_m + (lin % group_size) % gsz, tile_n = (lin % group_size) / gsz;
    const long long m0 = (long long)tile_m * TM, n0 = (long long)tile_n * BN;
    const int nk = (int)((K + BK - 1) / BK);
    const int kvalid = (int)K - (nk - 1) * BK;  // depth of the last K-tile, 8..64

    unsigned offa[2], offb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offa[i] = src_offset<false>(wave * 2 + i, lane, lda, m0, M);
#pragma unroll
        for (int h = 0; h < 2; ++h) offb[h][i] = src_offset<B_COL>(wave * 2 + i, lane, ldb, n0 + 128 * h, N);
    }
    bool tail_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tail_ok[i] = (((lane & 7) ^ ((4 * i + (lane >> 4)) & 7)) * 8) < kvalid;
    constexpr unsigned DEAD_OFF = 0x80000000u;
    const unsigned kstep_b = B_COL ? (unsigned)(BK * ldb * 2) : (unsigned)(BK * 2);

    auto dma = [&](int kt, auto slot_c, auto stage_c) {
        constexpr int SLOT = decltype(slot_c)::value, ST = decltype(stage_c)::value;
        char* dst = smem + ST * HSTAGE + SLOT * HALF + wave * 2048;
        const bool live = kt < nk;
        const bool tail = TAIL && kt == nk - 1 && kvalid < BK;
        if constexpr (SLOT == S_A) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, live ? (int)a_bytes : 0, 0x00020000);
            const unsigned so = (unsigned)kt * (unsigned)(BK * 2);
            unsigned v0 = offa[0], v1 = offa[1];
            if (TAIL && tail) { v0 = tail_ok[0] ? v0 : DEAD_OFF; v1 = tail_ok[1] ? v1 : DEAD_OFF; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, v0, so, 0, PK8P_AUX_A);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, v1, so, 0, PK8P_AUX_A);
        } else {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, live ? (int)b_bytes : 0, 0x00020000);
            const unsigned so = (unsigned)kt * kstep_b;
            unsigned v0 = offb[SLOT - 1][0], v1 = offb[SLOT - 1][1];
            if (TAIL && !B_COL && tail) { v0 = tail_ok[0] ? v0 : DEAD_OFF; v1 = tail_ok[1] ? v1 : DEAD_OFF; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, v0, so, 0, PK8P_AUX_B);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, v1, so, 0, PK8P_AUX_B);
        }
    };

    f32x4 acc[4][2][2];  // [m-tile][nh][n-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    V fa[4][2], fb0[2][2], fb1[2][2];

    unsigned cb[3][2];  // col-form B: per-lane LDS address of n-tile j at k-step 0, slot 0, per stage
    {
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)smem;
        const int q = (lane & 15) >> 2, p = lane & 3, krow = 8 * (lane >> 4) + q;
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wc * 32 + 16 * j + 4 * p;
                cb[st][j] = base + st * HSTAGE + HT<true>::offset(krow, col >> 3) + (col & 7) * 2;
            }
    }
#define PK_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
    auto col_frag = [&](unsigned addr, auto off_c) -> V {
        constexpr int OFF = decltype(off_c)::value;
        static_assert(OFF >= 0 && OFF + 4 * HT<true>::ROWB < 65536, "ds offset field");
        s16x4 lo, hi;
        PK_TR(lo, addr, OFF);
        PK_TR(hi, addr, OFF + 4 * HT<true>::ROWB);
        s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(V, f);
    };
#define PK_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    auto load_a = [&](auto st_c) {
        constexpr int ST = decltype(st_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[i][kk] = frag<T, false>(smem + ST * HSTAGE + S_A * HALF, wr * 64 + 16 * i, kk, lane);
    };
    auto load_b = [&](auto st_c, auto slot_c, V (&dst)[2][2]) {
        constexpr int ST = decltype(st_c)::value, SLOT = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (B_COL) {
                dst[j][0] = col_frag(cb[ST][j], std::integral_constant<int, SLOT * HALF>{});
                dst[j][1] = col_frag(cb[ST][j], std::integral_constant<int, SLOT * HALF + 32 * HT<true>::ROWB>{});
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    dst[j][kk] = frag<T, false>(smem + ST * HSTAGE + SLOT * HALF, wc * 32 + 16 * j, kk, lane);
            }
        }
    };
    auto mma = [&](int nh, V (&b)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][nh][j] = M16<T>::mfma(b[j][kk], fa[i][kk], acc[i][nh][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    using IA = std::integral_constant<int, S_A>; using IB0 = std::integral_constant<int, S_B0>;
    using IB1 = std::integral_constant<int, S_B1>;
    auto tile = [&](auto st_c, int kt) {
        constexpr int ST = decltype(st_c)::value;
        using S1 = std::integral_constant<int, (ST + 1) % 3>; using S2 = std::integral_constant<int, (ST + 2) % 3>;
        // ---- phase A: quadrant (0, 0) ----
        load_a(st_c);
        PK_WAIT(6);
        dma(kt + 2, IA{}, S2{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B: quadrant (0, 1); the B0 fragments of the NEXT K-tile ----
        load_b(st_c, IB1{}, fb1);
        load_b(S1{}, IB0{}, fb0);
        PK_WAIT(6);
        dma(kt + 2, IB1{}, S2{});
        dma(kt + 3, IB0{}, st_c);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (nk > 0) {
        dma(0, IB0{}, I0{}); dma(0, IA{}, I0{}); dma(0, IB1{}, I0{});
        dma(1, IB0{}, I1{}); dma(1, IA{}, I1{}); dma(1, IB1{}, I1{});
        dma(2, IB0{}, I2{});
        PK_WAIT(10);  // B0, A0 of tile 0
        asm volatile("; PK8P_LOOP_BEGIN" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_b(I0{}, IB0{}, fb0);
        if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave of every SIMD runs one barrier behind the first
        for (int kt = 0; kt < nk; kt += 3) {
            tile(I0{}, kt);
            if (kt + 1 >= nk) break;
            tile(I1{}, kt + 1);
            if (kt + 2 >= nk) break;
            tile(I2{}, kt + 2);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        asm volatile("; PK8P_LOOP_END" ::: "memory");
        PK_WAIT(0);
    }
#undef PK_WAIT
#undef PK_TR
    __syncthreads();

    // ---- epilogue: two 64-row passes through the fp32 staging buffer, the aux operand one pass ahead (as gemm8p_tile) ----
    float* cs = reinterpret_cast<float*>(smem);
    constexpr bool use_bits = BITS;  // (an instantiation of its own, as in gemm8p_tile)
    const bool pre_aux = ep.mode != 0 && !use_bits;
    const bool pre_bits = use_bits && ep.mode == 2;
    unsigned bw_next[4] = {0u, 0u, 0u, 0u};
    auto bits_load = [&](int p) {
        const int col = (tid & 31) * 8, r0 = tid >> 5;
        const long long gn = n0 + col;
        if (gn + 8 > N) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long long gm = m0 + p * 64 + r0 + 16 * it;
            if (gm < M) bw_next[it] = *reinterpret_cast<const unsigned*>(ep.bits + gm * ep.ldbits + ((gn >> 3) & ~3LL));
        }
    };
    if (pre_bits) bits_load(0);
    Vec16<T> av_next[4];
    auto aux_load = [&](int p) {
        const int col = (tid & 31) * 8, r0 = tid >> 5;
        const long long gn = n0 + col;
        if (gn + 8 > N) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long long gm = m0 + p * 64 + r0 + 16 * it;
            if (gm < M) av_next[it] = load16<T>(reinterpret_cast<const T*>(ep.aux) + gm * ep.ldaux + gn);
        }
    };
    if (pre_aux) aux_load(0);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (wr == p) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 v = acc[i][nh][j];
                        float* d = cs + (16 * i + (lane & 15)) * CP + 128 * nh + 32 * wc + 16 * j + 4 * (lane >> 4);
                        *reinterpret_cast<float4*>(d) = float4{v[0], v[1], v[2], v[3]};
                    }
        }
        __syncthreads();
        const long long mh = m0 + p * 64;
        Vec16<T> av[4];
        if (pre_aux) {
#pragma unroll
            for (int it = 0; it < 4; ++it) av[it] = av_next[it];
            if (p == 0) aux_load(1);
        }
        unsigned bw[4] = {bw_next[0], bw_next[1], bw_next[2], bw_next[3]};
        if constexpr (BITS) {
            if (pre_bits && p == 0) bits_load(1);
        }
        if constexpr (BITS) {
            if (ep.mode == 0) epilogue_pass_bits<T, 0>(cs, C, ep, mh, n0, M, N, tid, bw);
            else epilogue_pass_bits<T, 2>(cs, C, ep, mh, n0, M, N, tid, bw);
        } else if (ep.mode == 0) {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 0>(cs, C, ep, mh, n0, M, N, tid, av);
            else epilogue_pass<T, PK_ACT_NONE, 0>(cs, C, ep, mh, n0, M, N, tid, av);
        } else if (ep.mode == 1) {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 1>(cs, C, ep, mh, n0, M, N, tid, av);
            else epilogue_pass<T, PK_ACT_NONE, 1>(cs, C, ep, mh, n0, M, N, tid, av);
        } else {
            if (ep.act == PK_ACT_RELU) epilogue_pass<T, PK_ACT_RELU, 2>(cs, C, ep, mh, n0, M, N, tid, av);
            else epilogue_pass<T, PK_ACT_NONE, 2>(cs, C, ep, mh, n0, M, N, tid, av);
        }
        if (p == 0) __syncthreads();
    }
}

template <typename T, bool B_COL, bool TAIL, bool BITS = false>
__global__ __launch_bounds__(512, 2) void gemm8p_hm2_kernel(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                                           long long M, long long N, long long K, long long lda,
                                                           long long ldb, unsigned a_bytes, unsigned b_bytes, EpiParams ep) {
    gemm8p_hm2_tile<T, B_COL, TAIL, BITS>(A, B, C, M, N, K, lda, ldb, a_bytes, b_bytes, xcd_remap(blockIdx.x, gridDim.x), ep);
}

// ---- grouped weight gradients: up to PK_WGRAD_MAX (col,col) problems C_p = A_p^T B_p in ONE launch ----
// The weight-gradient GEMMs of a layer (q|k|v, out-proj, fc1, fc2, ...: outputs of 4..16 tiles, contraction over all
// B*T rows) each need split-K to fill 256 CUs on their own — 16..64 fp32 slabs per output and a reduction launch per
// GEMM.  Launched together they fill the chip with 4-5 slabs per output: the table below travels by value in the
// kernel arguments (scalar selects, no indexed access: an indexed by-value array would be copied to scratch).
struct GroupProb {
    const void* A; const void* B; void* C; float* ws; float* asum_ws; void* asum_out;
    long long M, N, K, lda, ldb, ldc;
    int kchunk, wg_begin, blk_begin, nslab;
    unsigned a_bytes, b_bytes;
    unsigned* pair_sync;  // pair mode (two K-slabs reduced inside the kernel): [tiles][2] ticket / flag words, else null
};
struct GroupArgs {
    GroupProb p[PK_WGRAD_MAX];
    int n;
    PairCtl pair;
};

__device__ __forceinline__ GroupProb group_select(const GroupArgs& g, int first_field_of, int pos) {
    // the LAST problem whose first workgroup (first_field_of = 0) / first reduction block (= 1) is <= pos
    GroupProb q = g.p[0];
#pragma unroll
    for (int i = 1; i < PK_WGRAD_MAX; ++i) {
        const int b = first_field_of ? g.p[i].blk_begin : g.p[i].wg_begin;
        if (i < g.n && pos >= b) q = g.p[i];
    }
    return q;
}

// Which workgroup works on what: the hardware deals workgroups to the eight XCDs round-robin (blockIdx % 8), and only the
// workgroups of ONE XCD share an L2.  Every (problem, K-slab) unit — the tiles that re-read the same rows of dY and X — is
// therefore kept on one XCD where it fits (plan_map: units of <= 32 tiles packed into bins of 32 = one round of an XCD's
// CUs), as runs of consecutive `lin` positions of a problem's slab-major walk; slot j = blockIdx / 8 of XCD x finds its
// run in the XCD's list and exits when the list is shorter (all workgroups of a launch run for the same time, so an XCD
// with fewer of them costs nothing).
constexpr int MAP_RUNS = 16;
struct GroupMap {
    unsigned w0[8][MAP_RUNS];  // first lin of the run | problem << 24
    unsigned w1[8][MAP_RUNS];  // first slot of the run in the XCD's list | workgroups << 16
};

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm8p_group_kernel(GroupArgs g, GroupMap mp, unsigned long long* stamps) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    int prob = -1, lin = 0;
#pragma unroll
    for (int b = 0; b < MAP_RUNS; ++b) {
        const unsigned w0 = mp.w0[x][b], w1 = mp.w1[x][b];
        const int beg = (int)(w1 & 0xffffu), cnt = (int)(w1 >> 16);
        if (j >= beg && j < beg + cnt) { prob = (int)(w0 >> 24); lin = (int)(w0 & 0xffffffu) + j - beg; }
    }
    if (prob < 0) return;
    GroupProb q = g.p[0];
#pragma unroll
    for (int i = 1; i < PK_WGRAD_MAX; ++i)
        if (prob == i) q = g.p[i];
    EpiParams ep;
    ep.bias = nullptr; ep.aux = nullptr; ep.preact = nullptr;
    ep.ldaux = 0; ep.ldc = q.ldc; ep.ldpre = 0;
    ep.act = PK_ACT_NONE; ep.mode = 0; ep.alpha = 1.f;
    gemm8p_tile<T, true, true, false, false>((const T*)q.A, (const T*)q.B, (T*)q.C, q.ws, q.asum_ws, (T*)q.asum_out, q.M,
                                             q.N, q.K, q.lda, q.ldb, q.kchunk, q.a_bytes, q.b_bytes, lin, stamps, ep,
                                             q.pair_sync, g.pair);
}

// C_p = sum over the K-slabs of problem p (fixed order: deterministic), 16-byte chunks; + the fused bias-gradient sums
template <typename T>
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(GroupArgs g, int total_blocks) {
    const GroupProb q = group_select(g, 1, (int)blockIdx.x);
    int next = total_blocks;
#pragma unroll
    for (int i = PK_WGRAD_MAX - 1; i >= 1; --i)
        if (i < g.n && g.p[i].blk_begin > q.blk_begin) next = g.p[i].blk_begin;
    const int nblk = next - q.blk_begin, blk = (int)blockIdx.x - q.blk_begin;
    if (q.nslab <= 1) return;  // (written by the GEMM itself; such a problem owns no blocks anyway)
    const long long M = q.M, N = q.N, slab = M * N;
    if (q.asum_ws) {
        for (long long m = (long long)blk * 256 + threadIdx.x; m < M; m += (long long)nblk * 256) {
            float t = 0.f;
            for (int z = 0; z < q.nslab; ++z) t += q.asum_ws[(long long)z * M + m];
            reinterpret_cast<T*>(q.asum_out)[m] = from_f32<T>(t);
        }
    }
    const long long nchunks_row = N / 8, total = M * nchunks_row;
    T* C = reinterpret_cast<T*>(q.C);
    for (long long c = (long long)blk * 256 + threadIdx.x; c < total; c += (long long)nblk * 256) {
        const long long gm = c / nchunks_row, gn = (c % nchunks_row) * 8;
        const float* p = q.ws + gm * N + gn;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll 4
        for (int z = 0; z < q.nslab; ++z) {
            const float4 a4 = *reinterpret_cast<const float4*>(p + (long long)z * slab);
            const float4 b4 = *reinterpret_cast<const float4*>(p + (long long)z * slab + 4);
            v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
            v[4] += b4.x; v[5] += b4.y; v[6] += b4.z; v[7] += b4.w;
        }
        store16_nt<T>(C + gm * q.ldc + gn, vec16_pack<T>(v));
    }
}

}  // namespace

namespace {
// extent of each operand in bytes (last row: only its valid part; col form with padded rows: the pad is readable)
void operand_bytes(long long M, long long N, long long K, long long lda, long long ldb, int a_col, int b_col,
                   long long* a_bytes, long long* b_bytes) {
    const long long a_rows = a_col ? K : M, a_cols = a_col ? std::min(lda, (M + 7) & ~7LL) : K;
    const long long b_rows = b_col ? K : N, b_cols = b_col ? N : K;
    *a_bytes = ((a_rows - 1) * lda + a_cols) * 2;
    *b_bytes = ((b_rows - 1) * ldb + b_cols) * 2;
}
}  // namespace

// What this kernel takes on top of the dispatcher's conditions (gemm.hip): K a multiple of 8 (partial last K-tile) and
// operands below 2 GiB (32-bit per-lane offsets + K offset of the buffer loads).
extern "C" int pk_gemm8p_eligible(long long M, long long N, long long K, long long lda, long long ldb, int a_col,
                                  int b_col, int want_asum) {
    (void)want_asum;
    if (a_col && !b_col) return 0;  // (no caller on the path; its instantiation would need more than 256 registers)
    if (K < 8 || K % 8) return 0;
    long long a_bytes, b_bytes;
    operand_bytes(M, N, K, lda, ldb, a_col, b_col, &a_bytes, &b_bytes);
    const long long lim = 0x7FFFFFFFLL - 65536;  // (2 GiB: an offset of 2^31 must be out of range without wrapping)
    return a_bytes <= lim && b_bytes <= lim;
}

// The persistent 128 x 256-tile kernel (gemmpw.hip, round 6): row-form A, whole K-tiles, many rounds of tiles, lean epilogue.
extern "C" int pk_gemmpw_eligible(long long M, long long N, long long K, long long lda, long long ldb, long long ldc, int b_col,
                                  const EpiParams* ep);
extern "C" int pk_gemmpw_launch(const void* A, const void* B, void* C, long long M, long long N, long long K, long long lda,
                                long long ldb, unsigned a_bytes, unsigned b_bytes, int b_col, EpiParams ep, int dtype,
                                void* stream);
extern "C" int pk_gemm_use_pw(int on);
// 1 if pk_gemm8p_launch runs this call as the persistent walk of 256 x 256 tiles (gemm8p_tile's PW form, gemm8p_pt_kernel):
// row-form A, an even number (>= 4) of whole K-tiles, no split, a lean or mask-bit epilogue, at least `PK_GEMM_PT_MIN_TILES`
// (default 512: two rounds of the chip) tiles.  Bit 1 of pk_gemm_use_pw (default on; env PK_GEMM_PW sets the mask).
extern "C" int pk_gemm8p_is_pt(long long M, long long N, long long K, long long lda, long long ldb, int a_col, int b_col,
                               int splitk, int has_ws, int has_asum, const EpiParams* ep) {
    static const int min_tiles = [] { const char* e = getenv("PK_GEMM_PT_MIN_TILES"); return e ? atoi(e) : 512; }();
    if (!(pk_gemm_use_pw(-1) & 2)) return 0;
    if (a_col || splitk != 1 || has_ws || has_asum || ep->kb_rows > 0) return 0;
    if (ep->preact || ep->mode == 3 || (ep->act != PK_ACT_NONE && ep->act != PK_ACT_RELU)) return 0;
    if (ep->bits ? (ep->act != PK_ACT_RELU || (ep->mode != 0 && ep->mode != 2)) : ep->mode != 0) return 0;
    if (K % (2 * BK) || K / BK < 4) return 0;
    const long long t256 = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    return t256 >= min_tiles && t256 < (1LL << 24);
}
// 1 if pk_gemm8p_launch sends this call to it (the dispatcher asks for the timing sample's tag)
extern "C" int pk_gemm8p_is_pw(long long M, long long N, long long K, long long lda, long long ldb, int a_col, int b_col,
                               int splitk, int has_ws, int has_asum, const EpiParams* ep) {
    if (!(pk_gemm_use_pw(-1) & 1)) return 0;
    if (a_col || splitk != 1 || has_ws || has_asum || ep->kb_rows > 0 || ep->mode == 3) return 0;
    return pk_gemmpw_eligible(M, N, K, lda, ldb, ep->ldc, b_col, ep);
}

// Returns 1 if the GEMM was launched, 0 if it is not eligible, or a hip error code.
extern "C" int pk_gemm8p_launch(const void* A, const void* B, void* C, float* ws, float* asum_ws, void* asum_out,
                                long long M, long long N, long long K, long long lda, long long ldb, int a_col,
                                int b_col, int kchunk, int splitk, EpiParams ep, int dtype, void* stream) {
    if (!pk_gemm8p_eligible(M, N, K, lda, ldb, a_col, b_col, asum_ws || asum_out)) return 0;
    long long a_bytes, b_bytes;
    operand_bytes(M, N, K, lda, ldb, a_col, b_col, &a_bytes, &b_bytes);
    if (ep.kb_rows > 0 && b_col) {  // K was rounded up to 8 for a zero-padded row-form A: B ends where it really ends
        long long unused;
        operand_bytes(M, N, ep.kb_rows, lda, ldb, a_col, b_col, &unused, &b_bytes);
    }
    if (pk_gemm8p_is_pw(M, N, K, lda, ldb, a_col, b_col, splitk, ws != nullptr, asum_ws || asum_out, &ep))
        return pk_gemmpw_launch(A, B, C, M, N, K, lda, ldb, (unsigned)a_bytes, (unsigned)b_bytes, b_col, ep, dtype, stream);
    if (pk_gemm8p_is_pt(M, N, K, lda, ldb, a_col, b_col, splitk, ws != nullptr, asum_ws || asum_out, &ep)) {
        const int total = (int)(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
        static const int wgs = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return std::max(8, cus / 8 * 8);
        }();
        dim3 grid((unsigned)std::min(wgs, total / 8 * 8)), block(512);
        hipStream_t s = (hipStream_t)stream;
        unsigned long long* stamps = nullptr;
#define PK_PT(TT, BC, BT) hipLaunchKernelGGL((gemm8p_pt_kernel<TT, BC, BT>), grid, block, 0, s, (const TT*)A, (const TT*)B, (TT*)C, M, N, K, \
                                             lda, ldb, (unsigned)a_bytes, (unsigned)b_bytes, total, stamps, ep)
        if (dtype == PK_F16) {
            if (ep.bits) { if (b_col) PK_PT(f16, true, true); else PK_PT(f16, false, true); }
            else { if (b_col) PK_PT(f16, true, false); else PK_PT(f16, false, false); }
        } else {
            if (ep.bits) { if (b_col) PK_PT(bf16, true, true); else PK_PT(bf16, false, true); }
            else { if (b_col) PK_PT(bf16, true, false); else PK_PT(bf16, false, false); }
        }
#undef PK_PT
        PK_LAUNCH_CHECK();
        return 1;
    }
    const bool hm = ep.half_m && !a_col && !(ep.preact || ep.mode == 3 || (ep.act != PK_ACT_NONE && ep.act != PK_ACT_RELU));
    const int tm = hm ? BM / 2 : BM;
    const int total = (int)(((M + tm - 1) / tm) * ((N + BN - 1) / BN) * splitk);
    unsigned long long* stamps = nullptr;  // PK8P_STAMP_PTR: device buffer of the diagnostic build's time stamps
#if defined(PK8P_STAMPS) || defined(PKBS_STAMPS)  /* (diagnostic builds only: the shipped library never reads the variable) */
    static unsigned long long* const stamp_buf = [] { const char* e = getenv("PK8P_STAMP_PTR"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }();
    stamps = stamp_buf;
#endif
    dim3 grid((unsigned)total), block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool any = ep.preact || ep.mode == 3 || (ep.act != PK_ACT_NONE && ep.act != PK_ACT_RELU);
    const bool tail = (K % BK) != 0;
#define PK_K(TT, AC, BC, ANYV, TAILV)                                                                                  \
    hipLaunchKernelGGL((gemm8p_kernel<TT, AC, BC, ANYV, TAILV>), grid, block, 0, s, (const TT*)A, (const TT*)B, (TT*)C, \
                       ws, asum_ws, (TT*)asum_out, M, N, K, lda, ldb, kchunk, (unsigned)a_bytes, (unsigned)b_bytes,    \
                       total, stamps, ep)
#define PK_D(TT, AC, BC)                                                       \
    do {                                                                       \
        if (any && tail) PK_K(TT, AC, BC, true, true);                         \
        else if (any) PK_K(TT, AC, BC, true, false);                           \
        else if (tail) PK_K(TT, AC, BC, false, true);                          \
        else PK_K(TT, AC, BC, false, false);                                   \
    } while (0)
#define PK_L(AC, BC)                                   \
    do {                                               \
        if (dtype == PK_F16) PK_D(f16, AC, BC);        \
        else PK_D(bf16, AC, BC);                       \
    } while (0)
    // the half-M tile: lean epilogues, row-form A (what pk_gemm sends it: see launch_gemm in gemm.hip)
#define PK_H(TT, BC)                                                                                                     \
    do {                                                                                                                 \
        if (tail)                                                                                                        \
            hipLaunchKernelGGL((gemm8p_kernel<TT, false, BC, false, true, true>), grid, block, 0, s, (const TT*)A, (const TT*)B, \
                               (TT*)C, ws, asum_ws, (TT*)asum_out, M, N, K, lda, ldb, kchunk, (unsigned)a_bytes,         \
                               (unsigned)b_bytes, total, stamps, ep);                                                    \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm8p_kernel<TT, false, BC, false, false, true>), grid, block, 0, s, (const TT*)A, (const TT*)B, \
                               (TT*)C, ws, asum_ws, (TT*)asum_out, M, N, K, lda, ldb, kchunk, (unsigned)a_bytes,         \
                               (unsigned)b_bytes, total, stamps, ep);                                                    \
    } while (0)
    static const bool hm2_on = [] { const char* e = getenv("PK_GEMM_HM2"); return !e || atoi(e) != 0; }();  // (A/B: 0 = the switched-off form)
    if (ep.bits) {  // the ReLU mask as bits (pk_gemm_relu_bits): instantiations of their own, lean epilogue, whole K-tiles, no split
        if (a_col || any || tail || splitk != 1 || ws) return 0;
#define PK_B8(TT, BC) hipLaunchKernelGGL((gemm8p_kernel<TT, false, BC, false, false, false, true>), grid, block, 0, s, (const TT*)A, \
                                         (const TT*)B, (TT*)C, ws, asum_ws, (TT*)asum_out, M, N, K, lda, ldb, kchunk,             \
                                         (unsigned)a_bytes, (unsigned)b_bytes, total, stamps, ep)
#define PK_BH(TT, BC) hipLaunchKernelGGL((gemm8p_hm2_kernel<TT, BC, false, true>), grid, block, 0, s, (const TT*)A, (const TT*)B,  \
                                         (TT*)C, M, N, K, lda, ldb, (unsigned)a_bytes, (unsigned)b_bytes, ep)
        if (hm) {
            if (dtype == PK_F16) { if (b_col) PK_BH(f16, true); else PK_BH(f16, false); }
            else { if (b_col) PK_BH(bf16, true); else PK_BH(bf16, false); }
        } else {
            if (dtype == PK_F16) { if (b_col) PK_B8(f16, true); else PK_B8(f16, false); }
            else { if (b_col) PK_B8(bf16, true); else PK_B8(bf16, false); }
        }
#undef PK_B8
#undef PK_BH
        PK_LAUNCH_CHECK();
        return 1;
    }
    if (hm && hm2_on && splitk == 1 && !ws) {  // the 128 x 256 tile with its own two-phase schedule
#define PK_H2(TT, BC, TL)                                                                                               \
    hipLaunchKernelGGL((gemm8p_hm2_kernel<TT, BC, TL>), grid, block, 0, s, (const TT*)A, (const TT*)B, (TT*)C, M, N, K, lda, \
                       ldb, (unsigned)a_bytes, (unsigned)b_bytes, ep)
        if (dtype == PK_F16) {
            if (b_col) { if (tail) PK_H2(f16, true, true); else PK_H2(f16, true, false); }
            else { if (tail) PK_H2(f16, false, true); else PK_H2(f16, false, false); }
        } else {
            if (b_col) { if (tail) PK_H2(bf16, true, true); else PK_H2(bf16, true, false); }
            else { if (tail) PK_H2(bf16, false, true); else PK_H2(bf16, false, false); }
        }
#undef PK_H2
    } else if (hm) {
        if (dtype == PK_F16) { if (b_col) PK_H(f16, true); else PK_H(f16, false); }
        else { if (b_col) PK_H(bf16, true); else PK_H(bf16, false); }
    } else if (!a_col && !b_col) PK_L(false, false);
    else if (!a_col && b_col) PK_L(false, true);
    else PK_L(true, true);
#undef PK_H
#undef PK_L
#undef PK_D
#undef PK_K
    PK_LAUNCH_CHECK();
    return 1;
}

// ---- grouped weight gradients (host side) ----
namespace {
struct GroupPlan {
    int nslab[PK_WGRAD_MAX], kchunk[PK_WGRAD_MAX], wg_begin[PK_WGRAD_MAX], blk_begin[PK_WGRAD_MAX];
    size_t ws_off[PK_WGRAD_MAX], asum_off[PK_WGRAD_MAX];  // in floats
    int total_wgs, total_blks;
    size_t ws_floats;
    // pair mode: problems with exactly two K-slabs are reduced inside the kernel (gemm8p_tile, role 1 / 2) and own no blocks
    // of the reduction launch; their ticket / flag words sit behind the slabs ([sync_off, sync_off + sync_words) floats)
    bool pair[PK_WGRAD_MAX];
    size_t sync_off[PK_WGRAD_MAX], sync_begin, sync_words;
};

inline long long tiles256(const PkWgradProblem& q) { return ((q.M + BM - 1) / BM) * ((q.N + BN - 1) / BN); }

// One K-chunk length L for every problem of the group (workgroups of equal duration), problem p cut into
// ceil(K_p / L) slabs.  L minimises  rounds x (time of a workgroup) + the slab traffic:  rounds = ceil(workgroups / 256),
// a workgroup = L / 64 K-tiles of ~1.45 us + ~6.5 us of prologue / epilogue (tools/gemm_phase_stamps.py), 512 KiB of
// fp32 slab written and read back per split workgroup at ~5 TB/s.
// Round 4: TWO lengths where one leaves the last round of the chip half empty — the problems with few tiles take L / 2 (twice
// the slabs): NLLB-1.3B's encoder layer at 8192 rows is 320 tiles; two slabs each = 640 workgroups = 2.5 rounds of the long
// kind (3 in practice), while fc1 / fc2 at two slabs (512 workgroups, two full rounds) + q|k|v and out-proj at four (256
// workgroups of half the duration) is 2.5 rounds' worth of time.  The mixed plan is taken only when list scheduling (longest
// first, 256 CUs) says it beats the best single length by 5 %; plans are cached by the group's shapes.
inline double wg_us(long long L) { return (double)L / BK * 1.45 + 6.5; }
double list_makespan(double dur_long, long long n_long, double dur_short, long long n_short) {
    std::priority_queue<double, std::vector<double>, std::greater<double>> h;
    for (int i = 0; i < 256; ++i) h.push(0.0);
    double last = 0.0;
    for (long long i = 0; i < n_long + n_short; ++i) {
        const double t = h.top() + (i < n_long ? dur_long : dur_short);
        h.pop();
        h.push(t);
        last = std::max(last, t);
    }
    return last;
}
// pair mode on / off: PK_WGRAD_PAIR=0 in the environment, or pk_gemm_wgrad_pair(0) at run time (how the tests obtain the
// reduction launch's result as the reference of the in-kernel one, in one process).  2 = on + the diagnostic that drops the
// first workgroup's publish and shortens the second's wait (tests of the error path; never set in production).
int g_pair_on = [] { const char* e = getenv("PK_WGRAD_PAIR"); return (!e || atoi(e) != 0) ? 1 : 0; }();
// The ticket / flag words of pair mode: ONE BUFFER PER (device, stream), owned by the library, zeroed when it is allocated.
// Tickets of one buffer only mean something in launch order, and launches are ordered within a stream only: two grouped
// launches running at once on two streams would interleave their tickets in a shared buffer and mis-pair.
// A stream beyond the table's capacity gets no buffer and its launches take the reduction launch instead.
constexpr size_t PAIR_SYNC_WORDS = 64 * 1024;
constexpr int PAIR_SYNC_STREAMS = 32;
struct PairSyncEntry { int dev; hipStream_t stream; unsigned* words; };
std::mutex g_pair_mu;
PairSyncEntry g_pair_bufs[PAIR_SYNC_STREAMS];
int g_pair_nbufs = 0;
unsigned* g_pair_err_host = nullptr;  // sticky error word: pinned host memory the kernels write (PairCtl::err)
unsigned* g_pair_err_dev = nullptr;

unsigned* pair_sync_buffer(hipStream_t s, bool allocate) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_pair_mu);
    for (int i = 0; i < g_pair_nbufs; ++i)
        if (g_pair_bufs[i].dev == dev && g_pair_bufs[i].stream == s) return g_pair_bufs[i].words;
    if (!allocate || g_pair_nbufs >= PAIR_SYNC_STREAMS) return nullptr;
    if (!g_pair_err_host) {
        unsigned* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *h = 0u;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
        g_pair_err_host = h;
        g_pair_err_dev = (unsigned*)d;
    }
    unsigned* p = nullptr;
    if (hipMalloc((void**)&p, PAIR_SYNC_WORDS * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(p, 0, PAIR_SYNC_WORDS * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
    g_pair_bufs[g_pair_nbufs++] = PairSyncEntry{dev, s, p};
    return p;
}

// A kernel of an EARLIER grouped launch reported a lost hand-off (its tile holds NaN): returns the word and clears it; every
// ticket buffer is re-zeroed behind a device synchronisation (an odd ticket left behind would mis-pair all later launches).
unsigned pair_error_take() {
    if (!g_pair_err_host) return 0u;
    const unsigned e = *(volatile unsigned*)g_pair_err_host;
    if (!e) return 0u;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> g(g_pair_mu);
    for (int i = 0; i < g_pair_nbufs; ++i) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (g_pair_bufs[i].dev != dev) (void)hipSetDevice(g_pair_bufs[i].dev);
        (void)hipMemset(g_pair_bufs[i].words, 0, PAIR_SYNC_WORDS * sizeof(unsigned));
        if (g_pair_bufs[i].dev != dev) (void)hipSetDevice(dev);
    }
    *(volatile unsigned*)g_pair_err_host = 0u;
    return e;
}

void plan_group_uncached(const PkWgradProblem* p, int n, GroupPlan* pl, bool pair_on) {
    long long kmax = 0;
    for (int i = 0; i < n; ++i) kmax = std::max(kmax, p[i].K);
    double best = 1e30;
    long long bestL = (kmax + BK - 1) / BK * BK, prevL = -1;
    for (int s = 1; s <= 64; ++s) {
        const long long L = ((kmax + s - 1) / s + BK - 1) / BK * BK;
        if (L == prevL) continue;
        prevL = L;
        long long wgs = 0, split_wgs = 0;
        for (int i = 0; i < n; ++i) {
            const long long sp = (p[i].K + L - 1) / L;
            wgs += tiles256(p[i]) * sp;
            if (sp > 1) split_wgs += tiles256(p[i]) * sp;
        }
        const double rounds = (double)((wgs + 255) / 256);
        const double cost = rounds * wg_us(L) + (double)split_wgs * 0.105 + (split_wgs ? 4.0 : 0.0);
        if (cost < best) { best = cost; bestL = L; }
        if (L <= 512) break;
    }
    // the mixed plans: problems of at most `cut` tiles at a FRACTION of the length (PK_WGRAD_MIXED=0: off, A/B).  Round 4 knew
    // L / 2 only; round 6 adds L / 3 and L / 4 (PK_WGRAD_MIXED_DIV=2: as before): NLLB-1.3B's ENCODER layer at 8192 rows is
    // fc1 / fc2 unsplit (256 workgroups of 128 K-tiles = one round of the chip) + q|k|v and out-proj (64 tiles) — at L / 2
    // those are 128 workgroups of 64 K-tiles on half the CUs (makespan 192 K-tiles where the work is 160 per CU), at L / 4
    // 256 workgroups of 32: every CU runs one long and one short workgroup.
    static const bool mixed_on = [] { const char* e = getenv("PK_WGRAD_MIXED"); return !e || atoi(e) != 0; }();
    static const int mixed_div = [] { const char* e = getenv("PK_WGRAD_MIXED_DIV"); return e ? std::max(2, std::min(4, atoi(e))) : 4; }();
    long long mixL = 0, mixLs = 0, mixcut = -1;
    double mixbest = 0.95 * best;
    for (int div = 2; mixed_on && div <= mixed_div; ++div) {
        prevL = -1;
        for (int s = 1; s <= 16; ++s) {
            const long long L = ((kmax + s - 1) / s + BK - 1) / BK * BK, Ls = (L / div + BK - 1) / BK * BK;
            if (L == prevL || Ls < 512) continue;
            prevL = L;
            for (int c = 0; c < n; ++c) {
                const long long cut = tiles256(p[c]);
                long long nl = 0, ns = 0, split_wgs = 0;
                bool any_big = false;
                for (int i = 0; i < n; ++i) {
                    const bool small = tiles256(p[i]) <= cut;
                    const long long sp = (p[i].K + (small ? Ls : L) - 1) / (small ? Ls : L);
                    (small ? ns : nl) += tiles256(p[i]) * sp;
                    if (sp > 1) split_wgs += tiles256(p[i]) * sp;
                    any_big |= !small;
                }
                if (!any_big || nl + ns > 4096) continue;
                const double cost = list_makespan(wg_us(L), nl, wg_us(Ls), ns) + (double)split_wgs * 0.105 + (split_wgs ? 4.0 : 0.0);
                if (cost < mixbest) { mixbest = cost; mixL = L; mixLs = Ls; mixcut = cut; }
            }
        }
    }
    int wg = 0, blk = 0;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const long long Li = mixcut < 0 ? bestL : (tiles256(p[i]) <= mixcut ? mixLs : mixL);
        const int sp = (int)((p[i].K + Li - 1) / Li);
        pl->nslab[i] = sp;
        pl->kchunk[i] = (int)Li;
        pl->wg_begin[i] = wg;
        wg += (int)tiles256(p[i]) * sp;
        pl->blk_begin[i] = blk;
        pl->ws_off[i] = pl->asum_off[i] = 0;
        // (PK_WGRAD_PAIR=0: every split problem through the reduction launch, as before round 5 — the A/B and the tests'
        // reference; the slab region of a pair must be addressable through one buffer descriptor: < 2 GiB)
        pl->pair[i] = pair_on && sp == 2 && 2 * p[i].M * p[i].N * 4 < (1LL << 31);
        if (sp > 1) {
            const long long chunks = p[i].M * (p[i].N / 8);
            if (!pl->pair[i]) blk += (int)std::min(2048LL, (chunks + 255) / 256);
            pl->ws_off[i] = off;
            off += ((size_t)sp * p[i].M * p[i].N + 3) & ~(size_t)3;
            if (p[i].asum_out) {
                pl->asum_off[i] = off;
                off += ((size_t)sp * p[i].M + 3) & ~(size_t)3;
            }
        }
    }
    pl->total_wgs = wg;
    pl->total_blks = blk;
    pl->ws_floats = off;
    // ticket / flag words of the pairs: offsets into the library's sync buffer (pair_sync_buffer); a group that would not
    // fit keeps its last problems on the reduction launch
    size_t words = 0;
    pl->sync_begin = 0;
    for (int i = 0; i < n; ++i) {
        pl->sync_off[i] = words;
        if (!pl->pair[i]) continue;
        const size_t need = ((size_t)2 * tiles256(p[i]) + 63) & ~(size_t)63;  // (256-byte lines of its own per problem)
        if (words + need > PAIR_SYNC_WORDS) {
            pl->pair[i] = false;
            pl->blk_begin[i] = -1;  // (marks: needs its blocks after all — fixed up below)
            continue;
        }
        words += need;
    }
    pl->sync_words = words;
    bool redo = false;
    for (int i = 0; i < n; ++i) redo |= pl->blk_begin[i] < 0;
    if (redo) {  // re-deal the reduction blocks now that some problems lost their pair status
        int b = 0;
        for (int i = 0; i < n; ++i) {
            pl->blk_begin[i] = b;
            if (pl->nslab[i] > 1 && !pl->pair[i]) b += (int)std::min(2048LL, (p[i].M * (p[i].N / 8) + 255) / 256);
        }
        pl->total_blks = b;
    }
}
// (the plan depends on the shapes and on which problems carry a bias sum: cached — four calls per launch ask for it)
void plan_group(const PkWgradProblem* p, int n, GroupPlan* pl, bool pair_on = g_pair_on != 0) {
    struct Key { long long v[PK_WGRAD_MAX][4]; int n, pair_on; };
    static std::mutex mu;
    static std::vector<std::pair<Key, GroupPlan>> cache;
    Key k;
    memset(&k, 0, sizeof k);
    k.n = n;
    k.pair_on = pair_on ? 1 : 0;
    for (int i = 0; i < n; ++i) { k.v[i][0] = p[i].M; k.v[i][1] = p[i].N; k.v[i][2] = p[i].K; k.v[i][3] = p[i].asum_out != nullptr; }
    {
        std::lock_guard<std::mutex> g(mu);
        for (const auto& e : cache)
            if (memcmp(&e.first, &k, sizeof k) == 0) { *pl = e.second; return; }
    }
    plan_group_uncached(p, n, pl, pair_on);
    std::lock_guard<std::mutex> g(mu);
    if (cache.size() >= 64) cache.erase(cache.begin());
    cache.emplace_back(k, *pl);
}

// The launch map (GroupMap above).  Units = (problem, K-slab); a unit of more than 32 tiles is cut into runs of 32
// consecutive tiles of its walk (4 x 8 or 8 x 4 tile blocks: gemm8p_tile's GROUP_M order).  First-fit-decreasing into
// bins of 32 workgroups, bins dealt to the least loaded XCD.  Falls back to the contiguous split of the slab-major walk
// (what xcd_remap gives the single-problem kernel) when the packing would need more rounds of the chip, or more runs per
// XCD than the map holds.  PK_WGRAD_MAP=0 (diagnostic, read once) forces the fallback.
struct MapRun { int prob, lin, cnt; };
int plan_map(const PkWgradProblem* p, int n, const GroupPlan& pl, GroupMap* mp) {
    static const int packed = [] { const char* e = getenv("PK_WGRAD_MAP"); return (!e || atoi(e) != 0) ? 1 : 0; }();
    std::vector<MapRun> xr[8];
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = packed != 0;
    if (ok) {
        std::vector<MapRun> items;
        for (int i = 0; i < n; ++i) {
            const int tiles = (int)tiles256(p[i]);
            for (int s = 0; s < pl.nslab[i]; ++s)
                for (int t = 0; t < tiles; t += 32) items.push_back(MapRun{i, s * tiles + t, std::min(32, tiles - t)});
        }
        // operand panels (256 rows of dY or X per K-tile) a run reads: what its XCD fetches for it per K-step
        auto panels = [&](const MapRun& r) {
            const int nt_m = (int)((p[r.prob].M + BM - 1) / BM), nt_n = (int)((p[r.prob].N + BN - 1) / BN);
            unsigned long long ms[64] = {0}, ns[64] = {0};
            for (int k = 0; k < r.cnt; ++k) {
                int t = (r.lin + k) % (nt_m * nt_n);
                const int GROUP_M = nt_n <= 2 ? 8 : 4, group_size = GROUP_M * nt_n, first_m = (t / group_size) * GROUP_M;
                const int gsz = std::min(nt_m - first_m, GROUP_M);
                const int tm = first_m + (t % group_size) % gsz, tn = (t % group_size) / gsz;
                ms[(tm >> 6) & 63] |= 1ull << (tm & 63);
                ns[(tn >> 6) & 63] |= 1ull << (tn & 63);
            }
            int c = 0;
            for (int k = 0; k < 64; ++k) c += __builtin_popcountll(ms[k]) + __builtin_popcountll(ns[k]);
            return c;
        };
        // Largest runs first, each into the bin (of 8 x rounds bins of 32 workgroups) that has room and the LEAST panel
        // load so far: small, poorly sharing units (a d x d problem: 4 tiles, 4 panels) end up spread over the XCDs beside
        // the large ones instead of eight of them in one bin — an XCD's pace is set by what it fetches per K-step.
        // (a mixed plan: the long workgroups first, so that every XCD runs its share of them before the short ones)
        std::stable_sort(items.begin(), items.end(), [&](const MapRun& a, const MapRun& b) {
            return pl.kchunk[a.prob] != pl.kchunk[b.prob] ? pl.kchunk[a.prob] > pl.kchunk[b.prob] : a.cnt > b.cnt;
        });
        const int nbins = 8 * ((pl.total_wgs + 255) / 256);
        std::vector<int> fill(nbins, 0), pload(nbins, 0);
        std::vector<std::vector<MapRun>> bins(nbins);
        for (const MapRun& it : items) {
            const int pn = panels(it);
            int best = -1;
            for (int b = 0; b < nbins; ++b)
                if (fill[b] + it.cnt <= 32 && (best < 0 || pload[b] < pload[best] || (pload[b] == pload[best] && fill[b] < fill[best])))
                    best = b;
            if (best < 0) { ok = false; break; }
            fill[best] += it.cnt;
            pload[best] += pn;
            bins[best].push_back(it);
        }
        for (int b = 0; ok && b < nbins; ++b) {  // bin b -> XCD b % 8, round b / 8
            load[b & 7] += fill[b];
            for (const MapRun& it : bins[b]) xr[b & 7].push_back(it);
        }
        int rounds = 0;
        for (int x = 0; x < 8; ++x) {
            rounds = std::max(rounds, (load[x] + 31) / 32);
            if ((int)xr[x].size() > MAP_RUNS) ok = false;
        }
        if (rounds > (pl.total_wgs + 255) / 256) ok = false;
    }
    if (!ok) {  // contiguous ranges of the problem-major, slab-major walk
        const int tot = pl.total_wgs, q = tot >> 3, r = tot & 7;
        for (int x = 0; x < 8; ++x) {
            xr[x].clear();
            int beg = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
            const int end = beg + (x < r ? q + 1 : q);
            load[x] = end - beg;
            for (int i = 0; i < n && beg < end; ++i) {
                const int pe = pl.wg_begin[i] + (int)tiles256(p[i]) * pl.nslab[i];
                if (beg >= pe) continue;
                const int e = std::min(end, pe);
                xr[x].push_back(MapRun{i, beg - pl.wg_begin[i], e - beg});
                beg = e;
            }
        }
    }
    int most = 0;
    for (int x = 0; x < 8; ++x) {
        int slot = 0, b = 0;
        for (const MapRun& it : xr[x]) {
            mp->w0[x][b] = (unsigned)it.lin | ((unsigned)it.prob << 24);
            mp->w1[x][b] = (unsigned)slot | ((unsigned)it.cnt << 16);
            slot += it.cnt;
            ++b;
        }
        for (; b < MAP_RUNS; ++b) { mp->w0[x][b] = 0; mp->w1[x][b] = 0; }
        most = std::max(most, slot);
    }
    return 8 * most;  // grid size
}
}  // namespace

extern "C" int pk_gemm_wgrad_pair(int on) {
    const int prev = g_pair_on;
    if (on >= 0) g_pair_on = on == 2 ? 2 : (on ? 1 : 0);
    return prev;
}

// 1 if the problem can ride in a grouped launch: what the dispatcher asks of a 256-tile (col,col) GEMM (gemm.hip)
extern "C" int pk_gemm8p_group_eligible(const PkWgradProblem* q) {
    auto al = [](const void* p, long long ld) { return ((uintptr_t)p % 16) == 0 && (ld % 8) == 0; };
    if (!q->A || !q->B || !q->C) return 0;
    // (round 5: outputs down to 64 rows / columns ride too — an adapter's d x 64 and 64 x d weight gradients fill a quarter of
    // their 256-tiles, but what they cost is the one pass over the 16 000 x 1024 operand, which the group's K-slabs spread over
    // the chip; one by one they ran as 128-tile split-K GEMMs + a reduction launch each: 21 + 23 + 2 x 6 us -> one launch)
    if (q->M < 64 || q->N < 64 || (q->M < 256 && q->N < 256) || q->K < 64 || q->N % 8) return 0;
    if (!al(q->A, q->lda) || !al(q->B, q->ldb) || !al(q->C, q->ldc)) return 0;
    if (q->M % 8 && q->lda < ((q->M + 7) & ~7LL)) return 0;
    if (tiles256(*q) > 4096) return 0;
    return pk_gemm8p_eligible(q->M, q->N, q->K, q->lda, q->ldb, 1, 1, q->asum_out != nullptr);
}

extern "C" int pk_gemm8p_group_plan(const PkWgradProblem* p, int n, size_t* ws_bytes, int* workgroups, int* slabs) {
    if (n < 1 || n > PK_WGRAD_MAX) return -1;
    GroupPlan pl;
    plan_group(p, n, &pl);
    if (ws_bytes) *ws_bytes = pl.ws_floats * sizeof(float);
    if (workgroups) *workgroups = pl.total_wgs;
    if (slabs) for (int i = 0; i < n; ++i) slabs[i] = pl.nslab[i];
    return 0;
}

// Host-only: the launch map as the kernel decodes it — out[2 b] = problem, out[2 b + 1] = position in that problem's
// slab-major walk for workgroup b (-1, -1: exits at once).  Returns the grid size (or -1); fills at most `cap` entries.
extern "C" int pk_gemm8p_group_map(const PkWgradProblem* p, int n, int* out, int cap) {
    if (n < 1 || n > PK_WGRAD_MAX) return -1;
    GroupPlan pl;
    plan_group(p, n, &pl);
    GroupMap mp;
    const int grid = plan_map(p, n, pl, &mp);
    for (int b = 0; b < grid && b < cap; ++b) {
        const int x = b & 7, j = b >> 3;
        int prob = -1, lin = -1;
        for (int r = 0; r < MAP_RUNS; ++r) {
            const int beg = (int)(mp.w1[x][r] & 0xffffu), cnt = (int)(mp.w1[x][r] >> 16);
            if (j >= beg && j < beg + cnt) { prob = (int)(mp.w0[x][r] >> 24); lin = (int)(mp.w0[x][r] & 0xffffffu) + j - beg; }
        }
        out[2 * b] = prob; out[2 * b + 1] = lin;
    }
    return grid;
}

// Returns 1 if launched, a hip error code otherwise.  The caller (gemm.hip: pk_gemm_wgrad_group) has checked eligibility
// and the workspace size.
extern "C" int pk_gemm8p_group_launch(const PkWgradProblem* p, int n, int dtype, float* workspace, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (const unsigned e = pair_error_take()) {
        pk_set_error("pk_gemm_wgrad_group: an earlier grouped launch lost a hand-off of its in-kernel two-slab reduction "
                     "(tile %u of a problem: its weight gradient was poisoned with NaN); the ticket words were reset",
                     e & 0x7fffffffu);
        return (int)hipErrorUnknown;
    }
    GroupPlan pl;
    plan_group(p, n, &pl);
    unsigned* sync_buf = pl.sync_words ? pair_sync_buffer(s, true) : nullptr;
    if (pl.sync_words && !sync_buf) plan_group(p, n, &pl, false);  // no ticket words for this stream: the reduction launch
    GroupArgs g;
    g = GroupArgs{};
    g.n = n;
    g.pair = PairCtl{g_pair_err_dev, g_pair_on == 2 ? (1u << 10) : (1u << 22), g_pair_on == 2 ? 1 : 0};
    for (int i = 0; i < n; ++i) {
        GroupProb& q = g.p[i];
        long long a_bytes, b_bytes;
        operand_bytes(p[i].M, p[i].N, p[i].K, p[i].lda, p[i].ldb, 1, 1, &a_bytes, &b_bytes);
        q.A = p[i].A; q.B = p[i].B; q.C = p[i].C; q.asum_out = p[i].asum_out;
        q.M = p[i].M; q.N = p[i].N; q.K = p[i].K; q.lda = p[i].lda; q.ldb = p[i].ldb; q.ldc = p[i].ldc;
        q.kchunk = pl.kchunk[i]; q.wg_begin = pl.wg_begin[i]; q.blk_begin = pl.blk_begin[i]; q.nslab = pl.nslab[i];
        q.a_bytes = (unsigned)a_bytes; q.b_bytes = (unsigned)b_bytes;
        q.ws = pl.nslab[i] > 1 ? workspace + pl.ws_off[i] : nullptr;
        q.asum_ws = (pl.nslab[i] > 1 && p[i].asum_out) ? workspace + pl.asum_off[i] : nullptr;
        q.pair_sync = pl.pair[i] ? sync_buf + pl.sync_off[i] : nullptr;
    }
    for (int i = n; i < PK_WGRAD_MAX; ++i) { g.p[i].wg_begin = 0x7fffffff; g.p[i].blk_begin = 0x7fffffff; }
    unsigned long long* stamps = nullptr;
#if defined(PK8P_STAMPS) || defined(PKBS_STAMPS)  /* (diagnostic builds only: the shipped library never reads the variable) */
    static unsigned long long* const stamp_buf = [] { const char* e = getenv("PK8P_STAMP_PTR"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }();
    stamps = stamp_buf;
#endif
    GroupMap mp;
    const int grid = plan_map(p, n, pl, &mp);
    if (dtype == PK_F16) hipLaunchKernelGGL((gemm8p_group_kernel<f16>), dim3(grid), dim3(512), 0, s, g, mp, stamps);
    else hipLaunchKernelGGL((gemm8p_group_kernel<bf16>), dim3(grid), dim3(512), 0, s, g, mp, stamps);
    PK_LAUNCH_CHECK();
    return 1;
}

extern "C" int pk_gemm8p_group_reduce(const PkWgradProblem* p, int n, int dtype, float* workspace, void* stream) {
    GroupPlan pl;
    plan_group(p, n, &pl);
    // (the plan the launch before this call used: without ticket words for this stream it ran every problem unpaired)
    if (pl.sync_words && !pair_sync_buffer((hipStream_t)stream, false)) plan_group(p, n, &pl, false);
    if (pl.total_blks == 0) return 1;
    GroupArgs g;
    g = GroupArgs{};
    g.n = n;
    for (int i = 0; i < n; ++i) {
        GroupProb& q = g.p[i];
        q.C = p[i].C; q.asum_out = p[i].asum_out; q.M = p[i].M; q.N = p[i].N; q.ldc = p[i].ldc;
        q.nslab = pl.pair[i] ? 1 : pl.nslab[i];  // (a pair is finished inside the GEMM kernel)
        // a problem without slabs owns no blocks: its first block is that of the next split problem (never selected:
        // group_select takes the LAST problem whose first block is <= the block index)
        q.blk_begin = pl.blk_begin[i];
        q.ws = pl.nslab[i] > 1 ? workspace + pl.ws_off[i] : nullptr;
        q.asum_ws = (pl.nslab[i] > 1 && p[i].asum_out) ? workspace + pl.asum_off[i] : nullptr;
    }
    for (int i = n; i < PK_WGRAD_MAX; ++i) { g.p[i].wg_begin = 0x7fffffff; g.p[i].blk_begin = 0x7fffffff; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PK_F16)
        hipLaunchKernelGGL((wgrad_group_reduce_kernel<f16>), dim3(pl.total_blks), dim3(256), 0, s, g, pl.total_blks);
    else
        hipLaunchKernelGGL((wgrad_group_reduce_kernel<bf16>), dim3(pl.total_blks), dim3(256), 0, s, g, pl.total_blks);
    PK_LAUNCH_CHECK();
    return 1;
}
