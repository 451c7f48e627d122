// Persistent 128 x 256-tile bf16 / fp16 MFMA GEMM for row-form A (forward and dX GEMMs: pasero/models/modules.py:92-96) whose
// output is MANY rounds of tiles over the chip and whose contraction is short (K = 1024: every projection of a d = 1024 model
// that reads the hidden state — NLLB-1.3B / transformer_big q|k|v, fc1 forward, the masked dH GEMM, the vocabulary logits).
//
// Why another kernel (round 6).  gemm8p.hip's tiles run [prologue 1.0-1.3 us][K loop][epilogue 4.5-5 us] one after the other,
// one workgroup per CU: at K = 1024 the K loop is 23 us of a 30 us tile, the matrix cores idle for a fifth of every round, and
// the same kernels reach 1.5 PFLOP/s at K = 8192 against 1.03-1.08 at K = 1024 (tools/gemm_in_model.py, C5 and the IWSLT
// recipe).  gemmbs.hip removes the epilogue phase for K <= 512 by keeping the B panel in registers; at K = 1024 that panel is
// the whole register file.  What does fit is the 128 x 256 tile of gemm8p_hm2_tile — 64 accumulator registers per lane — TWICE:
//   * one workgroup per CU walks its tiles (persistent: workgroup b takes the tiles xcd_remap(b + j * grid) of the same
//     XCD-contiguous, GROUP_M-panelled order the one-tile-per-workgroup kernels use);
//   * the K-tiles of consecutive tiles form ONE stream through the three-stage LDS ring of gemm8p_hm2_tile (same two phases
//     per K-tile, same DMA schedule, same hazards): the DMA that runs 1.5 K-tiles ahead simply continues into the next tile's
//     operands, so a tile has no prologue;
//   * at a tile's end the accumulators are copied to a second register set, and from there the tile leaves, one piece (16
//     rows x 128 columns of every wave) per K-tile in the first eight K-tiles of the next tile: bias / activation / rounding
//     in registers, two v_permlane16_swap so that a lane holds eight consecutive columns, ONE 16-byte buffer store per lane
//     (16 rows x 64 B per instruction, as gemmbs.hip) — no epilogue phase, no LDS staging, and the stores of the 256
//     workgroups are spread over the launch instead of arriving as one burst.
// Summation order per output element = that of gemm8p_hm2_tile / gemm8p_tile (K-tiles in order, k-steps 0, 1, one chain per
// accumulator): results are bit for bit pk_gemm's.
//
// Schedule of one K-tile t of the stream (stage = position % 3; gemm8p_hm2_tile's):
//   phase   fragment reads              MFMA quadrant   vector-memory operations, in issue order
//   A(t)    A0(t)                [8]    (0, 0)          [wait] DMA A0(t+2) ... store(s) of the piece that leaves (in the MFMA section)
//   B(t)    B1(t), B0(t+1)       [8]    (0, 1)          [wait] DMA B1(t+2), B0(t+3)
// RAW: the wait of A(t) needs B1(t), B0(t+1) (issued in B(t-2)), the wait of B(t) needs A0(t+1) (issued in A(t-1)).  vmcnt is
// ONE in-order queue and every K-tile puts NS store instructions into it behind its A0 request (NS = 1, or 2 with the mask
// bits; against dead offsets when no piece leaves, so that the count never varies): the wait of A(t) leaves 6 + NS operations
// in flight (A0(t+1), store(t-1), B1(t+1), B0(t+2)), the wait of B(t) 6 + 2 NS (store(t-1), B1(t+1), B0(t+2), A0(t+2),
// store(t)).  A store is first waited for two K-tiles after it issued (by the wait of A(t+2), through the B request behind
// it).  A first version waited for every store within ONE K-tile — side operands requested by loads sat behind it in the
// queue — and ran at the store's latency: 1.35 us per K-tile instead of 0.75.
// WAR: every slot is overwritten two phases after its last read, as in gemm8p_hm2_tile.
// Side operands of the piece that leaves (the tile's 256 bias values; the masked dH GEMM's 128 x 32 bytes of mask bits) do not
// travel through that queue: ONE LDS-DMA per tile brings them into LDS behind the ring (K-tile 7 of the tile they belong to:
// the previous tile's last piece has read its own by then), a piece reads its share with ds_read one K-tile ahead.
// Stage rotation: a tile of nk K-tiles advances the ring by nk % 3; the loop body names LOGICAL stages and the LDS bases of the
// three logical stages (scalars) are rotated at every tile boundary, so ONE unrolled body (three K-tiles) serves every nk.
// MEASURED (round 6, MI355X, random bf16, same box, us per call, persistent | tiled): 8192 x 8192 x 1024 159-183 | 123-132,
// 32768 x 4096 x 1024 282-288 | 226-228, 2048 x 256 208 x 1024 1311 | 1004: 20-30 % SLOWER, so pk_gemm does not use it unless
// asked (PK_GEMM_PW=1).  Why, by ablation builds: the stream WITHOUT the departing tile (no select, arithmetic, stores) runs
// 8192 x 8192 x 1024 in 112-122 us — only 4-10 % under the tiled kernel WITH its epilogues, because a K-tile of this tile
// takes 0.85-0.9 us against 1.47 us for the 256 x 256 tile's twice as many FLOPs — and the departing tile costs 0.28 us per
// K-tile = 4.5 us per 128 x 256 tile, twice what the LDS-staged burst epilogue costs per output element.  No single part of
// it is the cost (without the select +4 %, without the arithmetic +4 %, without the side reads +4 %, stores against dead
// offsets +7 %; in the MFMA sections instead of the load sections: -5 %): load and MFMA sections of the two wave groups are
// balanced against each other four times per K-tile, so EVERY instruction added to either lengthens the K-tile — the "unused
// issue slots beside the MFMAs" this design counted on do not exist in a loop that is bound by its own instruction stream
// and barriers, not by the matrix pipe.  What the experiment is worth keeping for: the tile stream (ring continued across
// tiles, rotating logical stages, DMA cursors) and the exact in-order-queue accounting, should a 256 x 256 persistent form
// with the burst epilogue be tried (its K loop is the efficient one; it would save the prologue, ~4 % at K = 1024).
// Past the last tile of a workgroup the same DMA instructions run against an EMPTY descriptor; rows past M read as zeros (the
// descriptor's range check covers the SGPR offset) and their stores are dropped the same way, column chunks past N by a
// per-lane dead offset.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "gemm_epi.h"
#include "gemm8p_common.h"

namespace {

constexpr int TM = 128, BN = 256, BK = 64;
constexpr int HALF = 16384;
constexpr int PSTAGE = 3 * HALF, PRING = 3 * PSTAGE;  // slots of a stage: A0 B0 B1; three stages = 144 KiB
constexpr int BIAS_OFF = PRING, MASK_OFF = PRING + 1024, PSMEM = MASK_OFF + 4096;  // + side operands of the departing tile
constexpr int S_A = 0, S_B0 = 1, S_B1 = 2;
constexpr unsigned DEAD_OFF = 0x80000000u;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// EPI: which results leave and how
constexpr int EPI_LEAN = 0;    // C = alpha v + bias                                                        (pk_gemm mode 0, no activation)
constexpr int EPI_BITSW = 1;   // C = relu(alpha v + bias) and the ReLU mask as one bit per element beside it (pk_gemm_relu_bits, mode 0)
constexpr int EPI_BITSR = 2;   // C = bit ? alpha v : 0                                                     (pk_gemm_relu_bits, mode 2)

struct PwArgs {
    const void* A; const void* B; void* C;
    long long M, N, K, lda, ldb, ldc;
    unsigned a_bytes, b_bytes, c_bytes, bits_bytes;
    int total;       // tiles
    float alpha;
    const void* bias;  // [N] or null (EPI_LEAN / EPI_BITSW)
    long long nstore;  // columns that may be stored (N, or N rounded up to 8 for padded rows)
    unsigned char* bits; long long ldbits;
    unsigned long long* stamps;
};

template <typename T, bool B_COL, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_pw_kernel(PwArgs g) {
    typedef typename M16<T>::vec V;
    typedef __attribute__((address_space(3))) void lds_void;
    __shared__ __attribute__((aligned(16))) char smem[PSMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(g.B);
    // (32-bit arithmetic throughout: every operand is below 2 GiB — the scalar registers are the scarce resource of this kernel)
    const int M = (int)g.M, N = (int)g.N;
    const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2), ldc2 = (unsigned)(g.ldc * 2), ldbits = (unsigned)g.ldbits;
    const int nk = (int)(g.K / BK);  // (whole K-tiles: the host sends nothing else here)
    const int nt_m = (int)((M + TM - 1) / TM), nt_n = (int)((N + BN - 1) / BN);
    const int GROUP_M = nt_n <= 2 ? 8 : 4;
    const int group_size = GROUP_M * nt_n;
    const int total = g.total, nwg = (int)gridDim.x;

    // tile `vb` of this workgroup's walk -> byte offsets of its operand panels (SGPR operands of the DMA) and its origin
    struct TileRef { unsigned a_so, b_so; int m0, n0; bool live; };
    auto tile_ref = [&](int vb) -> TileRef {
        TileRef t;
        t.live = vb < total;
        const int lin = xcd_remap(t.live ? vb : 0, total);
        const int gid = lin / group_size, first_m = gid * GROUP_M;
        const int gsz = min(nt_m - first_m, GROUP_M);
        const int tile_m = first_m + (lin % group_size) % gsz, tile_n = (lin % group_size) / gsz;
        t.m0 = tile_m * TM; t.n0 = tile_n * BN;
        t.a_so = (unsigned)t.m0 * lda2;
        t.b_so = B_COL ? (unsigned)t.n0 * 2u : (unsigned)t.n0 * ldb2;
        return t;
    };

    // ---- operand streams: per-lane offsets of this wave's two DMA pieces per half-tile, relative to the tile's panel ----
    unsigned offa[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        {   // row form: 8 rows x 128 B per piece
            const int row = piece * 8 + (lane >> 3), chunk = (lane & 7) ^ HT<false>::swz(row);
            offa[i] = (unsigned)row * lda2 + (unsigned)(chunk * 16);
            if constexpr (!B_COL) offb[i] = (unsigned)row * ldb2 + (unsigned)(chunk * 16);
        }
        if constexpr (B_COL) {  // col form: 4 k-rows x 256 B per piece
            const int krow = piece * 4 + (lane >> 4), chunk = (lane & 15) ^ HT<true>::swz(krow);
            offb[i] = (unsigned)krow * ldb2 + (unsigned)(chunk * 16);
        }
    }
    const unsigned kstep_b = B_COL ? (unsigned)BK * ldb2 : (unsigned)(BK * 2);
    const unsigned half_b = B_COL ? 256u : 128u * ldb2;  // B1 = columns n0 + 128 ..

    // ---- the ring: LDS byte bases of the three LOGICAL stages (rotated at every tile boundary): scalars — the per-lane part of a
    // read address is ONE register per operand, the stage base is added where a load section starts (an SGPR operand) ----
    int sd[3] = {0, PSTAGE, 2 * PSTAGE};
    // row-form fragment of rows r0 + (l & 15), k-step kk: chunk (4 kk + (l >> 4)) ^ swizzle of the row — the swizzle
    // ((row >> 1) & 7) only sees l & 15 (r0 is a multiple of 16), so kk = 1 is kk = 0 with bit 6 flipped
    const int frag0 = (lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4);
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_char*)smem;
    unsigned ra0 = (unsigned)(wr * 64 * 128 + frag0);  // A0 image, k-step 0
    unsigned rb0[2];  // row form: [0] the B image, k-step 0; col form: n-tile j (LDS address)
    if constexpr (!B_COL) {
        rb0[0] = (unsigned)(wc * 32 * 128 + frag0);
        rb0[1] = 0u;
    } else {
        const int q4 = (lane & 15) >> 2, p = lane & 3, krow = 8 * (lane >> 4) + q4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wc * 32 + 16 * j + 4 * p;
            rb0[j] = lds0 + (unsigned)(HT<true>::offset(krow, col >> 3) + (col & 7) * 2);
        }
    }
    asm volatile("" : "+v"(ra0));  // opaque: kept in registers, not re-derived in every load section
    asm volatile("" : "+v"(rb0[0]));
    if constexpr (B_COL) asm volatile("" : "+v"(rb0[1]));

    TileRef cur = tile_ref((int)blockIdx.x), nxt = tile_ref((int)blockIdx.x + nwg);
    int vb_next = (int)blockIdx.x + 2 * nwg;
    // the stream's DMA cursors: SGPR offset and descriptor size of the NEXT request of each image (A0 and B1 of K-tile t + 2,
    // B0 of K-tile t + 3); they step by one K-tile per request and jump to the next tile's panels behind a tile's last K-tile
    unsigned a_cur = cur.a_so, b1_cur = cur.b_so + half_b, b0_cur = cur.b_so;
    int a_rec = cur.live ? (int)g.a_bytes : 0, b1_rec = cur.live ? (int)g.b_bytes : 0, b0_rec = b1_rec;

    auto dma_a = [&](auto ls_c) {
        constexpr int LS = decltype(ls_c)::value;
        char* dst = smem + sd[LS] + S_A * HALF + wave * 2048;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_rec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, offa[0], a_cur, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, offa[1], a_cur, 0, 0);
    };
    auto dma_b = [&](auto slot_c, auto ls_c) {
        constexpr int SLOT = decltype(slot_c)::value, LS = decltype(ls_c)::value;
        char* dst = smem + sd[LS] + SLOT * HALF + wave * 2048;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, SLOT == S_B1 ? b1_rec : b0_rec, 0x00020000);
        const unsigned so = SLOT == S_B1 ? b1_cur : b0_cur;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, offb[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, offb[1], so, 0, 0);
    };
    // a cursor behind its request for K-tile k of the current tile: the next K-tile, or — behind the tile's last — the next tile's first
    auto step_a = [&](int k) {
        if (k + 1 == nk) { a_cur = nxt.a_so; a_rec = nxt.live ? (int)g.a_bytes : 0; }
        else a_cur += (unsigned)(BK * 2);
    };
    auto step_b1 = [&](int k) {
        if (k + 1 == nk) { b1_cur = nxt.b_so + half_b; b1_rec = nxt.live ? (int)g.b_bytes : 0; }
        else b1_cur += kstep_b;
    };
    auto step_b0 = [&](int k) {
        if (k + 1 == nk) { b0_cur = nxt.b_so; b0_rec = nxt.live ? (int)g.b_bytes : 0; }
        else b0_cur += kstep_b;
    };

    // acc: the tile that accumulates; accd: the finished tile whose pieces leave ([m-tile][nh][n-tile]: D'[n][m] of the swapped
    // product — lane: m = l & 15, n = 4 (l >> 4) + r).  A tile's first k-step starts its chains from zero (ktile's FIRST form),
    // so acc is never cleared; at a tile's end acc is COPIED to accd (32 v_mov_b64 per wave, ~1 % of a K = 1024 tile).  Two
    // sets that swap roles instead — the loop body instantiated twice — left hipcc's register allocator with phis of both
    // sets at every join of the tile loop: 128 register copies per tile and spills inside the K loop.
    f32x4 acc[4][2][2], accd[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) { acc[i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accd[i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // PRE (row-form B): the B0 fragments of K-tile t + 1 are read a phase early, in phase B(t), into a register set of their own
    // (8 / 8 fragment reads per phase instead of 12 / 4: gemm8p_hm2_tile's schedule).  The col-form instantiations cannot spare
    // those 16 registers (their transposed reads come in halves that hipcc keeps apart): B0 is read with A0, into the set B1
    // uses a phase later, and every half-tile of K-tile t + 2 is requested for ONE stage (A(t): A0, B0; B(t): B1).
    constexpr bool PRE = !B_COL;
    V fa[4][2], fb0[2][2], fb1[2][2];

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
    auto load_a = [&](auto ls_c) {
        constexpr int LS = decltype(ls_c)::value;
        const unsigned a0 = ra0 + (unsigned)sd[LS], a1 = a0 ^ 64u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *reinterpret_cast<const V*>(smem + a0 + (S_A * HALF + i * 2048));
            fa[i][1] = *reinterpret_cast<const V*>(smem + a1 + (S_A * HALF + i * 2048));
        }
    };
    auto load_b = [&](auto ls_c, auto slot_c, V (&dst)[2][2]) {
        constexpr int LS = decltype(ls_c)::value, SLOT = decltype(slot_c)::value;
        if constexpr (B_COL) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned b = rb0[j] + (unsigned)sd[LS];
                dst[j][0] = col_frag(b, std::integral_constant<int, SLOT * HALF>{});
                dst[j][1] = col_frag(b, std::integral_constant<int, SLOT * HALF + 32 * HT<true>::ROWB>{});
            }
        } else {
            const unsigned b0 = rb0[0] + (unsigned)sd[LS], b1 = b0 ^ 64u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dst[j][0] = *reinterpret_cast<const V*>(smem + b0 + (SLOT * HALF + j * 2048));
                dst[j][1] = *reinterpret_cast<const V*>(smem + b1 + (SLOT * HALF + j * 2048));
            }
        }
    };

    // ---- the departing tile ----
    // piece e = 2 i + nh of the tile that finished last (rows 16 i .. of every wave's 64, column half nh) leaves in K-tile e of
    // the tile behind it.  Lane (m = l & 15, q = l >> 4) holds columns 16 j + 4 q + r of the wave's 32-column slab; after the
    // arithmetic and the rounding, two v_permlane16_swap give the lanes of the even 16-lane rows the eight columns 4 q .. 4 q + 7
    // (j = 0) and those of the odd rows the columns 16 + 4 (q - 1) .. + 7 (j = 1): ONE 16-byte store per lane.
    constexpr int NS = EPI == EPI_BITSW ? 2 : 1;  // store instructions per K-tile (dead offsets when no piece leaves)
    const int q = lane >> 4;
    const int colstart = (q & 1) * 16 + (q >> 1) * 8;  // first of this lane's eight stored columns within the slab
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)g.c_bytes, 0x00020000);
    const unsigned c_voff = (unsigned)(64 * wr + (lane & 15)) * ldc2 + (unsigned)((32 * wc + colstart) * 2);
    int dr_e = 8, dr_m0 = 0, dr_n0 = 0;  // piece to leave next (8: none), origin of the departing tile
    __amdgpu_buffer_rsrc_t rbit = __builtin_amdgcn_make_buffer_rsrc((void*)g.bits, 0, (int)g.bits_bytes, 0x00020000);
    const unsigned bit_voff = (unsigned)(64 * wr + (lane & 15)) * ldbits + (unsigned)(4 * wc);
    // side operands of the piece that leaves NEXT, read from LDS one K-tile ahead:
    //   EPI_LEAN / EPI_BITSW: this lane's 2 x 4 bias values (n-tile j = 0, 1) as packed 16-bit pairs; EPI_BITSR: [0][0] = the
    //   mask dword of the piece's 32 columns of this lane's row
    u32x2 side[2] = {{0u, 0u}, {0u, 0u}};
    const unsigned side_lds = EPI == EPI_BITSR ? (unsigned)(MASK_OFF + (64 * wr + (lane & 15)) * 32 + 4 * wc)
                                               : (unsigned)(BIAS_OFF + (32 * wc + 4 * q) * 2);
    auto side_read = [&](int e) {  // piece e of the tile whose side operands LDS holds
        if constexpr (EPI == EPI_BITSR) {
            side[0][0] = *reinterpret_cast<const unsigned*>(smem + side_lds + (e >> 1) * 512 + (e & 1) * 16);
        } else {
            side[0] = *reinterpret_cast<const u32x2*>(smem + side_lds + (e & 1) * 256);
            side[1] = *reinterpret_cast<const u32x2*>(smem + side_lds + (e & 1) * 256 + 32);
        }
    };
    // the tile's side operands -> LDS: ONE LDS-DMA instruction of wave 0 (bias: 64 lanes x 16 B = the tile's 256 values and the
    // 256 behind them) / of waves 0..3 (mask: rows 32 w .. 32 w + 31 of the tile, 2 lanes x 16 B each)
    auto side_dma = [&](int m0, int n0) {
        if constexpr (EPI == EPI_BITSR) {
            if (wave < 4) {
                const unsigned voff = (unsigned)(32 * wave + (lane >> 1)) * ldbits + (unsigned)((lane & 1) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbit, (lds_void*)(smem + MASK_OFF + wave * 1024), 16, voff,
                                                         (unsigned)m0 * ldbits + (unsigned)(n0 >> 3), 0, 0);
            }
        } else {
            if (wave == 0) {
                __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)g.bias, 0, g.bias ? N * 2 : 0, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (lds_void*)(smem + BIAS_OFF), 16, (unsigned)(lane * 16), (unsigned)(n0 * 2), 0, 0);
            }
        }
    };
    float alpha_v = g.alpha;
    asm volatile("" : "+v"(alpha_v));  // (a vector register: as a scalar operand of eight packed multiplies it took eight SGPRs)

    // one piece leaves in two steps.  piece_select (the load section of phase A): the piece's eight accumulator values of this
    // lane.  piece_math (the MFMA section of phase A, a single basic block WITH the 16 MFMAs, interleaved with them by
    // sched_group_barrier: a wave that only issues MFMAs leaves half of its issue slots unused, and as a block of its own in
    // front of them the arithmetic added its full 300-400 cycles to every K-tile): -> the packed 16 bytes this lane stores
    // (and, EPI_BITSW, the mask dword of its row's 32 columns).
    auto piece_select = [&](const f32x4 (&ad)[4][2][2], float (&y)[8]) {
        const int e = dr_e;
        // (the piece index is a run-time value: the accumulators are chosen by a scalar switch, never by an indexed access, and
        // each case moves its eight values by inline asm: plain assignments are sunk by SimplifyCFG into one block behind a
        // computed index — while this lambda is still a function of its own, the set a pointer parameter — and the
        // accumulators end up in scratch memory.  The set is only READ here: clearing the piece in its case made every
        // accumulator a phi of eight versions at the join; a tile's first k-step starts its chains from zero instead — ktile's
        // FIRST form.  The multiply by alpha sits behind the join: inside the cases hipcc computed all 64 products ahead of
        // the switch and spilled them.)
#define PK_TAKE(I, NH)                                                                                      \
    {                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int r = 0; r < 4; ++r)         \
            asm volatile("v_mov_b32 %0, %1 ; piece %2" : "=v"(y[4 * j + r]) : "v"(ad[I][NH][j][r]), "i"(2 * I + NH)); \
    }
        switch (e) {
            case 0: PK_TAKE(0, 0) break;
            case 1: PK_TAKE(0, 1) break;
            case 2: PK_TAKE(1, 0) break;
            case 3: PK_TAKE(1, 1) break;
            case 4: PK_TAKE(2, 0) break;
            case 5: PK_TAKE(2, 1) break;
            case 6: PK_TAKE(3, 0) break;
            default: PK_TAKE(3, 1) break;
        }
#undef PK_TAKE
    };
    auto piece_math = [&](const float (&y0)[8], unsigned& bits_word) -> u32x4 {
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = y0[k] * alpha_v;
        if constexpr (EPI == EPI_BITSR) {
            const unsigned word = side[0][0] >> (4 * q);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[4 * j + r] = ((word >> (16 * j + r)) & 1u) ? y[4 * j + r] : 0.f;
        } else {
            const unsigned bw[4] = {side[0][0], side[0][1], side[1][0], side[1][1]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                y[2 * k] += H16<T>::val((unsigned short)(bw[k] & 0xffffu));
                y[2 * k + 1] += H16<T>::val((unsigned short)(bw[k] >> 16));
            }
            if constexpr (EPI == EPI_BITSW) {  // (a ReLU without the mask bits stays on gemm8p.hip's tiles: no caller at these shapes)
#pragma unroll
                for (int k = 0; k < 8; ++k) y[k] = fmaxf(y[k], 0.f);
            } else {  // (an opaque point between the arithmetic and the rounding: without one — the ReLU is one — hipcc's register
                // allocation of THIS instantiation falls apart: accumulators that change registers in 150 of 192 MFMAs, 260 bytes
                // of spills in the K loop; found by bisection against the instantiation with the ReLU)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(y[k]));
            }
        }
        typedef __attribute__((ext_vector_type(8))) float f32x8;
        f32x8 f = {y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7]};
        const u32x4 o = __builtin_bit_cast(u32x4, __builtin_convertvector(f, typename H16<T>::vec));
        // o[0], o[1]: columns 4 q .. + 3 of n-tile 0; o[2], o[3]: of n-tile 1
        const auto s0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
        const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
        bits_word = 0u;
        if constexpr (EPI == EPI_BITSW) {
            // bit k = (the stored value > 0), without compares: after max(0, .) no 16-bit pattern has its sign set, so adding
            // 0x7FFF carries into bit 15 / 31 exactly where a half is not zero (gemm8p.hip: epilogue_pass_bits)
            unsigned u = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned ow = out[w];
                u |= ((ow + 0x7FFF7FFFu) >> (6 - 2 * w)) & (0x80008000u >> (6 - 2 * w));
            }
            const unsigned byte = ((u >> 9) | (u >> 24)) & 0xFFu;  // this lane's eight columns: slab columns colstart .. + 7
            // the four bytes of a row sit in the lane groups q = 0, 2 (columns 0-7, 8-15) and 1, 3 (16-23, 24-31): the pair
            // (q, q ^ 1) by v_permlane16_swap, the two pairs by v_permlane32_swap; group 0 stores the dword
            const auto p16 = __builtin_amdgcn_permlane16_swap(byte, byte, false, false);  // [0]: the even group's byte, [1]: the odd one's
            const unsigned pair2 = p16[0] | (p16[1] << 16);  // columns (0-7 | 16-23) for q < 2, (8-15 | 24-31) for q >= 2
            const auto p32 = __builtin_amdgcn_permlane32_swap(pair2, pair2, false, false);  // lanes < 32: [0] own pair, [1] the upper half's
            bits_word = p32[0] | (p32[1] << 8);
        }
        return out;
    };
    // the K-tile's NS stores: the piece (dr_e < 8) or the same instructions against dead offsets
    auto piece_store = [&](bool on, int e, const u32x4& out, unsigned bits_word) {
        const int i = e >> 1, nh = e & 1;
        const int gn = dr_n0 + 128 * nh;  // (+ 32 wc + colstart: per lane)
        const bool ok = on && gn + 32 * wc + colstart + 8 <= (int)g.nstore;
        const unsigned so = on ? (unsigned)(dr_m0 + 16 * i) * ldc2 + (unsigned)gn * 2u : 0u;
        if constexpr (EPI == EPI_BITSW) {
            const unsigned bso = on ? (unsigned)(dr_m0 + 16 * i) * ldbits + (unsigned)(gn >> 3) : 0u;
            __builtin_amdgcn_raw_buffer_store_b32(bits_word, rbit, (q == 0 && ok) ? bit_voff : DEAD_OFF, bso, 0);
        }
#ifndef PKPW_STORE_AUX
#define PKPW_STORE_AUX 2 /* nt */
#endif
#ifdef PKPW_ABL_DEADSTORES
        __builtin_amdgcn_raw_buffer_store_b128(out, rc, DEAD_OFF, so, PKPW_STORE_AUX);
#else
        __builtin_amdgcn_raw_buffer_store_b128(out, rc, ok ? c_voff : DEAD_OFF, so, PKPW_STORE_AUX);
#endif
    };

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using IB0 = std::integral_constant<int, S_B0>; using IB1 = std::integral_constant<int, S_B1>;

    // one K-tile of the stream: logical stage LS; FIRST: the tile's K-tile 0 (always logical stage 0), whose first k-step
    // starts the accumulation chains from zero
    auto ktile = [&](auto ls_c, auto first_c, int kt) {
        constexpr int LS = decltype(ls_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        using L1 = std::integral_constant<int, (LS + 1) % 3>; using L2 = std::integral_constant<int, (LS + 2) % 3>;
        auto mma = [&](int nh, V (&b)[2][2]) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][nh][j] = M16<T>::mfma(b[j][kk], fa[i][kk], (FIRST && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][nh][j]);
        };
        // ---- phase A: quadrant (0, 0) ----
        load_a(ls_c);
        if constexpr (!PRE) load_b(ls_c, IB0{}, fb0);
        if constexpr (NS == 1) PK_WAIT(7); else PK_WAIT(8);  // B1(t) (PRE: and B0(t+1)) have landed
        dma_a(L2{});
        step_a(kt + 2 < nk ? kt + 2 : kt + 2 - nk);
        if constexpr (!PRE) {
            dma_b(IB0{}, L2{});
            step_b0(kt + 2 < nk ? kt + 2 : kt + 2 - nk);
        }
        if (kt == 7) side_dma(cur.m0, cur.n0);  // (for the pieces of THIS tile; the previous tile's last piece read its own in K-tile 6)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // EVERYTHING the departing tile needs sits in MFMA sections: a load section (fragment reads, the counted wait, the DMA
        // requests: ~400 cycles) is longer than an MFMA section (16 MFMAs: 256), and the K-tile lasts 2 x (load A + load B) —
        // work added to a load section costs twice its length, work added to an MFMA section nothing until the two are equal
        const bool leaving = dr_e < 8;
        float y0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if !defined(PKPW_ABL_NOEMIT) && !defined(PKPW_ABL_NOSELECT)
        if (leaving) piece_select(accd, y0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // the MFMA section also carries the arithmetic of the piece that leaves (of zeros when none does: the block has no branch)
        unsigned bits_word = 0u;
        __builtin_amdgcn_s_setprio(1);  // (around the whole block: a priority change inside it would pin the instruction order)
#if !defined(PKPW_ABL_NOEMIT) && !defined(PKPW_ABL_NOMATH)
        const u32x4 out = piece_math(y0, bits_word);
#elif defined(PKPW_ABL_NOMATH)
        const u32x4 out = {__float_as_uint(y0[0]), __float_as_uint(y0[1]), __float_as_uint(y0[2]), __float_as_uint(y0[3])};
#else
        const u32x4 out = {0u, 0u, 0u, 0u};
#endif
        mma(0, fb0);
#ifndef PKPW_ABL_NOEMIT
        piece_store(leaving, dr_e, out, bits_word);
#ifndef PKPW_NO_SGB
        // instruction order of the block: 16 x (one MFMA, three vector instructions), then whatever is left and the store(s)
#define PK_G __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G PK_G
#undef PK_G
        __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, NS, 0);
#endif
#endif
        __builtin_amdgcn_s_setprio(0);
        if (leaving) ++dr_e;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B: quadrant (0, 1); PRE: the B0 fragments of the NEXT K-tile; the side operands of the piece that leaves next ----
        if constexpr (PRE) {
            load_b(ls_c, IB1{}, fb1);
            load_b(L1{}, IB0{}, fb0);
        } else {
            load_b(ls_c, IB1{}, fb0);
        }
        if constexpr (NS == 1) PK_WAIT(8); else PK_WAIT(10);  // A0(t+1) (!PRE: and B0(t+1)) have landed
        dma_b(IB1{}, L2{});
        step_b1(kt + 2 < nk ? kt + 2 : kt + 2 - nk);
        if constexpr (PRE) {
            dma_b(IB0{}, ls_c);
            step_b0(kt + 3 < nk ? kt + 3 : kt + 3 - nk);
        }
        asm volatile("" :: "v"(out));  // (the store's data registers stay untouched until here)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#if !defined(PKPW_ABL_NOEMIT) && !defined(PKPW_ABL_NOSIDE)
        if (dr_e < 8) side_read(dr_e);        // (the side operands of the piece that leaves in the NEXT K-tile)
        else if (kt == nk - 1) side_read(0);  // (piece 0 of THIS tile: it leaves in the next one's K-tile 0)
#endif
        __builtin_amdgcn_s_setprio(1);
        if constexpr (PRE) mma(1, fb1); else mma(1, fb0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    // rotate the logical stages by one ring position (logical 0 <- old logical 1, ...)
    auto rotate1 = [&]() {
        const int s0 = sd[0]; sd[0] = sd[1]; sd[1] = sd[2]; sd[2] = s0;
    };
    // one tile of the stream
    auto run_tile = [&]() {
        ktile(I0{}, std::true_type{}, 0);  // (nk >= 10: the first three K-tiles need no bound check)
        ktile(I1{}, std::false_type{}, 1);
        ktile(I2{}, std::false_type{}, 2);
        for (int kt = 3; kt < nk; kt += 3) {
            ktile(I0{}, std::false_type{}, kt);
            if (kt + 1 >= nk) break;
            ktile(I1{}, std::false_type{}, kt + 1);
            if (kt + 2 >= nk) break;
            ktile(I2{}, std::false_type{}, kt + 2);
        }
        // (asm moves: a plain assignment is coalesced away — accd becomes the registers acc had, the next tile accumulates
        // into fresh ones, and the sets are permuted back at every join of the loop: ~450 moves and spills in the K loop)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 lo = {acc[i][b][j][0], acc[i][b][j][1]}, hi = {acc[i][b][j][2], acc[i][b][j][3]};
                    f32x2 dlo, dhi;
                    asm volatile("v_mov_b64 %0, %1" : "=v"(dlo) : "v"(lo));
                    asm volatile("v_mov_b64 %0, %1" : "=v"(dhi) : "v"(hi));
                    accd[i][b][j] = f32x4{dlo[0], dlo[1], dhi[0], dhi[1]};
                }
        const int rot = nk % 3;
        if (rot >= 1) rotate1();
        if (rot == 2) rotate1();
        dr_e = 0; dr_m0 = cur.m0; dr_n0 = cur.n0;
        cur = nxt;
        nxt = tile_ref(vb_next);
        vb_next += nwg;
    };

    // ---- prologue of the stream: K-tiles 0, 1 whole and K-tile 2's B0, in the issue order of the steady state — with the NS
    // (dead) stores where a K-tile has them, so that the first counted waits see the queue they were counted for ----
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    dma_b(IB0{}, I0{}); step_b0(0);
    dma_a(I0{}); step_a(0);
    dma_b(IB1{}, I0{}); step_b1(0);
    dma_b(IB0{}, I1{}); step_b0(1);
    dma_a(I1{}); step_a(1);
    piece_store(false, 0, zero4, 0u);
    dma_b(IB1{}, I1{}); step_b1(1);
    if constexpr (PRE) { dma_b(IB0{}, I2{}); step_b0(2); }
    PK_WAIT(8);  // B0, A0 of K-tile 0 (and whatever the prologue put behind them, but for four half-tiles)
    asm volatile("; PK8P_LOOP_BEGIN" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (PRE) load_b(I0{}, IB0{}, fb0);  // B0 of K-tile 0 (every later one is read a phase ahead, inside the loop)
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave of every SIMD runs one barrier behind the first
    while (cur.live) run_tile();
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("; PK8P_LOOP_END" ::: "memory");
    PK_WAIT(0);
#undef PK_WAIT
#undef PK_TR
    // ---- the last tile's results (its side operands are in LDS since its K-tile 7; piece 0's were read in its last K-tile) ----
    if ((int)blockIdx.x < total) {
        __syncthreads();
        for (int e = 0; e < 8; ++e) {
            unsigned bits_word = 0u;
            float y0[8];
            piece_select(accd, y0);
            const u32x4 out = piece_math(y0, bits_word);
            piece_store(true, dr_e, out, bits_word);
            ++dr_e;
            if (dr_e < 8) side_read(dr_e);
            asm volatile("" :: "v"(out));
        }
    }
}

}  // namespace

namespace {
// OFF by default (PK_GEMM_PW=1 / pk_gemm_use_pw(1) turn it on): correct — bit for bit the tiled kernels — but measured SLOWER
// than them (see the note at the top of the file and docs/experiments.md, "Round 6: the persistent GEMM").
// (a mask: bit 0 = this file's kernel, bit 1 = the persistent walk of 256 x 256 tiles in gemm8p.hip; default 2)
int g_use_pw = [] { const char* e = getenv("PK_GEMM_PW"); return e ? (atoi(e) & 3) : 2; }();
}
extern "C" int pk_gemm_use_pw(int on) {
    const int old = g_use_pw;
    if (on >= 0) g_use_pw = on & 3;
    return old;
}

// 1 if pk_gemm8p_launch should hand this GEMM to the persistent kernel: row-form A, whole K-tiles, at least ten of them (the
// previous tile needs eight K-tiles to leave), an output of at least two rounds of
// 128 x 256 tiles, a lean epilogue.  PK_GEMM_PW=0: off (A/B).
extern "C" int pk_gemmpw_eligible(long long M, long long N, long long K, long long lda, long long ldb, long long ldc, int b_col,
                                  const EpiParams* ep) {
    static const int min_tiles = [] { const char* e = getenv("PK_GEMM_PW_MIN_TILES"); return e ? atoi(e) : 512; }();
    static const int max_nk = [] { const char* e = getenv("PK_GEMM_PW_MAX_NK"); return e ? atoi(e) : 32; }();
    if (!(g_use_pw & 1)) return 0;
    if (K % BK || K / BK < 10 || K / BK > max_nk) return 0;
    // (N % 8 != 0: pk_gemm_ex's padded rows — `nstore` columns may be stored; rows of a row-form B past N read as zeros)
    if ((N % 8 && (b_col || !ep->nstore)) || (lda % 8) || (ldb % 8) || (ldc % 8)) return 0;
    const long long tiles = ((M + TM - 1) / TM) * ((N + BN - 1) / BN);
    if (tiles < min_tiles || tiles > 0x7fffffffLL / 4) return 0;
    if (ep->preact || ep->aux) return 0;
    if (ep->bits) {
        if (ep->act != PK_ACT_RELU || (ep->mode != 0 && ep->mode != 2) || N % 32 || ep->ldbits % 4) return 0;
    } else if (ep->mode != 0 || ep->act != PK_ACT_NONE) return 0;
    const long long nstore = ep->nstore ? ep->nstore : N;
    const long long c_bytes = ((M - 1) * ldc + nstore) * 2;
    if (c_bytes > 0x7FFFFFFFLL - 65536) return 0;
    if (ep->bits && ((M - 1) * ep->ldbits + N / 8) > 0x7FFFFFFFLL - 65536) return 0;
    return 1;
}

// Returns 1 if launched, a hip error code otherwise.  The caller (gemm8p.hip: pk_gemm8p_launch) has checked pk_gemm8p_eligible
// (operands below 2 GiB, K % 8 == 0) and pk_gemmpw_eligible.
extern "C" int pk_gemmpw_launch(const void* A, const void* B, void* C, long long M, long long N, long long K, long long lda,
                                long long ldb, unsigned a_bytes, unsigned b_bytes, int b_col, EpiParams ep, int dtype,
                                void* stream) {
    PwArgs g;
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ep.ldc;
    g.a_bytes = a_bytes; g.b_bytes = b_bytes;
    g.nstore = ep.nstore ? ep.nstore : N;
    g.c_bytes = (unsigned)(((M - 1) * ep.ldc + g.nstore) * 2);
    g.bits = ep.bits; g.ldbits = ep.ldbits;
    g.bits_bytes = ep.bits ? (unsigned)((M - 1) * ep.ldbits + N / 8) : 0u;
    g.total = (int)(((M + TM - 1) / TM) * ((N + BN - 1) / BN));
    g.alpha = ep.alpha;
    g.bias = ep.mode == 0 ? ep.bias : nullptr;
    g.stamps = nullptr;
    static const int wgs = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return std::max(8, cus / 8 * 8);
    }();
    dim3 grid((unsigned)std::min(wgs, g.total / 8 * 8 > 0 ? g.total / 8 * 8 : 8)), block(512);
    hipStream_t s = (hipStream_t)stream;
    const int epi = ep.bits ? (ep.mode == 0 ? EPI_BITSW : EPI_BITSR) : EPI_LEAN;
#define PK_P(TT, BC, E) hipLaunchKernelGGL((gemm8p_pw_kernel<TT, BC, E>), grid, block, 0, s, g)
#define PK_PE(TT, BC)                              \
    do {                                           \
        if (epi == EPI_BITSW) PK_P(TT, BC, EPI_BITSW); \
        else if (epi == EPI_BITSR) PK_P(TT, BC, EPI_BITSR); \
        else PK_P(TT, BC, EPI_LEAN);               \
    } while (0)
    if (dtype == PK_F16) { if (b_col) PK_PE(f16, true); else PK_PE(f16, false); }
    else { if (b_col) PK_PE(bf16, true); else PK_PE(bf16, false); }
#undef PK_PE
#undef PK_P
    PK_LAUNCH_CHECK();
    return 1;
}
