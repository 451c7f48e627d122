// Attention kernels on the LDS-DMA frame (round 4): forward (`attn_fwd_long_kernel`, 64 keys and more), dQ and dK / dV for long
// streamed sequences (`attn_dq_long_kernel`, `attn_dkv_long_kernel`, 256 and more); heads of 64, no rotation.  The tiled kernels
// of attention.hip keep everything else and remain the reference for the arithmetic.  Built with -fno-slp-vectorize: packed f32
// instructions beside MFMAs are slower than the scalar pairs (MI355X_MICROARCH.md, cycle constants), and the measured-and-dropped
// two-tiles-in-flight form of the forward needed its hand-placed instruction groups left where they were put.
#include "attention_common.h"

namespace {

// ---- forward for key sequences of 64 and more (round 4): head_dim 64, no rotation ----
// attn_q_kernel (attention.hip) spent ~10 vector instructions per (query, key) score (scale + bias, the causal select, running maximum,
// subtraction, exp, sum, accumulator rescale, conversion) against 16 MFMAs per 2048 scores: at S >= 500 it is paced by its
// VALU work, not by the matrix pipe (0.5 ps per pair whatever the shape).  This form keeps 4 per score (causal launches too:
// whole tiles in a wave's future are skipped, the diagonal's are masked in a wave-uniform branch):
//   * the maximum LAGS: a query's scores are taken relative to the maximum m its row was last anchored at, x = s c - m (no
//     subtraction: the MFMA chain starts from the accumulator -m / c, one multiply by c follows), and m moves only when a tile's
//     maximum exceeds it by more than LAG_THR (exp2 domain: p <= 2^LAG_THR, in range for both 16-bit types, relative precision
//     unchanged) — the accumulator rescale, the exp of the correction and the
//     subtraction leave the loop body for a wave-uniform branch that is rare after the first tile;
//   * keys past S / padding keys are found per tile by one ballot; only tiles that hold one pay for the selects;
//   * K and V tiles arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, rows past S read as zeros through the buffer
//     bound) into a two-deep ring of dual-use images (lds_off<DUAL>: a 1-KiB DMA piece is one 8-row group, the lane picks the
//     global chunk that belongs at its LDS position), one barrier per tile, nothing staged through registers;
//   * workgroups of one (batch, head) pair sit on one XCD (their K / V stay in that L2).
// Same accumulator layouts, Q fragments and row stores as attn_q_kernel.  A row's arithmetic depends on its own data only
// (a lane re-anchors only when ITS maximum says so).
constexpr float LAG_THR = 8.f;
#ifndef PKL_DQ_WAVES
#define PKL_DQ_WAVES 3  // (the dQ kernel below: 174 registers left alone, 168 under this budget without a spill)
#endif
#ifndef PKL_WAVES
#define PKL_WAVES 3  // waves per SIMD the register budget leaves room for
#endif

template <typename T, bool DROP>
__global__ __launch_bounds__(256, PKL_WAVES) void attn_fwd_long_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, T* __restrict__ o,
                                                               float* __restrict__ lse, AttnParams p, int nqb, int npairs) {
    constexpr int HD = 64, NF = 4, ND = 2;
    constexpr int IMG = img_bytes<DUAL>();  // 8 KiB: [64 rows][64 x 16 bit]
    typedef __attribute__((address_space(3))) void lds_void;
#ifndef PKL_LDS_PAD
#define PKL_LDS_PAD 0  // diagnostic builds: extra LDS per workgroup (fewer workgroups per CU)
#endif
    __shared__ __attribute__((aligned(1024))) char ring[2 * 2 * IMG + PKL_LDS_PAD];  // stage st: K image at 2 st IMG, V image behind it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lin = blockIdx.x, grp = (lin >> 3) / nqb;
    // (causal: the heaviest query blocks — the last ones see the most keys — are dealt first)
    const int qb = p.causal ? nqb - 1 - (lin >> 3) % nqb : (lin >> 3) % nqb;
    const int pair = grp * 8 + (lin & 7);
    if (pair >= npairs) return;
    const int b = pair / p.H, h = pair % p.H;
    const int t = qb * 128 + wave * 32 + (lane & 31);
    const bool valid = t < p.T;
    const float c = p.scale * LOG2E, inv_c = 1.f / c;

    bf16x8_t qf[NF];
    load_row_frags(qf, q + b * p.q_bs + h * HD, p.q_rs, t, valid, lane);
    f32x16 acc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float negm = 0.f, l = 0.f;  // -m (exp2 domain) once the row is anchored, 0 before
    f32x16 negm16;              // -m / c in every register: the initial accumulator of the score MFMAs (units of the raw score)
#pragma unroll
    for (int r = 0; r < 16; ++r) negm16[r] = 0.f;
    bool anch = false;

    // LDS-DMA: wave w brings the 8-row pieces w and w + 4 of the K and of the V tile; lane L lands at byte 16 L of its piece
    const T* kbase = k + b * p.k_bs + h * HD;
    const T* vbase = v + b * p.v_bs + h * HD;
    const unsigned k_rsb = (unsigned)(p.k_rs * 2), v_rsb = (unsigned)(p.v_rs * 2);
    const int kbytes = (int)(((long long)(p.S - 1) * p.k_rs + HD) * 2), vbytes = (int)(((long long)(p.S - 1) * p.v_rs + HD) * 2);
    unsigned koff[2], voff[2];
    {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * (wave + 4 * i) + ((lane >> 2) & 7);
            const int ch = 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));  // (the inverse of lds_off<DUAL>)
            koff[i] = (unsigned)row * k_rsb + 16 * ch;
            voff[i] = (unsigned)row * v_rsb + 16 * ch;
        }
    }
    // transposed reads of the V image (tr_frag<DUAL>): lane addresses of rows 4 h + q and 4 h + q + 8 of d-tile 0; key block kb and
    // k-step s add 1024 (4 kb + 2 s) bytes (16 keys = two 8-row groups of 1 KiB, the swizzle class of a row does not change),
    // d-tile 1 adds 512 (chunks 4..7 of a row)
    unsigned vaddr[2];
    {
        typedef __attribute__((address_space(3))) char lds_char;
        const int qd = (lane & 15) >> 2, p4 = lane & 3;
        const int col = 16 * ((lane >> 4) & 1) + 4 * p4, row = 4 * (lane >> 5) + qd;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)ring + (col & 7) * 2;
        vaddr[0] = base + lds_off<DUAL>(row, col >> 3);
        vaddr[1] = base + lds_off<DUAL>(row + 8, col >> 3);
    }
    // causal (query t sees keys <= t + S - T): tiles beyond the last query of the workgroup are never visible; a wave skips the
    // tiles that lie in the future of all its queries and masks inside the ones its diagonal crosses (both wave-uniform)
    const int off = p.S - p.T, wt0 = qb * 128 + wave * 32;
    const int ntiles = p.causal ? max(0, min((p.S + KT - 1) / KT, (qb * 128 + 128 + off + KT - 1) / KT)) : (p.S + KT - 1) / KT;
    auto dma = [&](int tile, int st) {
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, kbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, vbytes, 0x00020000);
        char* kd = ring + st * 2 * IMG + wave * 1024;
        const unsigned ks = (unsigned)(tile * KT) * k_rsb, vs = (unsigned)(tile * KT) * v_rsb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(kd + 4096 * i), 16, koff[i], ks, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(kd + IMG + 4096 * i), 16, voff[i], vs, 0, 0);
        }
    };
    // one byte per (tile, lane): is key 64 tile + lane past S or a padding key
    const unsigned char* padrow = p.key_pad ? p.key_pad + (long long)b * p.S : nullptr;
    auto pad_of = [&](int tile) -> unsigned {
        const int sk = tile * KT + lane;
        if (sk >= p.S) return 1u;
        return padrow ? (unsigned)padrow[sk] : 0u;
    };
    const long long mrow = ((long long)b * p.H + h) * p.T + t;

    auto body = [&](int tile, const char* k_lds, const char* v_lds, unsigned padb) {
        const int s0 = tile * KT;
        if (p.causal && s0 > wt0 + 31 + off) return;               // every key is in the future of every query of this wave
        const bool check = p.causal && s0 + KT - 1 > wt0 + off;    // some (query, key) pairs are masked
        const unsigned long long dead = __ballot(padb != 0);
        f32x16 sc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            sc[kb] = mm<T>(row_frag<DUAL>(k_lds, kb * 32, 0, lane), qf[0], negm16);
#pragma unroll
            for (int kk = 1; kk < NF; ++kk) sc[kb] = mm<T>(row_frag<DUAL>(k_lds, kb * 32, kk, lane), qf[kk], sc[kb]);
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kb][r] *= c;  // x = s c - m
        }
        if (dead) {  // wave-uniform: this tile holds masked keys
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                unsigned dm = (unsigned)(dead >> (32 * kb)) >> (4 * (lane >> 5));
                asm volatile("; masked keys" : "+v"(dm));  // (keeps this a real branch: if-converted, the selects cost 3 instructions per score in every tile)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((dm >> ((r & 3) + 8 * (r >> 2))) & 1u) sc[kb][r] = -INFINITY;
            }
        }
        if (check) {  // wave-uniform: the causal boundary crosses this (wave, tile) block
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                int kq = t + off - s0 - kb * 32 - 4 * (lane >> 5);  // key index (in the block) > kq is in this query's future
                asm volatile("; causal block" : "+v"(kq));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) > kq) sc[kb][r] = -INFINITY;
            }
        }
        float tmax = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sc[0][r]), sc[1][r]);
        tmax = half_wave_max(tmax);
        const bool move = anch ? tmax > LAG_THR : tmax > -INFINITY;
        if (__any(move)) {  // (re-)anchor the rows that ask for it: everything held at the old maximum is scaled exactly once
            const float delta = move ? tmax : 0.f;
            const float alpha = (move && anch) ? __builtin_amdgcn_exp2f(-delta) : 1.f;  // unanchored rows hold zeros
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            l *= alpha;
            negm -= delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm16[r] = negm * inv_c;
            anch = anch || move;
        }
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};  // packed adds: half the instructions of the row sum
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(sc[kb][r]), p1 = __builtin_amdgcn_exp2f(sc[kb][r + 1]);
                sc[kb][r] = p0;
                sc[kb][r + 1] = p1;
                ps2[kb] += f32x2{p0, p1};
            }
        l += (ps2[0].x + ps2[1].x) + (ps2[0].y + ps2[1].y);  // the softmax denominator does not see the dropout
        if constexpr (DROP) {  // the bits of the kernel above: one Philox draw per 8 keys of a row, a dword of keep bits per 32 keys
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                unsigned word = 0;
                const unsigned thr16 = p.drop_thr >> 16;
                const bool hi = lane >= 32;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int g_own = 2 * gp + (hi ? 1 : 0);
                    const unsigned long long ctr = ((unsigned long long)mrow * 8ull * p.mask_pitch + (s0 + kb * 32 + 8 * g_own)) >> 3;
                    Philox4 rr = philox4x32_10(p.seed, p.offset, ctr);
                    // the lower half-wave keeps (x, y) = its 4 keys of group 2 gp and needs the upper one's (x, y) for group 2 gp + 1; the
                    // upper one keeps (z, w) and needs the lower one's: one swap per register pair
                    half_wave_swap(rr.x, rr.z);
                    half_wave_swap(rr.y, rr.w);
                    const unsigned wv[2][2] = {{rr.x, rr.y}, {rr.z, rr.w}};
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int g = 2 * gp + u;
                        const bool keep[4] = {(wv[u][0] & 0xffffu) >= thr16, (wv[u][0] >> 16) >= thr16,
                                              (wv[u][1] & 0xffffu) >= thr16, (wv[u][1] >> 16) >= thr16};
                        unsigned nib = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            nib |= (unsigned)keep[j] << j;
                            if (!keep[j]) sc[kb][4 * g + j] = 0.f;
                        }
                        word |= nib << (8 * g + 4 * (lane >> 5));
                    }
                }
                {
                    unsigned wa = word, wb = word;  // (the two half-waves hold the low / high nibbles of the same bytes)
                    half_wave_swap(wa, wb);
                    word = wa | wb;
                }
                if (valid && lane < 32)
                    *reinterpret_cast<unsigned*>(p.drop_mask + mrow * p.mask_pitch + ((s0 + kb * 32) >> 3)) = word;
            }
        }
        // Oᵀ[d][query] += Vᵀ[d][key] · P[key][query].  The transposed reads are inline asm: behind the builtin the compiler drains
        // vmcnt before every LDS read that might alias the DMA in flight (the next tile's), which would put the whole flight
        // time in front of these MFMAs.  The wait that covers them names the fragments, so no MFMA can be scheduled above it.
        s16x4 vt[2][2][ND][2];  // [key block][k-step][d-tile][rows +0 / +8]
        const unsigned va0 = vaddr[0] + (unsigned)(v_lds - ring), va8 = vaddr[1] + (unsigned)(v_lds - ring);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[kb][s][dt][0]) : "v"(va0), "i"(1024 * (4 * kb + 2 * s) + 512 * dt));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[kb][s][dt][1]) : "v"(va8), "i"(1024 * (4 * kb + 2 * s) + 512 * dt));
                }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vt[0][0][0][0]), "+v"(vt[0][0][0][1]), "+v"(vt[0][0][1][0]), "+v"(vt[0][0][1][1]),
                       "+v"(vt[0][1][0][0]), "+v"(vt[0][1][0][1]), "+v"(vt[0][1][1][0]), "+v"(vt[0][1][1][1]),
                       "+v"(vt[1][0][0][0]), "+v"(vt[1][0][0][1]), "+v"(vt[1][0][1][0]), "+v"(vt[1][0][1][1]),
                       "+v"(vt[1][1][0][0]), "+v"(vt[1][1][0][1]), "+v"(vt[1][1][1][0]), "+v"(vt[1][1][1][1]));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t pf = acc_frag<T>(sc[kb], s);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const s16x4 lo = vt[kb][s][dt][0], hi = vt[kb][s][dt][1];
                    const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    acc[dt] = mm<T>(__builtin_bit_cast(bf16x8_t, f), pf, acc[dt]);
                }
            }
    };

    if (ntiles > 0) dma(0, 0);
    unsigned padb = ntiles > 0 ? pad_of(0) : 0u;
    for (int tile = 0; tile < ntiles; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile (and its stores of the tile before)
        __syncthreads();                                   // everyone's pieces; the other stage is free
        unsigned padn = 0;
        if (tile + 1 < ntiles) {
            dma(tile + 1, (tile + 1) & 1);
            padn = pad_of(tile + 1);
        }
        const char* st = ring + (tile & 1) * 2 * IMG;
        body(tile, st, st + IMG, padb);
        padb = padn;
    }
    l += __shfl_xor(l, 32, 64);
    float inv = l > 0.f ? 1.f / l : 0.f;
    if constexpr (DROP) inv *= p.drop_scale;
    store_rowT(o + b * p.o_bs + h * HD, p.o_rs, t, valid, acc, inv, lane);
    if (valid && lane < 32) lse[mrow] = l > 0.f ? (log2f(l) - negm) * LN2 : 0.f;
}


// ---- dQ for long key sequences: the frame of the forward kernel above around the dQ arithmetic of attention.hip ----
// (query on the lane; K and V tiles by LDS-DMA into the two-deep ring, one barrier per tile, masked keys per tile by one
// ballot, one 32-key block at a time: S starts from the accumulator -lse / scale, dP from -delta, p = exp2(c acc),
// dS = p dP', dQᵀ += Kᵀ dSᵀ with the transposed K reads in inline asm — see the forward kernel for why.)  Also writes
// delta = rowsum(dO o) for the dK / dV kernel that follows.  Heads of 64, no causal mask, no rotation.
template <typename T, bool DROP>
__global__ __launch_bounds__(256, PKL_DQ_WAVES) void attn_dq_long_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, const T* __restrict__ o,
                                                              const T* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ delta, T* __restrict__ dq, AttnParams p,
                                                              int nqb, int npairs) {
    constexpr int HD = 64, NF = 4, ND = 2;
    constexpr int IMG = img_bytes<DUAL>();
    typedef __attribute__((address_space(3))) void lds_void;
    __shared__ __attribute__((aligned(1024))) char ring[2 * 2 * IMG];  // stage st: K image at 2 st IMG, V image behind it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lin = blockIdx.x, grp = (lin >> 3) / nqb;
    const int qb = p.causal ? nqb - 1 - (lin >> 3) % nqb : (lin >> 3) % nqb;  // (causal: heaviest query blocks first)
    const int pair = grp * 8 + (lin & 7);
    if (pair >= npairs) return;
    const int b = pair / p.H, h = pair % p.H;
    const int t = qb * 128 + wave * 32 + (lane & 31);
    const bool valid = t < p.T;
    const float c = p.scale * LOG2E;
    const long long mrow = ((long long)b * p.H + h) * p.T + t;

    bf16x8_t qf[NF], dof[NF];
    load_row_frags(qf, q + b * p.q_bs + h * HD, p.q_rs, t, valid, lane);
    load_row_frags(dof, d_o + b * p.do_bs + h * HD, p.do_rs, t, valid, lane);
    float dl = 0.f, s_init = -INFINITY;  // rows past T: exp2(c (s - inf)) = 0
    {
        bf16x8_t of[NF];
        load_row_frags(of, o + b * p.o_bs + h * HD, p.o_rs, t, valid, lane);
#pragma unroll
        for (int kk = 0; kk < NF; ++kk) dl += frag_dot<T>(dof[kk], of[kk]);
        dl += __shfl_xor(dl, 32, 64);
        if (valid) {
            s_init = -lse[mrow] / p.scale;
            if (lane < 32) delta[mrow] = dl;
        }
    }
    f32x16 acc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    const T* kbase = k + b * p.k_bs + h * HD;
    const T* vbase = v + b * p.v_bs + h * HD;
    const unsigned k_rsb = (unsigned)(p.k_rs * 2), v_rsb = (unsigned)(p.v_rs * 2);
    const int kbytes = (int)(((long long)(p.S - 1) * p.k_rs + HD) * 2), vbytes = (int)(((long long)(p.S - 1) * p.v_rs + HD) * 2);
    unsigned koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + 4 * i) + ((lane >> 2) & 7);
        const int ch = 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));  // (the inverse of lds_off<DUAL>)
        koff[i] = (unsigned)row * k_rsb + 16 * ch;
        voff[i] = (unsigned)row * v_rsb + 16 * ch;
    }
    unsigned kaddr[2];  // transposed reads of the K image: rows 4 h + q and + 8 of d-tile 0 (see the forward kernel)
    {
        typedef __attribute__((address_space(3))) char lds_char;
        const int qd = (lane & 15) >> 2, p4 = lane & 3;
        const int col = 16 * ((lane >> 4) & 1) + 4 * p4, row = 4 * (lane >> 5) + qd;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)ring + (col & 7) * 2;
        kaddr[0] = base + lds_off<DUAL>(row, col >> 3);
        kaddr[1] = base + lds_off<DUAL>(row + 8, col >> 3);
    }
    // causal: as in the forward kernel (tiles beyond the workgroup's last query never staged, a wave skips the tiles in the
    // future of all its queries and masks inside the ones its diagonal crosses)
    const int off = p.S - p.T, wt0 = qb * 128 + wave * 32;
    const int ntiles = p.causal ? max(0, min((p.S + KT - 1) / KT, (qb * 128 + 128 + off + KT - 1) / KT)) : (p.S + KT - 1) / KT;
    auto dma = [&](int tile, int st) {
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, kbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, vbytes, 0x00020000);
        char* kd = ring + st * 2 * IMG + wave * 1024;
        const unsigned ks = (unsigned)(tile * KT) * k_rsb, vs = (unsigned)(tile * KT) * v_rsb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(kd + 4096 * i), 16, koff[i], ks, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(kd + IMG + 4096 * i), 16, voff[i], vs, 0, 0);
        }
    };
    const unsigned char* padrow = p.key_pad ? p.key_pad + (long long)b * p.S : nullptr;
    auto pad_of = [&](int tile) -> unsigned {
        const int sk = tile * KT + lane;
        if (sk >= p.S) return 1u;
        return padrow ? (unsigned)padrow[sk] : 0u;
    };
    // the stored keep bits of this query's keys: a dword per 32 keys (rows past T: zeros)
    auto bits_of = [&](int tile, int kb) -> unsigned {
        if (!DROP || !valid) return 0u;
        return *reinterpret_cast<const unsigned*>(p.drop_mask + mrow * p.mask_pitch + ((tile * KT + kb * 32) >> 3));
    };

    auto body = [&](int tile, const char* k_lds, const char* v_lds, unsigned padb, unsigned w0, unsigned w1) {
        const int s0 = tile * KT;
        if (p.causal && s0 > wt0 + 31 + off) return;
        const bool check = p.causal && s0 + KT - 1 > wt0 + off;
        const unsigned long long dead = __ballot(padb != 0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s1, d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) s1[r] = s_init;
#pragma unroll
            for (int kk = 0; kk < NF; ++kk) s1 = mm<T>(row_frag<DUAL>(k_lds, kb * 32, kk, lane), qf[kk], s1);
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[r] = DROP ? 0.f : -dl;  // (with dropout dP is scaled before delta comes off)
#pragma unroll
            for (int kk = 0; kk < NF; ++kk) d1 = mm<T>(row_frag<DUAL>(v_lds, kb * 32, kk, lane), dof[kk], d1);
            // Kᵀ fragments of this block for the dQ product (asm: no vmcnt drain in front of them)
            s16x4 kt[2][ND][2];
            const unsigned ka0 = kaddr[0] + (unsigned)(k_lds - ring), ka8 = kaddr[1] + (unsigned)(k_lds - ring);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    if (kb == 0) {
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(kt[s][dt][0]) : "v"(ka0), "i"(1024 * (2 * s) + 512 * dt));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(kt[s][dt][1]) : "v"(ka8), "i"(1024 * (2 * s) + 512 * dt));
                    } else {
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(kt[s][dt][0]) : "v"(ka0), "i"(1024 * (4 + 2 * s) + 512 * dt));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(kt[s][dt][1]) : "v"(ka8), "i"(1024 * (4 + 2 * s) + 512 * dt));
                    }
                }
            if (dead) {  // wave-uniform: this tile holds masked keys
                unsigned dm = (unsigned)(dead >> (32 * kb)) >> (4 * (lane >> 5));
                asm volatile("; masked keys" : "+v"(dm));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((dm >> ((r & 3) + 8 * (r >> 2))) & 1u) s1[r] = -INFINITY;
            }
            if (check) {  // wave-uniform: the causal boundary crosses this (wave, tile) block
                int kq = t + off - s0 - kb * 32 - 4 * (lane >> 5);  // key index (in the block) > kq is in this query's future
                asm volatile("; causal block" : "+v"(kq));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) > kq) s1[r] = -INFINITY;
            }
            if constexpr (DROP) {
                const unsigned w4 = kb ? w1 : w0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned byte = (w4 >> (8 * g)) & 0xffu;
                    byte >>= 4 * (lane >> 5);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d1[4 * g + j] = ((byte >> j) & 1) ? d1[4 * g + j] * p.drop_scale : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pw = __builtin_amdgcn_exp2f(s1[r] * c);  // masked: exp2(-inf) = 0
                s1[r] = DROP ? pw * (d1[r] - dl) : pw * d1[r];        // dSᵀ
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(kt[0][0][0]), "+v"(kt[0][0][1]), "+v"(kt[0][1][0]), "+v"(kt[0][1][1]),
                           "+v"(kt[1][0][0]), "+v"(kt[1][0][1]), "+v"(kt[1][1][0]), "+v"(kt[1][1][1]));
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t pf = acc_frag<T>(s1, s);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const s16x4 lo = kt[s][dt][0], hi = kt[s][dt][1];
                    const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    acc[dt] = mm<T>(__builtin_bit_cast(bf16x8_t, f), pf, acc[dt]);
                }
            }
        }
    };

    unsigned padb = 0, w0 = 0, w1 = 0;
    if (ntiles > 0) {
        w0 = bits_of(0, 0); w1 = bits_of(0, 1);
        padb = pad_of(0);
        dma(0, 0);
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile (and the bits / pad byte behind them)
        __syncthreads();                                   // everyone's pieces; the other stage is free
        unsigned padn = 0, n0 = 0, n1 = 0;
        if (tile + 1 < ntiles) {
            n0 = bits_of(tile + 1, 0); n1 = bits_of(tile + 1, 1);
            padn = pad_of(tile + 1);
            dma(tile + 1, (tile + 1) & 1);
        }
        const char* st = ring + (tile & 1) * 2 * IMG;
        body(tile, st, st + IMG, padb, w0, w1);
        padb = padn; w0 = n0; w1 = n1;
    }
    store_rowT(dq + b * p.dq_bs + h * HD, p.dq_rs, t, valid, acc, p.scale, lane);
}


// ---- dK / dV for long query sequences: the same frame, key on the lane ----
// One workgroup = 128 keys of a (batch, head) pair (wave w: keys 32 w .. on its lanes, K / V fragments in registers), the
// pair's queries stream by in tiles of 64: Q and dO tiles by LDS-DMA into the two-deep ring of dual-use images (read by rows
// for S = Q Kᵀ and dP = dO Vᵀ, transposed — inline asm, see the forward kernel — for dVᵀ += dOᵀ P and dKᵀ += Qᵀ dS), the
// tile's -lse / scale and -delta rows (the initial accumulators) and, with dropout, its keep dwords through registers into a
// two-deep LDS row next to them; one barrier per tile.  Arithmetic and results of attn_bwd_dkv_kernel (attention.hip); heads of
// 64, no causal mask, no rotation.
template <typename T, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_dkv_long_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, const T* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               T* __restrict__ dk, T* __restrict__ dv, AttnParams p, int nkb,
                                                               int npairs) {
    constexpr int HD = 64, NF = 4, ND = 2;
    constexpr int IMG = img_bytes<DUAL>();
    typedef __attribute__((address_space(3))) void lds_void;
    __shared__ __attribute__((aligned(1024))) char ring[2 * 2 * IMG];  // stage st: Q image at 2 st IMG, dO image behind it
    __shared__ __attribute__((aligned(16))) float rowc[2][2][KT];      // stage st: -lse / scale, -delta of the tile's rows
    __shared__ unsigned m_lds[DROP ? 2 * KT * 4 : 1];                  // stage st: keep dwords [row][wave]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lin = blockIdx.x, grp = (lin >> 3) / nkb, blk = (lin >> 3) % nkb;
    const int pair = grp * 8 + (lin & 7);
    if (pair >= npairs) return;
    const int b = pair / p.H, h = pair % p.H;
    const int s = blk * 128 + wave * 32 + (lane & 31);
    const bool kvalid = s < p.S && !(p.key_pad && p.key_pad[(long long)b * p.S + min(s, p.S - 1)]);
    const float c = p.scale * LOG2E, inv_scale = 1.f / p.scale;
    const unsigned keep_bit = s < p.S ? 1u << (lane & 31) : 0u;
    const long long row0 = ((long long)b * p.H + h) * p.T;

    bf16x8_t kf[NF], vf[NF];
    load_row_frags(kf, k + b * p.k_bs + h * HD, p.k_rs, s, s < p.S, lane);
    load_row_frags(vf, v + b * p.v_bs + h * HD, p.v_rs, s, s < p.S, lane);
    f32x16 dka[ND], dva[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dka[dt][r] = dva[dt][r] = 0.f;

    const T* qbase = q + b * p.q_bs + h * HD;
    const T* dobase = d_o + b * p.do_bs + h * HD;
    const unsigned q_rsb = (unsigned)(p.q_rs * 2), do_rsb = (unsigned)(p.do_rs * 2);
    const int qbytes = (int)(((long long)(p.T - 1) * p.q_rs + HD) * 2), dobytes = (int)(((long long)(p.T - 1) * p.do_rs + HD) * 2);
    unsigned qoff[2], dooff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + 4 * i) + ((lane >> 2) & 7);
        const int ch = 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));  // (the inverse of lds_off<DUAL>)
        qoff[i] = (unsigned)row * q_rsb + 16 * ch;
        dooff[i] = (unsigned)row * do_rsb + 16 * ch;
    }
    unsigned raddr[2];  // row reads of an image (row_frag<DUAL>): this lane's row, chunks 2 kk + h for even / odd kk (kk >> 1 adds 512)
    {
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)ring;
        raddr[0] = base + lds_off<DUAL>(lane & 31, lane >> 5);
        raddr[1] = base + lds_off<DUAL>(lane & 31, 2 + (lane >> 5));
    }
    unsigned taddr[2];  // transposed reads of an image: rows 4 h + q and + 8 of d-tile 0 (see the forward kernel)
    {
        typedef __attribute__((address_space(3))) char lds_char;
        const int qd = (lane & 15) >> 2, p4 = lane & 3;
        const int col = 16 * ((lane >> 4) & 1) + 4 * p4, row = 4 * (lane >> 5) + qd;
        const unsigned base = (unsigned)(unsigned long)(lds_char*)ring + (col & 7) * 2;
        taddr[0] = base + lds_off<DUAL>(row, col >> 3);
        taddr[1] = base + lds_off<DUAL>(row + 8, col >> 3);
    }
    // causal (query t sees keys <= t + S - T): the query tiles before the workgroup's first key are never staged; a wave skips
    // the tiles whose queries all lie before its keys and masks inside the ones its diagonal crosses (both wave-uniform)
    const int off = p.S - p.T, ws0 = blk * 128 + wave * 32;
    const int ntiles = (p.T + KT - 1) / KT;
    const int tile0 = p.causal ? max(0, blk * 128 - off) / KT : 0;
    auto dma = [&](int tile, int st) {
        __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, qbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dobase, 0, dobytes, 0x00020000);
        char* qd = ring + st * 2 * IMG + wave * 1024;
        const unsigned qs = (unsigned)(tile * KT) * q_rsb, ds = (unsigned)(tile * KT) * do_rsb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void*)(qd + 4096 * i), 16, qoff[i], qs, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void*)(qd + IMG + 4096 * i), 16, dooff[i], ds, 0, 0);
        }
    };
    // thread tid < 64: the row constants of row tid of a tile; every thread: dword tid & 3 of the keep bits of row tid >> 2
    // (round 5: the loads are RAW and unconditional — rows past T read row T - 1 — and the sign, the scale and the select are
    // applied where the registers are consumed, a tile later.  Written `l_n = t < T ? -lse[..] * inv_scale : -inf`, hipcc
    // waited for each load where the product stood: two serial trips to memory per tile on wave 0, in front of the DMA request,
    // with the other three waves waiting at the next barrier.)
    float l_n = 0.f, d_n = 0.f;
    unsigned m_n = 0u;
    auto rows_g2r = [&](int tile) {
        const int t0 = tile * KT;
        if (tid < KT) {
            const long long t = row0 + min(t0 + tid, p.T - 1);
            l_n = lse[t];
            d_n = delta[t];
        }
        if constexpr (DROP) {
            const int t = t0 + (tid >> 2);
            const long long byte0 = (long long)blk * 16 + (tid & 3) * 4;
            const bool okw = t < p.T && byte0 + 4 <= p.mask_pitch;
            m_n = *reinterpret_cast<const unsigned*>(p.drop_mask + (row0 + (okw ? t : 0)) * p.mask_pitch + (okw ? byte0 : 0));
        }
    };
    // the registers of `rows_g2r(tile)` into stage st of the LDS rows
    auto rows_r2s = [&](int tile, int st) {
        const int t0 = tile * KT;
        if (tid < KT) {
            const bool in = t0 + tid < p.T;
            rowc[st][0][tid] = in ? -l_n * inv_scale : -INFINITY;  // -inf -> p = 0 for rows past T
            rowc[st][1][tid] = in ? -d_n : 0.f;
        }
        if constexpr (DROP) {
            const int t = t0 + (tid >> 2);
            const long long byte0 = (long long)blk * 16 + (tid & 3) * 4;
            m_lds[st * KT * 4 + tid] = (t < p.T && byte0 + 4 <= p.mask_pitch) ? m_n : 0u;
        }
    };

    auto body = [&](int tile, int st) {
        const char* q_lds = ring + st * 2 * IMG;
        const char* do_lds = q_lds + IMG;
        const int t0 = tile * KT;
        if (p.causal && ws0 > t0 + KT - 1 + off) return;       // every key of this wave is in the future of every query
        const bool check = p.causal && ws0 + 31 > t0 + off;   // some pairs are masked
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // Row fragments of the Q and dO images in inline asm as well (round 5): hipcc's wait-count pass drains vmcnt before a
            // `ds_read` that may alias an LDS-DMA in flight — here the NEXT tile's pieces, requested a few instructions earlier:
            // the prefetch was waited for in front of the first fragment read of every tile (an asm read carries no memory
            // operand; the forward and dQ kernels escape the drain only because theirs lose it in a branch fold).
            bf16x8_t qr[NF], dr[NF];
            {
                const unsigned r0 = raddr[0] + (unsigned)(q_lds - ring), r1 = raddr[1] + (unsigned)(q_lds - ring);
#pragma unroll
                for (int kk = 0; kk < NF; ++kk) {
                    if (kk & 1) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qr[kk]) : "v"(r1), "i"(4096 * qb + 512 * (kk >> 1)));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dr[kk]) : "v"(r1), "i"(8192 + 4096 * qb + 512 * (kk >> 1)));
                    } else {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qr[kk]) : "v"(r0), "i"(4096 * qb + 512 * (kk >> 1)));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dr[kk]) : "v"(r0), "i"(8192 + 4096 * qb + 512 * (kk >> 1)));
                    }
                }
            }
            f32x16 sc, dp;
            float4 d4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are 4 consecutive queries: one 16-B read each of -lse, -delta
                const int tl = qb * 32 + 8 * g + 4 * (lane >> 5);
                const float4 l4 = *reinterpret_cast<const float4*>(&rowc[st][0][tl]);
                d4[g] = *reinterpret_cast<const float4*>(&rowc[st][1][tl]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sc[4 * g + j] = (&l4.x)[j];
                    dp[4 * g + j] = DROP ? 0.f : (&d4[g].x)[j];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(qr[0]), "+v"(qr[1]), "+v"(qr[2]), "+v"(qr[3]), "+v"(dr[0]), "+v"(dr[1]), "+v"(dr[2]), "+v"(dr[3]));
#pragma unroll
            for (int kk = 0; kk < NF; ++kk) {
                sc = mm<T>(qr[kk], kf[kk], sc);
                dp = mm<T>(dr[kk], vf[kk], dp);
            }
            if (check) {  // wave-uniform: the causal boundary crosses this (tile, wave) block
                int sq = s - off - t0 - qb * 32 - 4 * (lane >> 5);  // key s is visible to query index (in the block) >= sq
                asm volatile("; causal block" : "+v"(sq));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) < sq) sc[r] = -INFINITY;
            }
            s16x4 tq[2][ND][2], td[2][ND][2];  // Qᵀ / dOᵀ fragments of this query block: [k-step][d-tile][rows +0 / +8]
            const unsigned a0 = taddr[0] + (unsigned)(q_lds - ring), a8 = taddr[1] + (unsigned)(q_lds - ring);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(td[s2][dt][0]) : "v"(a0), "i"(8192 + 1024 * (4 * qb + 2 * s2) + 512 * dt));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(td[s2][dt][1]) : "v"(a8), "i"(8192 + 1024 * (4 * qb + 2 * s2) + 512 * dt));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tq[s2][dt][0]) : "v"(a0), "i"(1024 * (4 * qb + 2 * s2) + 512 * dt));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tq[s2][dt][1]) : "v"(a8), "i"(1024 * (4 * qb + 2 * s2) + 512 * dt));
                }
            f32x16 ds;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pw = __builtin_amdgcn_exp2f(sc[4 * g + j] * c);
                    if constexpr (DROP) {  // keep bit of (query tl + j, this lane's key): bit lane & 31 of the wave's dword
                        const int tl = qb * 32 + 8 * g + 4 * (lane >> 5);
                        const unsigned wd = m_lds[st * KT * 4 + (tl + j) * 4 + wave];
                        const float km = (wd & keep_bit) ? p.drop_scale : 0.f;  // (rows past T were staged as zeros)
                        sc[4 * g + j] = pw * km;
                        ds[4 * g + j] = pw * fmaf(dp[4 * g + j], km, (&d4[g].x)[j]);
                    } else {
                        sc[4 * g + j] = pw;
                        ds[4 * g + j] = pw * dp[4 * g + j];
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(tq[0][0][0]), "+v"(tq[0][0][1]), "+v"(tq[0][1][0]), "+v"(tq[0][1][1]),
                           "+v"(tq[1][0][0]), "+v"(tq[1][0][1]), "+v"(tq[1][1][0]), "+v"(tq[1][1][1]),
                           "+v"(td[0][0][0]), "+v"(td[0][0][1]), "+v"(td[0][1][0]), "+v"(td[0][1][1]),
                           "+v"(td[1][0][0]), "+v"(td[1][0][1]), "+v"(td[1][1][0]), "+v"(td[1][1][1]));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t pf = acc_frag<T>(sc, s2), dsf = acc_frag<T>(ds, s2);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    // dVᵀ[d][key] += dOᵀ[d][query] · P[query][key] ;  dKᵀ[d][key] += Qᵀ[d][query] · dS[query][key]
                    const s16x8 fd = {td[s2][dt][0][0], td[s2][dt][0][1], td[s2][dt][0][2], td[s2][dt][0][3],
                                      td[s2][dt][1][0], td[s2][dt][1][1], td[s2][dt][1][2], td[s2][dt][1][3]};
                    const s16x8 fq = {tq[s2][dt][0][0], tq[s2][dt][0][1], tq[s2][dt][0][2], tq[s2][dt][0][3],
                                      tq[s2][dt][1][0], tq[s2][dt][1][1], tq[s2][dt][1][2], tq[s2][dt][1][3]};
                    dva[dt] = mm<T>(__builtin_bit_cast(bf16x8_t, fd), pf, dva[dt]);
                    dka[dt] = mm<T>(__builtin_bit_cast(bf16x8_t, fq), dsf, dka[dt]);
                }
            }
        }
    };

    if (tile0 < ntiles) {
        rows_g2r(tile0);
        dma(tile0, tile0 & 1);
    }
    for (int tile = tile0; tile < ntiles; ++tile) {
        const int st = tile & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile and the row constants behind them
        rows_r2s(tile, st);
        __syncthreads();                                   // everyone's pieces and rows; the other stage is free
        if (tile + 1 < ntiles) {
            rows_g2r(tile + 1);
            dma(tile + 1, st ^ 1);
        }
        body(tile, st);
    }
    if (!kvalid) {  // padding keys: zero rows (a select, not a product: their probabilities are unbounded)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dka[dt][r] = dva[dt][r] = 0.f;
    }
    store_rowT(dk + b * p.dk_bs + h * HD, p.dk_rs, s, s < p.S, dka, p.scale, lane);
    store_rowT(dv + b * p.dv_bs + h * HD, p.dv_rs, s, s < p.S, dva, 1.f, lane);
}

}  // namespace

int pk_attn_fwd_long_launch(const void* q, const void* k, const void* v, void* o, float* lse, const pkattn::AttnParams& p, int dtype,
                            hipStream_t s) {
    const int nqb = (p.T + 127) / 128, npairs = p.B * p.H;
    const dim3 lg((unsigned)((npairs + 7) / 8 * 8) * nqb);
#define PK_LONG(TT, DR) hipLaunchKernelGGL((attn_fwd_long_kernel<TT, DR>), lg, dim3(256), 0, s, (const TT*)q, (const TT*)k, \
                                           (const TT*)v, (TT*)o, lse, p, nqb, npairs)
    if (dtype == PK_BF16) { if (p.drop_thr) PK_LONG(bf16, true); else PK_LONG(bf16, false); }
    else { if (p.drop_thr) PK_LONG(f16, true); else PK_LONG(f16, false); }
#undef PK_LONG
    return (int)hipGetLastError();
}

int pk_attn_dq_long_launch(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                           float* delta, void* dq, const pkattn::AttnParams& p, int dtype, hipStream_t s) {
    const int nqb = (p.T + 127) / 128, npairs = p.B * p.H;
    const dim3 lg((unsigned)((npairs + 7) / 8 * 8) * nqb);
#define PK_LONG(TT, DR) hipLaunchKernelGGL((attn_dq_long_kernel<TT, DR>), lg, dim3(256), 0, s, (const TT*)q, (const TT*)k, \
                                           (const TT*)v, (const TT*)o, (const TT*)d_o, lse, delta, (TT*)dq, p, nqb, npairs)
    if (dtype == PK_BF16) { if (p.drop_thr) PK_LONG(bf16, true); else PK_LONG(bf16, false); }
    else { if (p.drop_thr) PK_LONG(f16, true); else PK_LONG(f16, false); }
#undef PK_LONG
    return (int)hipGetLastError();
}

int pk_attn_dkv_long_launch(const void* q, const void* k, const void* v, const void* d_o, const float* lse, const float* delta,
                            void* dk, void* dv, const pkattn::AttnParams& p, int dtype, hipStream_t s) {
    const int nkb = (p.S + 127) / 128, npairs = p.B * p.H;
    const dim3 lg((unsigned)((npairs + 7) / 8 * 8) * nkb);
#define PK_LONG(TT, DR) hipLaunchKernelGGL((attn_dkv_long_kernel<TT, DR>), lg, dim3(256), 0, s, (const TT*)q, (const TT*)k, \
                                           (const TT*)v, (const TT*)d_o, lse, delta, (TT*)dk, (TT*)dv, p, nkb, npairs)
    if (dtype == PK_BF16) { if (p.drop_thr) PK_LONG(bf16, true); else PK_LONG(bf16, false); }
    else { if (p.drop_thr) PK_LONG(f16, true); else PK_LONG(f16, false); }
#undef PK_LONG
    return (int)hipGetLastError();
}
