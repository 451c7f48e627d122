// Scaled-dot-product attention, forward + backward, head_dim 64 or 128 (K3 of SURVEY §2c).
//   reference: F.scaled_dot_product_attention(q, k, v, attn_mask, is_causal, scale)  pasero/models/modules.py:707-720
//              mask assembly (bool (B,S) key-padding -> -inf, causal triu)            pasero/models/modules.py:654-677
//              custom fallback with fp32 softmax + nan_to_num                         pasero/models/modules.py:742-771
// q (B,T,H,64), k/v (B,S,H,64) are read in place from the projection outputs (row strides given), o is written in
// the (B,T,H,64) layout out_proj consumes: no transposes, no mask tensor, no (B,H,T,S) score matrix.
// Masks: key_pad[b][s] != 0 -> key masked (the bool tensor of the reference is passed as is); causal -> query t sees
// keys s <= t + (S - T).  A query whose keys are all masked outputs 0 (the reference's nan_to_num behaviour).
//
// bf16 path (MFMA 32x32x16, flash-style, online softmax in fp32, exp2 domain):
//   fwd / bwd_dq : one wave owns 32 queries, the QUERY sits on the MFMA lane (Sᵀ = K·Qᵀ), so row max / sum / lse are
//                  per-lane scalars and the probability tile feeds the next MFMA as its B operand straight from the
//                  accumulator registers (no LDS round trip); K rows are read with ds_read_b128, Vᵀ / Kᵀ operands with
//                  ds_read_b64_tr_b16 from the same row-major LDS tile.
//   bwd_dkv      : one wave owns 32 keys, the KEY sits on the lane (S = Q·Kᵀ, dP = dO·Vᵀ), so P and dS feed
//                  dVᵀ += dOᵀ·P and dKᵀ += Qᵀ·dS from registers and dK/dV need no cross-workgroup sum (no atomics).
// fp32 path: one thread per query (or key) row with broadcast LDS reads — exact fp32 arithmetic for parity runs.
#include "attention_common.h"

namespace {

// =====================================================================================================
// fp32 path
// =====================================================================================================
constexpr int F32_TILE = 32;

template <int HD>
__global__ __launch_bounds__(128) void attn_fwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ o,
                                                           float* __restrict__ lse, AttnParams p) {
    __shared__ float ks[F32_TILE][HD], vs[F32_TILE][HD];
    const int b = blockIdx.z, h = blockIdx.y, t = blockIdx.x * 128 + threadIdx.x;
    const bool valid = t < p.T;
    float qr[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        qr[d] = valid ? q[b * p.q_bs + (long long)t * p.q_rs + h * HD + d] : 0.f;
        acc[d] = 0.f;
    }
    if (p.rope_cos) rope_row_f32<HD, false>(qr, p, p.rope_q0 + t);
    float m = -INFINITY, l = 0.f;
    const float c = p.scale * LOG2E;
    const long long row = ((long long)b * p.H + h) * p.T + t;
    unsigned mbyte = 0;
    for (int s0 = 0; s0 < p.S; s0 += F32_TILE) {
        __syncthreads();
        for (int i = threadIdx.x; i < F32_TILE * HD; i += 128) {
            int r = i / HD, d = i % HD, s = s0 + r;
            const float* krow = k + b * p.k_bs + (long long)s * p.k_rs + h * HD;
            ks[r][d] = s < p.S ? (p.rope_cos ? rope_elem_f32<HD>(krow, d, p, p.rope_k0 + s) : krow[d]) : 0.f;
            vs[r][d] = s < p.S ? v[b * p.v_bs + (long long)s * p.v_rs + h * HD + d] : 0.f;
        }
        __syncthreads();
        for (int r = 0; r < F32_TILE; ++r) {
            int s = s0 + r;
            if (!valid || s >= p.S) continue;
            bool keep = true;
            if (p.drop_thr) {  // keep bits of all S keys of this query, eight per byte
                keep = drop_keep1(p, row, s);
                mbyte |= (unsigned)keep << (s & 7);
                if ((s & 7) == 7 || s == p.S - 1) {
                    p.drop_mask[row * p.mask_pitch + (s >> 3)] = (unsigned char)mbyte;
                    mbyte = 0;
                }
            }
            if (key_masked(p, b, t, s)) continue;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dot += qr[d] * ks[r][d];
            float s2 = dot * c;
            float mn = fmaxf(m, s2);
            float alpha = exp2f(m - mn), pw = exp2f(s2 - mn);
            l = l * alpha + pw;  // the softmax denominator does not see the dropout
            const float pv = keep ? pw : 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = acc[d] * alpha + pv * vs[r][d];
            m = mn;
        }
    }
    if (!valid) return;
    float inv = (l > 0.f ? 1.f / l : 0.f) * (p.drop_thr ? p.drop_scale : 1.f);
#pragma unroll
    for (int d = 0; d < HD; ++d) o[b * p.o_bs + (long long)t * p.o_rs + h * HD + d] = acc[d] * inv;
    lse[row] = l > 0.f ? (m + log2f(l)) * LN2 : 0.f;
}

// dQ (thread per query) + delta = rowsum(dO * O)
template <int HD>
__global__ __launch_bounds__(128) void attn_bwd_dq_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ o,
                                                              const float* __restrict__ d_o,
                                                              const float* __restrict__ lse, float* __restrict__ delta,
                                                              float* __restrict__ dq, AttnParams p) {
    __shared__ float ks[F32_TILE][HD], vs[F32_TILE][HD];
    const int b = blockIdx.z, h = blockIdx.y, t = blockIdx.x * 128 + threadIdx.x;
    const bool valid = t < p.T;
    float qr[HD], dor[HD], acc[HD];
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        qr[d] = valid ? q[b * p.q_bs + (long long)t * p.q_rs + h * HD + d] : 0.f;
        dor[d] = valid ? d_o[b * p.do_bs + (long long)t * p.do_rs + h * HD + d] : 0.f;
        float ov = valid ? o[b * p.o_bs + (long long)t * p.o_rs + h * HD + d] : 0.f;
        dl += dor[d] * ov;
        acc[d] = 0.f;
    }
    if (p.rope_cos) rope_row_f32<HD, false>(qr, p, p.rope_q0 + t);
    const long long row = ((long long)b * p.H + h) * p.T + t;
    const float L2 = valid ? lse[row] * LOG2E : 0.f;
    if (valid) delta[row] = dl;
    const float c = p.scale * LOG2E;
    for (int s0 = 0; s0 < p.S; s0 += F32_TILE) {
        __syncthreads();
        for (int i = threadIdx.x; i < F32_TILE * HD; i += 128) {
            int r = i / HD, d = i % HD, s = s0 + r;
            const float* krow = k + b * p.k_bs + (long long)s * p.k_rs + h * HD;
            ks[r][d] = s < p.S ? (p.rope_cos ? rope_elem_f32<HD>(krow, d, p, p.rope_k0 + s) : krow[d]) : 0.f;
            vs[r][d] = s < p.S ? v[b * p.v_bs + (long long)s * p.v_rs + h * HD + d] : 0.f;
        }
        __syncthreads();
        for (int r = 0; r < F32_TILE; ++r) {
            int s = s0 + r;
            if (!valid || key_masked(p, b, t, s)) continue;
            float dot = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dot += qr[d] * ks[r][d];
                dp += dor[d] * vs[r][d];
            }
            float pw = exp2f(dot * c - L2);
            if (p.drop_thr) dp = drop_bit(p, row, s) ? dp * p.drop_scale : 0.f;  // dP = M/(1-p) * (dO . V)
            float ds = pw * (dp - dl) * p.scale;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] += ds * ks[r][d];
        }
    }
    if (!valid) return;
    if (p.rope_cos) rope_row_f32<HD, true>(acc, p, p.rope_q0 + t);  // (the gradient of the UNROTATED query)
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[b * p.dq_bs + (long long)t * p.dq_rs + h * HD + d] = acc[d];
}

// dK, dV (thread per key)
template <int HD>
__global__ __launch_bounds__(128) void attn_bwd_dkv_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v,
                                                               const float* __restrict__ d_o,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta, float* __restrict__ dk,
                                                               float* __restrict__ dv, AttnParams p) {
    __shared__ float qs[F32_TILE][HD], dos[F32_TILE][HD], ls[F32_TILE], dls[F32_TILE];
    const int b = blockIdx.z, h = blockIdx.y, s = blockIdx.x * 128 + threadIdx.x;
    const bool valid = s < p.S;
    float kr[HD], vr[HD], dka[HD], dva[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        kr[d] = valid ? k[b * p.k_bs + (long long)s * p.k_rs + h * HD + d] : 0.f;
        vr[d] = valid ? v[b * p.v_bs + (long long)s * p.v_rs + h * HD + d] : 0.f;
        dka[d] = 0.f;
        dva[d] = 0.f;
    }
    if (p.rope_cos) rope_row_f32<HD, false>(kr, p, p.rope_k0 + s);
    const float c = p.scale * LOG2E;
    for (int t0 = 0; t0 < p.T; t0 += F32_TILE) {
        __syncthreads();
        for (int i = threadIdx.x; i < F32_TILE * HD; i += 128) {
            int r = i / HD, d = i % HD, t = t0 + r;
            const float* qrow = q + b * p.q_bs + (long long)t * p.q_rs + h * HD;
            qs[r][d] = t < p.T ? (p.rope_cos ? rope_elem_f32<HD>(qrow, d, p, p.rope_q0 + t) : qrow[d]) : 0.f;
            dos[r][d] = t < p.T ? d_o[b * p.do_bs + (long long)t * p.do_rs + h * HD + d] : 0.f;
        }
        if (threadIdx.x < F32_TILE) {
            int t = t0 + threadIdx.x;
            long long row = ((long long)b * p.H + h) * p.T + t;
            ls[threadIdx.x] = t < p.T ? lse[row] * LOG2E : 0.f;
            dls[threadIdx.x] = t < p.T ? delta[row] : 0.f;
        }
        __syncthreads();
        for (int r = 0; r < F32_TILE; ++r) {
            int t = t0 + r;
            if (!valid || t >= p.T || key_masked(p, b, t, s)) continue;
            float dot = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dot += qs[r][d] * kr[d];
                dp += dos[r][d] * vr[d];
            }
            float pw = exp2f(dot * c - ls[r]);
            float pv = pw;
            if (p.drop_thr) {
                const bool keep = drop_bit(p, ((long long)b * p.H + h) * p.T + t, s);
                pv = keep ? pw * p.drop_scale : 0.f;
                dp = keep ? dp * p.drop_scale : 0.f;
            }
            float ds = pw * (dp - dls[r]) * p.scale;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dva[d] += pv * dos[r][d];
                dka[d] += ds * qs[r][d];
            }
        }
    }
    if (!valid) return;
    if (p.rope_cos) rope_row_f32<HD, true>(dka, p, p.rope_k0 + s);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        dk[b * p.dk_bs + (long long)s * p.dk_rs + h * HD + d] = dka[d];
        dv[b * p.dv_bs + (long long)s * p.dv_rs + h * HD + d] = dva[d];
    }
}

// ---- forward (MODE 0) and dQ backward (MODE 1): query on the lane ----
template <typename T, int MODE, int HD, bool DROP>
__global__ __launch_bounds__(256, q_min_waves(MODE, HD)) void attn_q_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                     const T* __restrict__ v, T* __restrict__ o,
                                                     const T* __restrict__ d_o, float* __restrict__ lse,
                                                     float* __restrict__ delta, T* __restrict__ dq, AttnParams p) {
    // forward: K by rows only, V transposed only.  dQ backward: K by rows AND transposed (dual image), V by rows only
    constexpr int KP = MODE == 0 ? PITCH : DUAL, VP = MODE == 0 ? VPITCH : PITCH;
    constexpr int NF = HD / 16, ND = HD / 32, NI = HD / 64;  // k-steps per row, 32-row d-tiles, 64-column LDS images
    __shared__ __attribute__((aligned(16))) char k_lds[NI * img_bytes<KP>()];
    __shared__ __attribute__((aligned(16))) char v_lds[NI * img_bytes<VP>()];
    __shared__ unsigned long long dead_lds;  // bit i: key i of the staged tile is past S or a padding key
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, the blocks of one (batch, head) pair on ONE XCD (pair_block): they re-read the pair's K / V from its L2
    int b, h, blk;
    if (!pair_block(p, p.nqb, b, h, blk)) return;
    const int t = blk * 128 + wave * 32 + (lane & 31);
    const bool valid = t < p.T;
    const float c = p.scale * LOG2E, inv_c = 1.f / c;

    bf16x8_t qf[NF], dof[NF];
    load_row_frags(qf, q + b * p.q_bs + h * HD, p.q_rs, t, valid, lane);
    if (p.rope_cos) rope_frags<NF, T>(qf, p, p.rope_q0 + t, lane);
    // In the dQ pass the score MFMAs start from the accumulator -lse / scale and leave s - lse / scale, dP = V dOᵀ starts from
    // -delta: p = exp2(c acc), dS = p dP' — a multiply, the exp and a multiply per score; masked keys and the causal boundary
    // cost selects only in the tiles that hold them (wave-uniform branches).  (Q itself stays unscaled: folding c into its
    // 16-bit fragments saves the multiply and costs 1.7e-3 on lse — tests/test_fullsize_gpu.py holds 1e-3.)
    float m = -INFINITY, l = 0.f, L2 = 0.f, dl = 0.f;
    f32x16 acc[ND];  // Oᵀ (MODE 0) or dQᵀ (MODE 1): [d-tile][d rows] x query lane
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    if constexpr (MODE == 1) {
        load_row_frags(dof, d_o + b * p.do_bs + h * HD, p.do_rs, t, valid, lane);
        bf16x8_t of[NF];
        load_row_frags(of, o + b * p.o_bs + h * HD, p.o_rs, t, valid, lane);
#pragma unroll
        for (int kk = 0; kk < NF; ++kk) dl += frag_dot<T>(dof[kk], of[kk]);
        dl += __shfl_xor(dl, 32, 64);
        const long long row = ((long long)b * p.H + h) * p.T + t;
        if (valid) {
            L2 = lse[row] * LOG2E;
            if (lane < 32) delta[row] = dl;
        }
    }

    if (MODE == 1 && !valid) L2 = INFINITY;  // exp2(x - inf) = 0: rows past T contribute nothing
    // causal: keys beyond the last query of this workgroup are never visible
    const int off = p.S - p.T;
    int s_end = p.S;
    if (p.causal) s_end = min(p.S, blk * 128 + 128 + off);
    const int wt0 = blk * 128 + wave * 32;  // first query of this wave
    const T* kbase = k + b * p.k_bs + h * HD;
    const T* vbase = v + b * p.v_bs + h * HD;
    uint4 kreg[2 * NI], vreg[2 * NI];
    if (s_end > 0) {
        tile_g2r(kreg, kbase, p.k_rs, 0, p.S, tid);
        tile_g2r(vreg, vbase, p.v_rs, 0, p.S, tid);
        if (p.rope_cos) rope_tile<2 * NI, T>(kreg, p, p.rope_k0, tid);
    }
    for (int s0 = 0; s0 < s_end; s0 += KT) {
        __syncthreads();  // previous tile fully consumed
        tile_r2s<KP>(kreg, k_lds, tid);
        tile_r2s<VP>(vreg, v_lds, tid);
        if (tid < KT) {  // keys past S / padding keys (modules.py:654-677): wave 0 is exactly the tile's 64 keys
            const int sk = s0 + tid;
            const bool dead = sk >= p.S || (p.key_pad && p.key_pad[(long long)b * p.S + sk]);
            const unsigned long long dm = __ballot(dead);
            if (tid == 0) dead_lds = dm;
        }
        if (s0 + KT < s_end) {  // prefetch the next tile into registers
            tile_g2r(kreg, kbase, p.k_rs, s0 + KT, p.S, tid);
            tile_g2r(vreg, vbase, p.v_rs, s0 + KT, p.S, tid);
            if (p.rope_cos) rope_tile<2 * NI, T>(kreg, p, p.rope_k0 + s0 + KT, tid);
        }
        __syncthreads();
        // causal classification of (this wave's 32 queries) x (this tile's 64 keys): wave-uniform
        const bool skip = p.causal && s0 > wt0 + 31 + off;        // every key is in the future of every query
        if (skip) continue;
        const bool check = p.causal && s0 + KT - 1 > wt0 + off;   // some (query, key) pairs are masked

        const unsigned long long dead = dead_lds;
        if constexpr (MODE == 1) {
            // dQ pass: one 32-key block at a time — scores, dP, dS and the dQ product of a block before the next one starts (the
            // live set is one block's S and dP: the kernel fits three waves per SIMD; both blocks at once took 232 registers)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s1, d1;
#pragma unroll
                for (int r = 0; r < 16; ++r) s1[r] = -L2 * inv_c;
#pragma unroll
                for (int kk = 0; kk < NF; ++kk) s1 = mm<T>(row_frag<KP>(k_lds, kb * 32, kk, lane), qf[kk], s1);
#pragma unroll
                for (int r = 0; r < 16; ++r) s1[r] *= c;
                if (dead) {  // wave-uniform: this tile holds masked keys
                    unsigned dm = (unsigned)(dead >> (32 * kb)) >> (4 * (lane >> 5));
                    asm volatile("; masked keys" : "+v"(dm));
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((dm >> ((r & 3) + 8 * (r >> 2))) & 1u) s1[r] = -INFINITY;
                }
                if (check) {  // wave-uniform: the causal boundary crosses this (wave, tile) block
                    int kq = t + off - s0 - kb * 32 - 4 * (lane >> 5);
                    asm volatile("; causal block" : "+v"(kq));
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r & 3) + 8 * (r >> 2) > kq) s1[r] = -INFINITY;
                }
                // dPᵀ[key][query] = V[key][:] · dO[query][:]
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = DROP ? 0.f : -dl;  // (with dropout dP is scaled before delta comes off)
#pragma unroll
                for (int kk = 0; kk < NF; ++kk) d1 = mm<T>(row_frag<VP>(v_lds, kb * 32, kk, lane), dof[kk], d1);
                if constexpr (DROP) {  // dP = M / (1 - p) * (dO . V): the stored keep bits of this query's keys
                    const long long mrow = ((long long)b * p.H + h) * p.T + t;
                    const unsigned w4 = valid ? *reinterpret_cast<const unsigned*>(p.drop_mask + mrow * p.mask_pitch + ((s0 + kb * 32) >> 3)) : 0u;
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
                    const float pw = __builtin_amdgcn_exp2f(s1[r]);          // masked: exp2(-inf) = 0
                    s1[r] = DROP ? pw * (d1[r] - dl) : pw * d1[r];           // dSᵀ
                }
                // dQᵀ[d][query] += Kᵀ[d][key] · dSᵀ[key][query]
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8_t pf = acc_frag<T>(s1, s);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) acc[dt] = mm<T>(tr_frag<KP>(k_lds, kb * 32, s, dt * 32, lane), pf, acc[dt]);
                }
            }
            continue;
        }
        f32x16 sc[2];  // Sᵀ[key][query] for the two 32-key blocks of the tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NF; ++kk)
                sc[kb] = mm<T>(row_frag<KP>(k_lds, kb * 32, kk, lane), qf[kk],
                                                                 sc[kb]);
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kb][r] *= c;
        }
        if (dead) {  // wave-uniform: this tile holds masked keys
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                unsigned dm = (unsigned)(dead >> (32 * kb)) >> (4 * (lane >> 5));
                asm volatile("; masked keys" : "+v"(dm));  // (a real branch: if-converted, the selects cost 3 instructions per score in every tile)
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
        float tmax = -INFINITY;
        if constexpr (MODE == 0) {
            tmax = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sc[0][r]), sc[1][r]);
        }
        if constexpr (MODE == 0) {
            tmax = half_wave_max(tmax);
            const float mn = fmaxf(m, tmax);
            const float ms = mn == -INFINITY ? 0.f : mn;  // all keys masked so far: exp2(-inf - 0) = 0 everywhere
            const float alpha = __builtin_amdgcn_exp2f(m - ms);
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pw = __builtin_amdgcn_exp2f(sc[kb][r] - ms);
                    sc[kb][r] = pw;
                    psum += pw;
                }
            l = l * alpha + psum;  // the softmax denominator does not see the dropout
            m = mn;
            if constexpr (DROP) {
                // registers 4g..4g+3 of a block are 4 consecutive keys of this lane's query: one Philox draw each; the
                // two half-waves (keys +0..3 / +4..7) merge their nibbles into the byte of the stored bit mask
                const long long mrow = ((long long)b * p.H + h) * p.T + t;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    unsigned word = 0;
                    // One Philox4x32-10 draw covers 8 consecutive keys of a query row: words x, y serve keys +0..3 (lanes 0..31),
                    // z, w keys +4..7 (lanes 32..63).  Each half-wave evaluates the draw of ONE of the two 8-key groups of a pair
                    // and hands the partner the half it needs — half the evaluations (the forward kernel with dropout was bound
                    // by them: 151 against 71 us at the IWSLT recipe's encoder shape); the bits are those of dropout_keep4.
                    const unsigned thr16 = p.drop_thr >> 16;
                    const bool hi = lane >= 32;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int g_own = 2 * gp + (hi ? 1 : 0);
                        const unsigned long long ctr = ((unsigned long long)mrow * 8ull * p.mask_pitch + (s0 + kb * 32 + 8 * g_own)) >> 3;
                        Philox4 r = philox4x32_10(p.seed, p.offset, ctr);
                        // the lower half-wave keeps (x, y) = its 4 keys of group 2 gp and needs the upper one's (x, y) for group 2 gp + 1; the
                        // upper one keeps (z, w) and needs the lower one's: one swap per register pair
                        half_wave_swap(r.x, r.z);
                        half_wave_swap(r.y, r.w);
                        const unsigned wv[2][2] = {{r.x, r.y}, {r.z, r.w}};
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
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            // Oᵀ[d][query] += Vᵀ[d][key] · P[key][query]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8_t pf = acc_frag<T>(sc[kb], s);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
                        acc[dt] = mm<T>(tr_frag<VP>(v_lds, kb * 32, s, dt * 32, lane),
                                                                          pf, acc[dt]);
                }
        } else {
            // dPᵀ[key][query] = V[key][:] · dO[query][:]
            f32x16 dp[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[kb][r] = DROP ? 0.f : -dl;  // (with dropout dP is scaled before delta comes off)
#pragma unroll
                for (int kk = 0; kk < NF; ++kk)
                    dp[kb] = mm<T>(row_frag<VP>(v_lds, kb * 32, kk, lane), dof[kk],
                                                                     dp[kb]);
                if constexpr (DROP) {  // dP = M / (1 - p) * (dO . V): the stored keep bits of this query's keys
                    const long long mrow = ((long long)b * p.H + h) * p.T + t;
                    const unsigned w4 = valid ? *reinterpret_cast<const unsigned*>(p.drop_mask + mrow * p.mask_pitch + ((s0 + kb * 32) >> 3)) : 0u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned byte = (w4 >> (8 * g)) & 0xffu;
                        byte >>= 4 * (lane >> 5);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            dp[kb][4 * g + j] = ((byte >> j) & 1) ? dp[kb][4 * g + j] * p.drop_scale : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pw = __builtin_amdgcn_exp2f(sc[kb][r]);         // masked: exp2(-inf) = 0
                    sc[kb][r] = DROP ? pw * (dp[kb][r] - dl) : pw * dp[kb][r];  // dSᵀ
                }
            }
            // dQᵀ[d][query] += Kᵀ[d][key] · dSᵀ[key][query]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8_t pf = acc_frag<T>(sc[kb], s);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
                        acc[dt] = mm<T>(
                            tr_frag<KP>(k_lds, kb * 32, s, dt * 32, lane), pf, acc[dt]);
                }
        }
    }
    if constexpr (MODE == 0) {
        l += __shfl_xor(l, 32, 64);
        float inv = l > 0.f ? 1.f / l : 0.f;
        if constexpr (DROP) inv *= p.drop_scale;
        store_rowT(o + b * p.o_bs + h * HD, p.o_rs, t, valid, acc, inv, lane);
        if (valid && lane < 32) lse[((long long)b * p.H + h) * p.T + t] = l > 0.f ? (m + log2f(l)) * LN2 : 0.f;
    } else {
        if (p.rope_cos) rope_acc_inverse<ND>(acc, p, p.rope_q0 + t, lane);  // (the gradient of the UNROTATED query)
        store_rowT(dq + b * p.dq_bs + h * HD, p.dq_rs, t, valid, acc, p.scale, lane);
    }
}

// ---- dK / dV backward: key on the lane ----
// WHICH: 0 = dK and dV (head_dim 64); 1 = dV only, 2 = dK only (head_dim 128: two launches, the accumulators of both
// would not fit the register file)
template <typename T, int HD, int WHICH, bool DROP>
__global__ __launch_bounds__(256, dkv_min_waves(HD)) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                           const T* __restrict__ v, const T* __restrict__ d_o,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta, T* __restrict__ dk,
                                                           T* __restrict__ dv, AttnParams p) {
    constexpr int NF = HD / 16, ND = HD / 32, NI = HD / 64;
    constexpr bool DO_V = WHICH != 2, DO_K = WHICH != 1;
    __shared__ __attribute__((aligned(16))) char q_lds[NI * KT * 128];   // dual-use images: read by rows and transposed
    __shared__ __attribute__((aligned(16))) char do_lds[NI * KT * 128];
    __shared__ __attribute__((aligned(16))) float l2_lds[KT], dl_lds[KT];
    // attention-probability dropout: the stored keep bits of (this tile's 64 queries) x (this workgroup's 128 keys), one dword
    // per (query, wave) — staged with the tile instead of one global byte load per score (the DROP instantiation ran 3.7x
    // longer per (query, key) pair than the plain one: 201 us for the IWSLT recipe's encoder self-attention)
    __shared__ unsigned m_lds[DROP ? KT * 4 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b, h, blk;  // (the key blocks of a (batch, head) pair on one XCD: they re-read the pair's Q / dO from its L2)
    if (!pair_block(p, p.nkb, b, h, blk)) return;
    const int s = blk * 128 + wave * 32 + (lane & 31);
    const bool kvalid = s < p.S && !(p.key_pad && p.key_pad[(long long)b * p.S + min(s, p.S - 1)]);
    // thread tid brings dword tid & 3 (32 keys) of query row tid >> 2 of the tile; rows past T and dwords past the row: 0
    auto mask_g2r = [&](int t0) -> unsigned {
        const int t = t0 + (tid >> 2);
        const long long byte0 = (long long)blk * 16 + (tid & 3) * 4;
        if (t >= p.T || byte0 + 4 > p.mask_pitch) return 0u;
        return *reinterpret_cast<const unsigned*>(p.drop_mask + (((long long)b * p.H + h) * p.T + t) * p.mask_pitch + byte0);
    };
    const float c = p.scale * LOG2E, inv_scale = 1.f / p.scale;

    bf16x8_t kf[NF], vf[NF];
    load_row_frags(kf, k + b * p.k_bs + h * HD, p.k_rs, s, s < p.S, lane);
    load_row_frags(vf, v + b * p.v_bs + h * HD, p.v_rs, s, s < p.S, lane);
    if (p.rope_cos) rope_frags<NF, T>(kf, p, p.rope_k0 + s, lane);
    // S = Q Kᵀ starts from the accumulator -lse / scale (rows of the staged tile) and leaves s - lse / scale, dP = dO Vᵀ starts
    // from -delta: p = exp2(c acc), dS = p dP' — three vector instructions per score besides the conversions (was ~10).  A
    // padding key / key past S is a whole lane here: its dK, dV rows are zeroed at the end instead of a -inf bias in every
    // score.  (With dropout dP is scaled before delta comes off: explicit form, zero accumulators.)
    f32x16 dka[ND], dva[ND];  // dKᵀ, dVᵀ: [d-tile][d rows] x key lane
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dka[dt][r] = dva[dt][r] = 0.f;

    // causal: queries before the first key of this workgroup (minus the offset) never see it
    const int off = p.S - p.T;
    int t_begin = 0;
    if (p.causal) t_begin = max(0, (int)(blk * 128) - off) / KT * KT;
    const int ws0 = blk * 128 + wave * 32;          // first key of this wave
    const unsigned keep_bit = s < p.S ? 1u << (lane & 31) : 0u;  // (dropout) this lane's bit of a stored keep dword
    const T* qbase = q + b * p.q_bs + h * HD;
    const T* dobase = d_o + b * p.do_bs + h * HD;
    uint4 qreg[2 * NI], doreg[2 * NI];
    unsigned mreg = 0u;
    if (t_begin < p.T) {
        tile_g2r(qreg, qbase, p.q_rs, t_begin, p.T, tid);
        tile_g2r(doreg, dobase, p.do_rs, t_begin, p.T, tid);
        if (p.rope_cos) rope_tile<2 * NI, T>(qreg, p, p.rope_q0 + t_begin, tid);
        if constexpr (DROP) mreg = mask_g2r(t_begin);
    }
    for (int t0 = t_begin; t0 < p.T; t0 += KT) {
        __syncthreads();
        tile_r2s<DUAL>(qreg, q_lds, tid);
        tile_r2s<DUAL>(doreg, do_lds, tid);
        if constexpr (DROP) m_lds[tid] = mreg;
        if (tid < KT) {
            int t = t0 + tid;
            long long row = ((long long)b * p.H + h) * p.T + t;
            l2_lds[tid] = t < p.T ? -lse[row] * inv_scale : -INFINITY;  // (initial accumulators, in units of the raw score) -inf -> p = 0 for rows past T
            dl_lds[tid] = t < p.T ? -delta[row] : 0.f;
        }
        if (t0 + KT < p.T) {  // prefetch the next query tile into registers
            tile_g2r(qreg, qbase, p.q_rs, t0 + KT, p.T, tid);
            tile_g2r(doreg, dobase, p.do_rs, t0 + KT, p.T, tid);
            if (p.rope_cos) rope_tile<2 * NI, T>(qreg, p, p.rope_q0 + t0 + KT, tid);
            if constexpr (DROP) mreg = mask_g2r(t0 + KT);
        }
        __syncthreads();
        // causal classification of (this tile's 64 queries) x (this wave's 32 keys): wave-uniform
        if (p.causal && ws0 > t0 + KT - 1 + off) continue;        // every key is in the future of every query
        const bool check = p.causal && ws0 + 31 > t0 + off;      // some pairs are masked
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 sc, dp;  // S[query][key], dP[query][key]
            float4 d4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are 4 consecutive queries: one 16-B read each of -lse, -delta
                const int tl = qb * 32 + 8 * g + 4 * (lane >> 5);
                const float4 l4 = *reinterpret_cast<const float4*>(l2_lds + tl);
                d4[g] = *reinterpret_cast<const float4*>(dl_lds + tl);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sc[4 * g + j] = (&l4.x)[j];
                    dp[4 * g + j] = DROP ? 0.f : (&d4[g].x)[j];
                }
            }
#pragma unroll
            for (int kk = 0; kk < NF; ++kk) {
                sc = mm<T>(row_frag<DUAL>(q_lds, qb * 32, kk, lane), kf[kk], sc);
                if constexpr (DO_K)
                    dp = mm<T>(row_frag<DUAL>(do_lds, qb * 32, kk, lane), vf[kk], dp);
            }
            if (check) {  // wave-uniform: the causal boundary crosses this (tile, wave) block
                int sq = s - off - t0 - qb * 32 - 4 * (lane >> 5);  // key s is visible to query index (in the block) >= sq
                asm volatile("; causal block" : "+v"(sq));          // (a real branch: if-converted, the selects cost 4 instructions per score in every tile)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) < sq) sc[r] = -INFINITY;
            }
            f32x16 ds;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pw = __builtin_amdgcn_exp2f(sc[4 * g + j] * c);
                    if constexpr (DROP) {  // keep bit of (query t0+tl+j, this lane's key): bit lane & 31 of the wave's dword
                        const int tl = qb * 32 + 8 * g + 4 * (lane >> 5);
                        const unsigned wd = m_lds[(tl + j) * 4 + wave];
                        const float km = (wd & keep_bit) ? p.drop_scale : 0.f;  // (rows past T were staged as zeros)
                        sc[4 * g + j] = pw * km;
                        ds[4 * g + j] = pw * fmaf(dp[4 * g + j], km, (&d4[g].x)[j]);
                    } else {
                        sc[4 * g + j] = pw;
                        ds[4 * g + j] = pw * dp[4 * g + j];
                    }
                }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8_t pf = acc_frag<T>(sc, st), dsf = acc_frag<T>(ds, st);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    // dVᵀ[d][key] += dOᵀ[d][query] · P[query][key] ;  dKᵀ[d][key] += Qᵀ[d][query] · dS[query][key]
                    if constexpr (DO_V)
                        dva[dt] = mm<T>(
                            tr_frag<DUAL>(do_lds, qb * 32, st, dt * 32, lane), pf, dva[dt]);
                    if constexpr (DO_K)
                        dka[dt] = mm<T>(
                            tr_frag<DUAL>(q_lds, qb * 32, st, dt * 32, lane), dsf, dka[dt]);
                }
            }
        }
    }
    if (!kvalid) {  // padding keys: zero rows (a select, not a product: their probabilities are unbounded)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dka[dt][r] = dva[dt][r] = 0.f;
    }
    if constexpr (DO_K) {
        if (p.rope_cos) rope_acc_inverse<ND>(dka, p, p.rope_k0 + s, lane);
        store_rowT(dk + b * p.dk_bs + h * HD, p.dk_rs, s, s < p.S, dka, p.scale, lane);
    }
    if constexpr (DO_V) store_rowT(dv + b * p.dv_bs + h * HD, p.dv_rs, s, s < p.S, dva, 1.f, lane);
}

// ---- whole backward of one (batch, head) in one workgroup, for T <= 128 queries and S <= 128 keys ----
// (the training shapes of the path: sentences of <= 128 tokens).  Q, dO, O, K are read from HBM exactly once; S and dP
// are computed once (the split dQ / dKdV kernels each recompute them).  Phase 1 (key on the lane, wave w owns keys
// 32w..32w+31) produces dK, dV and leaves dSᵀ[key][query] in LDS; phase 2 (query on the lane, wave w owns queries
// 32w..) contracts it with the K tile: dQᵀ[d][q] = Σ_key Kᵀ[d][key] · dSᵀ[key][q].
constexpr int FUSED_LDS = 2 * (2 * KT * 128) + 128 * 256 + 2 * 128 * 4 + 128 * 16;  // Q | dO (K later) | dSᵀ | lse, delta | keep bits
__device__ __forceinline__ int ds_off(int key, int qcol) {  // dSᵀ image: 256-B rows, 16-B chunk swizzled so that both
    // the 8-B row-segment writes of phase 1 and the transposed reads of phase 2 spread over the banks
    return key * 256 + ((((qcol >> 3) ^ ((key & 3) << 2) ^ ((key >> 2) & 3)) & 15) << 4) + (qcol & 7) * 2;
}
__device__ __forceinline__ bf16x8_t ds_tr_frag(const char* lds, int row0, int s, int c0, int lane) {
    int q = (lane & 15) >> 2, p4 = lane & 3;
    int col = c0 + 16 * ((lane >> 4) & 1) + 4 * p4;
    int row = row0 + 16 * s + 4 * (lane >> 5) + q;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + ds_off(row, col)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + ds_off(row + 8, col)));
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, f);
}

template <typename T, bool DROP>
__global__ __launch_bounds__(256, PK_ATTN_FUSED_WAVES) void attn_bwd_fused128_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ o,
    const T* __restrict__ d_o, const float* __restrict__ lse, float* __restrict__ delta, T* __restrict__ dq,
    T* __restrict__ dk, T* __restrict__ dv, AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char fused_lds[];
    char* q_lds = fused_lds;                    // [128 queries] dual image
    char* do_lds = fused_lds + 2 * KT * 128;    // [128 queries] dual image; reused for the K tile in phase 2
    char* ds_lds = do_lds + 2 * KT * 128;       // dSᵀ [128 keys][128 queries]
    float* l2_lds = reinterpret_cast<float*>(ds_lds + 128 * 256);
    float* dl_lds = l2_lds + 128;
    unsigned* m_lds = reinterpret_cast<unsigned*>(dl_lds + 128);  // DROP: keep bits [128 queries][4 dwords of 32 keys]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x;
    const float c = p.scale * LOG2E;
    const int off = p.S - p.T;

    // ---- loads: Q, dO (+O for delta) tiles, the K tile (kept in registers until phase 2), own-key K/V fragments ----
    // EVERY global load of the prologue is requested before the first wait, without a branch in between: rows past T / S
    // read the last row and are zeroed afterwards.  (Behind `if (r < p.T)` each of the four row groups was a block of its
    // own, and at a block join hipcc waits for every load in flight — the prologue was four tile round trips, then the
    // statistics, the padding byte and the K / V fragments one after the other: seven serial trips to memory out of the
    // ~19 us a workgroup lives.)
    const T* qbase = q + b * p.q_bs + h * HD64;
    const T* dobase = d_o + b * p.do_bs + h * HD64;
    const T* obase = o + b * p.o_bs + h * HD64;
    const T* kbase = k + b * p.k_bs + h * HD64;
    uint4 kreg[4], qreg[4], doreg[4], oreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cidx = tid + i * 256, r = cidx >> 3, ch = cidx & 7;
        const int rq = min(r, p.T - 1), rk = min(r, p.S - 1);
        qreg[i] = att_ld<1>(qbase + (long long)rq * p.q_rs + ch * 8);
        doreg[i] = att_ld<1>(dobase + (long long)rq * p.do_rs + ch * 8);
        oreg[i] = att_ld<1>(obase + (long long)rq * p.o_rs + ch * 8);
        kreg[i] = att_ld<1>(kbase + (long long)rk * p.k_rs + ch * 8);
    }
    const int s = wave * 32 + (lane & 31);
    bf16x8_t kf[4], vf[4];
    load_row_frags(kf, kbase, p.k_rs, s, s < p.S, lane);
    load_row_frags(vf, v + b * p.v_bs + h * HD64, p.v_rs, s, s < p.S, lane);
    const float lse_t = lse[((long long)b * p.H + h) * p.T + min(tid & 127, p.T - 1)];
    // (no padding mask: any readable byte — the select below ignores it)
    const unsigned char* padp = p.key_pad ? p.key_pad + (long long)b * p.S + min(s, p.S - 1) : reinterpret_cast<const unsigned char*>(lse);
    const unsigned char padv = *padp;
    if constexpr (DROP) {  // (staged with the tiles; one dword per (query, wave) instead of a global byte load per score)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, t = idx >> 2, byte0 = (idx & 3) * 4;
            unsigned w = 0u;
            if (t < p.T && byte0 + 4 <= p.mask_pitch)
                w = *reinterpret_cast<const unsigned*>(p.drop_mask + (((long long)b * p.H + h) * p.T + t) * p.mask_pitch + byte0);
            m_lds[idx] = w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cidx = tid + i * 256, r = cidx >> 3, ch = cidx & 7;
        uint4 qv = sel4(r < p.T, qreg[i]);
        const uint4 dov = sel4(r < p.T, doreg[i]), ov = sel4(r < p.T, oreg[i]);
        kreg[i] = sel4(r < p.S, kreg[i]);
        if (p.rope_cos) {
            qv = rope_chunk<8, T>(qv, ch, p, p.rope_q0 + r);
            kreg[i] = rope_chunk<8, T>(kreg[i], ch, p, p.rope_k0 + r);
        }
        *reinterpret_cast<uint4*>(q_lds + lds_off<DUAL>(r, ch)) = qv;
        *reinterpret_cast<uint4*>(do_lds + lds_off<DUAL>(r, ch)) = dov;
        float part = frag_dot<T>(__builtin_bit_cast(bf16x8_t, dov), __builtin_bit_cast(bf16x8_t, ov));
        part = lanes8_sum(part);  // the 8 lanes of a row
        if (ch == 0) {
            dl_lds[r] = DROP ? part : -part;  // (negated without dropout: the initial accumulator of dP)
            if (r < p.T) delta[((long long)b * p.H + h) * p.T + r] = part;
        }
    }
    // the initial accumulator of S, in units of the raw score (rows past T: -inf -> p = 0)
    if (tid < 128) l2_lds[tid] = tid < p.T ? -lse_t / p.scale : -INFINITY;
    const bool kvalid = s < p.S && !(p.key_pad && padv);
    if (p.rope_cos) rope_frags<4, T>(kf, p, p.rope_k0 + s, lane);
    // S starts from the accumulator -lse / scale (-inf on the lane of a padding key / key past S), dP from -delta:
    // p = exp2(c acc), dS = p dP' (as in the dK / dV kernel above)
    f32x16 dka[2], dva[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dka[dt][r] = dva[dt][r] = 0.f;
    __syncthreads();

    // ---- phase 1: key on the lane ----
    const int ws0 = wave * 32;
    const unsigned keep_bit = s < p.S ? 1u << (lane & 31) : 0u;  // (dropout) this lane's bit of a stored keep dword
    for (int qb = 0; qb < 4; ++qb) {
        const int t0 = qb * 32;
        if (t0 >= p.T) break;
        if (p.causal && ws0 > t0 + 31 + off) continue;       // every key of this wave is in the future of these queries
        const bool check = p.causal && ws0 + 31 > t0 + off;
        f32x16 sc, dp;
        float4 d4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int tl = t0 + 8 * g + 4 * (lane >> 5);
            const float4 l4 = *reinterpret_cast<const float4*>(l2_lds + tl);
            d4[g] = *reinterpret_cast<const float4*>(dl_lds + tl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[4 * g + j] = kvalid ? (&l4.x)[j] : -INFINITY;
                dp[4 * g + j] = DROP ? 0.f : (&d4[g].x)[j];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            sc = mm<T>(row_frag<DUAL>(q_lds, t0, kk, lane), kf[kk], sc);
            dp = mm<T>(row_frag<DUAL>(do_lds, t0, kk, lane), vf[kk], dp);
        }
        if (check) {  // wave-uniform: the causal boundary crosses this (queries, keys) block
            int sq = s - off - t0 - 4 * (lane >> 5);  // key s is visible to query index (in the block) >= sq
            asm volatile("; causal block" : "+v"(sq));  // (a real branch: if-converted, the selects cost 4 instructions per score in every block)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) < sq) sc[r] = -INFINITY;
        }
        f32x16 ds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int tl = t0 + 8 * g + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pw = __builtin_amdgcn_exp2f(sc[4 * g + j] * c);
                if constexpr (DROP) {
                    const unsigned wd = m_lds[(tl + j) * 4 + wave];
                    const float km = (wd & keep_bit) ? p.drop_scale : 0.f;  // (rows past T were staged as zeros)
                    sc[4 * g + j] = pw * km;
                    ds[4 * g + j] = pw * fmaf(dp[4 * g + j], km, -(&d4[g].x)[j]);
                } else {
                    sc[4 * g + j] = pw;
                    ds[4 * g + j] = pw * dp[4 * g + j];
                }
            }
            unsigned lo = (unsigned)H16<T>::bits(ds[4 * g]) | ((unsigned)H16<T>::bits(ds[4 * g + 1]) << 16);
            unsigned hi = (unsigned)H16<T>::bits(ds[4 * g + 2]) | ((unsigned)H16<T>::bits(ds[4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(ds_lds + ds_off(s, tl)) = make_uint2(lo, hi);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8_t pf = acc_frag<T>(sc, st), dsf = acc_frag<T>(ds, st);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dva[dt] = mm<T>(tr_frag<DUAL>(do_lds, t0, st, dt * 32, lane), pf,
                                                                  dva[dt]);
                dka[dt] = mm<T>(tr_frag<DUAL>(q_lds, t0, st, dt * 32, lane), dsf,
                                                                  dka[dt]);
            }
        }
    }
    if (p.rope_cos) rope_acc_inverse<2>(dka, p, p.rope_k0 + s, lane);
    store_rowT(dk + b * p.dk_bs + h * HD64, p.dk_rs, s, s < p.S, dka, p.scale, lane);
    store_rowT(dv + b * p.dv_bs + h * HD64, p.dv_rs, s, s < p.S, dva, 1.f, lane);
    __syncthreads();  // dSᵀ complete; the dO tile is dead
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cidx = tid + i * 256;
        *reinterpret_cast<uint4*>(do_lds + lds_off<DUAL>(cidx >> 3, cidx & 7)) = kreg[i];
    }
    __syncthreads();

    // ---- phase 2: query on the lane ----
    const int t = wave * 32 + (lane & 31);
    if (wave * 32 >= p.T) return;
    f32x16 acc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    for (int kb = 0; kb < 4; ++kb) {
        if (kb * 32 >= p.S) break;
        if (p.causal && kb * 32 > wave * 32 + 31 + off) break;  // the (queries, keys) blocks phase 1 skipped
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8_t dsf = ds_tr_frag(ds_lds, kb * 32, st, wave * 32, lane);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                acc[dt] = mm<T>(tr_frag<DUAL>(do_lds, kb * 32, st, dt * 32, lane), dsf,
                                                                  acc[dt]);
        }
    }
    if (p.rope_cos) rope_acc_inverse<2>(acc, p, p.rope_q0 + t, lane);
    store_rowT(dq + b * p.dq_bs + h * HD64, p.dq_rs, t, t < p.T, acc, p.scale, lane);
}

// ---- attention weights (B, T, H, S) for `return_attn` / return_layers: modules.py:742-771 (fp32 softmax, nan_to_num) ----
// Not on the training path (the flash kernels never materialise the weights): one thread per query row, two passes over
// the keys (max / sum, then the normalised weights), K tiles staged in LDS as fp32.
template <typename T, int HD>
__global__ __launch_bounds__(128) void attn_probs_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         T* __restrict__ probs, AttnParams p) {
    __shared__ float ks[F32_TILE][HD + 1];
    const int b = blockIdx.z, h = blockIdx.y, t = blockIdx.x * 128 + threadIdx.x;
    const bool valid = t < p.T;
    float qr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] = valid ? to_f32(q[b * p.q_bs + (long long)t * p.q_rs + h * HD + d]) : 0.f;
    const float c = p.scale * LOG2E;
    float m = -INFINITY, l = 0.f;
    T* out = probs + (((long long)b * p.T + t) * p.H + h) * p.S;
    for (int pass = 0; pass < 2; ++pass) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        for (int s0 = 0; s0 < p.S; s0 += F32_TILE) {
            __syncthreads();
            for (int i = threadIdx.x; i < F32_TILE * HD; i += 128) {
                int r = i / HD, d = i % HD, s = s0 + r;
                ks[r][d] = s < p.S ? to_f32(k[b * p.k_bs + (long long)s * p.k_rs + h * HD + d]) : 0.f;
            }
            __syncthreads();
            for (int r = 0; r < F32_TILE; ++r) {
                const int s = s0 + r;
                if (!valid || s >= p.S) continue;
                const bool masked = key_masked(p, b, t, s);
                float s2 = 0.f;
                if (!masked) {
#pragma unroll
                    for (int d = 0; d < HD; ++d) s2 += qr[d] * ks[r][d];
                    s2 *= c;
                }
                if (pass == 0) {
                    if (masked) continue;
                    const float mn = fmaxf(m, s2);
                    l = l * exp2f(m - mn) + exp2f(s2 - mn);
                    m = mn;
                } else {
                    out[s] = from_f32<T>(masked ? 0.f : exp2f(s2 - m) * inv);  // a fully masked row is all zeros
                }
            }
        }
    }
}

int check_common(const AttnParams& p, int hd, int dtype, const char* who) {
    PK_CHECK_ARG(hd == 64 || hd == 128, "%s: head_dim %d not supported (64 or 128)", who, hd);
    PK_CHECK_ARG(dtype == PK_F32 || dtype == PK_BF16 || dtype == PK_F16, "%s: dtype %d not supported", who, dtype);
    PK_CHECK_ARG(p.B >= 0 && p.H > 0 && p.T >= 0 && p.S >= 0, "%s: bad sizes", who);
    PK_CHECK_ARG(p.B <= 65535 && p.H <= 65535, "%s: B and H must be <= 65535", who);
    if (dtype != PK_F32) {
        PK_CHECK_ARG(p.q_rs % 8 == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0 && p.o_rs % 8 == 0 && p.q_bs % 8 == 0 &&
                         p.k_bs % 8 == 0 && p.v_bs % 8 == 0 && p.o_bs % 8 == 0,
                     "%s: bf16 strides must be multiples of 8 elements", who);
    }
    return 0;
}

struct RopeArg {
    const float* cos_t;
    const float* sin_t;
    int max_pos, q0, k0;
};
int set_rope(AttnParams& p, const RopeArg* r, int hd, const char* who) {
    if (!r || !r->cos_t) return 0;
    PK_CHECK_ARG(r->sin_t && r->max_pos > 0 && r->q0 >= 0 && r->k0 >= 0, "%s: bad rotary tables / offsets", who);
    PK_CHECK_ARG(r->q0 + p.T <= r->max_pos && r->k0 + p.S <= r->max_pos, "%s: positions exceed the cos/sin table (%d rows)", who,
                 r->max_pos);
    PK_CHECK_ARG(((uintptr_t)r->cos_t % 16) == 0 && ((uintptr_t)r->sin_t % 16) == 0, "%s: cos/sin tables must be 16-byte aligned", who);
    p.rope_cos = r->cos_t; p.rope_sin = r->sin_t; p.rope_max = r->max_pos; p.rope_q0 = r->q0; p.rope_k0 = r->k0;
    (void)hd;
    return 0;
}

int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse,
                           const unsigned char* key_pad, int B, int H, int T, int S, int hd, long long q_bs,
                           long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                           long long o_bs, long long o_rs, int causal, float scale, float drop_p,
                           unsigned long long seed, unsigned long long offset, unsigned char* drop_mask, int dtype,
                           void* stream, const RopeArg* rope) {
    AttnParams p = {};
    p.B = B; p.H = H; p.T = T; p.S = S;
    if (int rc = set_rope(p, rope, hd, "pk_attn_fwd_rope")) return rc;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.key_pad = key_pad; p.causal = causal; p.scale = scale;
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_attn_fwd: bad dropout %f", drop_p);
    PK_CHECK_ARG(drop_p == 0.f || drop_mask, "pk_attn_fwd: dropout needs the drop_mask buffer");
    if (drop_p > 0.f) {
        p.drop_thr = dropout_threshold(drop_p);
        p.drop_scale = 1.f / (1.f - drop_p);
        p.seed = seed; p.offset = offset; p.drop_mask = drop_mask;
        p.mask_pitch = 8LL * ((S + 63) / 64);
    }
    if (int rc = check_common(p, hd, dtype, "pk_attn_fwd")) return rc;
    if (B == 0 || T == 0) return 0;
    PK_CHECK_ARG(q && k && v && o && lse, "pk_attn_fwd: null tensor");
    if (B == 0 || T == 0) return 0;
    dim3 grid((T + 127) / 128, H, B);
    p.nqb = (T + 127) / 128;
    const dim3 grid16((unsigned)((B * H + 7) / 8 * 8) * p.nqb);  // (16-bit kernels: 1-D, pair_block)
    hipStream_t s = (hipStream_t)stream;
    // heads of 64, no causal mask, no rotation, at least one whole key tile: the lagging-maximum kernel (attention_long.hip).  Built
    // for long key sequences, it is also the faster one at the short ones (B 256, H 8, T 128: S = 64 / 96 / 128 / 192 25.3 / 31.0 /
    // 32.2 / 38.7 -> 22.7 / 27.1 / 28.3 / 33.9 us), and with the threshold below every padded length of the training shapes a batch
    // and its padded twin run on one kernel.  (PK_ATTN_LONG_MIN_S: diagnostic)
    static const int long_min_s = [] { const char* e = getenv("PK_ATTN_LONG_MIN_S"); return e ? atoi(e) : 64; }();
    if (dtype != PK_F32 && hd == 64 && !p.rope_cos && S >= long_min_s && (long long)S * std::max(k_rs, v_rs) * 2 < (1LL << 31)) {
        PK_CHECK_ARG(pk_attn_fwd_long_launch(q, k, v, o, lse, p, dtype, s) == 0, "pk_attn_fwd: launch of the long-sequence kernel failed");
        PK_LAUNCH_CHECK();
        return 0;
    }
#define PK_FWD16(TT, D)                                                                                              \
    do {                                                                                                             \
        if (p.drop_thr)                                                                                              \
            hipLaunchKernelGGL((attn_q_kernel<TT, 0, D, true>), grid16, dim3(256), 0, s, (const TT*)q, (const TT*)k,   \
                               (const TT*)v, (TT*)o, (const TT*)nullptr, lse, (float*)nullptr, (TT*)nullptr, p);     \
        else                                                                                                         \
            hipLaunchKernelGGL((attn_q_kernel<TT, 0, D, false>), grid16, dim3(256), 0, s, (const TT*)q, (const TT*)k,  \
                               (const TT*)v, (TT*)o, (const TT*)nullptr, lse, (float*)nullptr, (TT*)nullptr, p);     \
    } while (0)
#define PK_FWD(D)                                                                                                    \
    do {                                                                                                             \
        if (dtype == PK_BF16) PK_FWD16(bf16, D);                                                                     \
        else if (dtype == PK_F16) PK_FWD16(f16, D);                                                                  \
        else                                                                                                         \
            hipLaunchKernelGGL((attn_fwd_f32_kernel<D>), grid, dim3(128), 0, s, (const float*)q, (const float*)k,    \
                               (const float*)v, (float*)o, lse, p);                                                  \
    } while (0)
    if (hd == 64) PK_FWD(64);
    else PK_FWD(128);
#undef PK_FWD16
#undef PK_FWD
    PK_LAUNCH_CHECK();
    return 0;
}

int attn_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse, float* delta, void* dq, void* dk, void* dv,
                           const unsigned char* key_pad, int B, int H, int T, int S, int hd, long long q_bs,
                           long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                           long long o_bs, long long o_rs, long long do_bs, long long do_rs, long long dq_bs,
                           long long dq_rs, long long dk_bs, long long dk_rs, long long dv_bs, long long dv_rs,
                           int causal, float scale, float drop_p, const unsigned char* drop_mask, int dtype,
                           void* stream, const RopeArg* rope) {
    AttnParams p = {};
    p.B = B; p.H = H; p.T = T; p.S = S;
    if (int rc = set_rope(p, rope, hd, "pk_attn_bwd_rope")) return rc;
    PK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pk_attn_bwd: bad dropout %f", drop_p);
    PK_CHECK_ARG(drop_p == 0.f || drop_mask, "pk_attn_bwd: dropout needs the drop_mask of the forward call");
    if (drop_p > 0.f) {
        p.drop_thr = dropout_threshold(drop_p);
        p.drop_scale = 1.f / (1.f - drop_p);
        p.drop_mask = const_cast<unsigned char*>(drop_mask);
        p.mask_pitch = 8LL * ((S + 63) / 64);
    }
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.do_bs = do_bs; p.do_rs = do_rs; p.dq_bs = dq_bs; p.dq_rs = dq_rs; p.dk_bs = dk_bs; p.dk_rs = dk_rs;
    p.dv_bs = dv_bs; p.dv_rs = dv_rs;
    p.key_pad = key_pad; p.causal = causal; p.scale = scale;
    if (int rc = check_common(p, hd, dtype, "pk_attn_bwd")) return rc;
    if (B == 0) return 0;
    PK_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "pk_attn_bwd: null tensor");
    if (dtype != PK_F32)
        PK_CHECK_ARG(do_rs % 8 == 0 && dq_rs % 8 == 0 && dk_rs % 8 == 0 && dv_rs % 8 == 0 && do_bs % 8 == 0 &&
                         dq_bs % 8 == 0 && dk_bs % 8 == 0 && dv_bs % 8 == 0,
                     "pk_attn_bwd: bf16 strides must be multiples of 8 elements");
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    p.nqb = (T + 127) / 128; p.nkb = (S + 127) / 128;
    const unsigned np8 = (unsigned)((B * H + 7) / 8 * 8);
    dim3 gq(np8 * p.nqb), gk(np8 * p.nkb);  // (1-D: pair_block)
    const dim3 gq3((T + 127) / 128, H, B), gk3((S + 127) / 128, H, B);  // (the fp32 kernels: one thread per row, 3-D)
    static const bool no_fused = getenv("PK_ATTN_NO_FUSED_BWD") != nullptr;
    const bool half = dtype != PK_F32;
#define PK_T16(...) do { if (dtype == PK_F16) { using TT = f16; __VA_ARGS__; } else { using TT = bf16; __VA_ARGS__; } } while (0)
    if (half && hd == 64 && T > 0 && S > 0 && T <= 128 && S <= 128 && !no_fused) {
        static const int attr_rc = [] {
            int rc = 0;
            const void* fns[] = {(const void*)attn_bwd_fused128_kernel<bf16, false>, (const void*)attn_bwd_fused128_kernel<bf16, true>,
                                 (const void*)attn_bwd_fused128_kernel<f16, false>, (const void*)attn_bwd_fused128_kernel<f16, true>};
            for (const void* f : fns) {
                int e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS);
                if (e) rc = e;
            }
            return rc;
        }();
        PK_CHECK_ARG(attr_rc == 0, "pk_attn_bwd: cannot reserve %d B of LDS", FUSED_LDS);
#define PK_FUSED(DR)                                                                                                   \
    PK_T16(hipLaunchKernelGGL((attn_bwd_fused128_kernel<TT, DR>), dim3(H, B), dim3(256), FUSED_LDS, s, (const TT*)q,   \
                              (const TT*)k, (const TT*)v, (const TT*)o, (const TT*)d_o, lse, delta, (TT*)dq, (TT*)dk, \
                              (TT*)dv, p))
        if (p.drop_thr) PK_FUSED(true);
        else PK_FUSED(false);
#undef PK_FUSED
    } else if (half) {
#define PK_DQ_(D, DR)                                                                                                \
    PK_T16(hipLaunchKernelGGL((attn_q_kernel<TT, 1, D, DR>), gq, dim3(256), 0, s, (const TT*)q, (const TT*)k,        \
                              (const TT*)v, (TT*)const_cast<void*>(o), (const TT*)d_o, const_cast<float*>(lse), delta, \
                              (TT*)dq, p))
#define PK_DQ(D) do { if (p.drop_thr) PK_DQ_(D, true); else PK_DQ_(D, false); } while (0)
#define PK_DKV_(D, W, DR)                                                                                            \
    PK_T16(hipLaunchKernelGGL((attn_bwd_dkv_kernel<TT, D, W, DR>), gk, dim3(256), 0, s, (const TT*)q, (const TT*)k,  \
                              (const TT*)v, (const TT*)d_o, lse, (const float*)delta, (TT*)dk, (TT*)dv, p))
#define PK_DKV(D, W) do { if (p.drop_thr) PK_DKV_(D, W, true); else PK_DKV_(D, W, false); } while (0)
        // long key sequences, heads of 64, no causal mask, no rotation: the dQ kernel with the forward's frame (attention_long.hip)
        static const int long_min_s = [] { const char* ev = getenv("PK_ATTN_LONG_MIN_BWD"); return ev ? atoi(ev) : 256; }();  // (the streamed length)
        static const bool long_dq = [] { const char* ev = getenv("PK_ATTN_LONG_DQ"); return !ev || atoi(ev) != 0; }();  // (A/B)
        if (T > 0 && long_dq && hd == 64 && !p.rope_cos && S >= long_min_s &&
            (long long)S * std::max(k_rs, v_rs) * 2 < (1LL << 31)) {
            PK_CHECK_ARG(pk_attn_dq_long_launch(q, k, v, o, d_o, lse, delta, dq, p, dtype, s) == 0,
                         "pk_attn_bwd: launch of the long-sequence dQ kernel failed");
        } else if (T > 0) {
            if (hd == 64) PK_DQ(64);
            else PK_DQ(128);
        }
        PK_LAUNCH_CHECK();
        static const bool long_dkv = [] { const char* ev = getenv("PK_ATTN_LONG_DKV"); return !ev || atoi(ev) != 0; }();  // (A/B)
        if (S > 0 && long_dkv && hd == 64 && !p.rope_cos && T >= long_min_s &&
            (long long)T * std::max(q_rs, do_rs) * 2 < (1LL << 31)) {
            PK_CHECK_ARG(pk_attn_dkv_long_launch(q, k, v, d_o, lse, delta, dk, dv, p, dtype, s) == 0,
                         "pk_attn_bwd: launch of the long-sequence dK / dV kernel failed");
        } else if (S > 0) {
            if (hd == 64) {
                PK_DKV(64, 0);
            } else {  // head_dim 128: dV and dK in two launches (each recomputes S; both accumulators do not fit)
                PK_DKV(128, 1);
                PK_LAUNCH_CHECK();
                PK_DKV(128, 2);
            }
        }
#undef PK_DQ
#undef PK_DQ_
#undef PK_DKV
#undef PK_DKV_
    } else {
#define PK_BWD32(D)                                                                                                  \
    do {                                                                                                             \
        if (T > 0)                                                                                                   \
            hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<D>), gq3, dim3(128), 0, s, (const float*)q, (const float*)k,   \
                               (const float*)v, (const float*)o, (const float*)d_o, lse, delta, (float*)dq, p);      \
        if (S > 0)                                                                                                   \
            hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<D>), gk3, dim3(128), 0, s, (const float*)q, (const float*)k,  \
                               (const float*)v, (const float*)d_o, lse, (const float*)delta, (float*)dk, (float*)dv, \
                               p);                                                                                   \
    } while (0)
        if (hd == 64) PK_BWD32(64);
        else PK_BWD32(128);
#undef PK_BWD32
    }
#undef PK_T16
    PK_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int pk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                           const unsigned char* key_pad, int B, int H, int T, int S, int hd, long long q_bs,
                           long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                           long long o_bs, long long o_rs, int causal, float scale, float drop_p,
                           unsigned long long seed, unsigned long long offset, unsigned char* drop_mask, int dtype,
                           void* stream) {
    return attn_fwd_impl(q, k, v, o, lse, key_pad, B, H, T, S, hd, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, causal, scale,
                         drop_p, seed, offset, drop_mask, dtype, stream, nullptr);
}

extern "C" int pk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse, float* delta, void* dq, void* dk, void* dv,
                           const unsigned char* key_pad, int B, int H, int T, int S, int hd, long long q_bs,
                           long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                           long long o_bs, long long o_rs, long long do_bs, long long do_rs, long long dq_bs,
                           long long dq_rs, long long dk_bs, long long dk_rs, long long dv_bs, long long dv_rs,
                           int causal, float scale, float drop_p, const unsigned char* drop_mask, int dtype,
                           void* stream) {
    return attn_bwd_impl(q, k, v, o, d_o, lse, delta, dq, dk, dv, key_pad, B, H, T, S, hd, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
                         o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, causal, scale, drop_p, drop_mask,
                         dtype, stream, nullptr);
}

// The same with rotary positions applied inside the kernels (AttnParams::rope_cos): q and k are the UNROTATED projection,
// dq and dk the gradients with respect to it.  cos_t / sin_t: fp32 [max_pos][hd / 2]; query t sits at position q_pos0 + t,
// key s at k_pos0 + s.
extern "C" int pk_attn_fwd_rope(const void* q, const void* k, const void* v, void* o, float* lse,
                                const unsigned char* key_pad, int B, int H, int T, int S, int hd, long long q_bs,
                                long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                                long long o_bs, long long o_rs, int causal, float scale, float drop_p,
                                unsigned long long seed, unsigned long long offset, unsigned char* drop_mask,
                                const float* cos_t, const float* sin_t, int max_pos, int q_pos0, int k_pos0, int dtype,
                                void* stream) {
    PK_CHECK_ARG(cos_t && sin_t, "pk_attn_fwd_rope: null cos/sin table");
    const RopeArg r = {cos_t, sin_t, max_pos, q_pos0, k_pos0};
    return attn_fwd_impl(q, k, v, o, lse, key_pad, B, H, T, S, hd, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, causal, scale,
                         drop_p, seed, offset, drop_mask, dtype, stream, &r);
}

extern "C" int pk_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                const float* lse, float* delta, void* dq, void* dk, void* dv,
                                const unsigned char* key_pad, int B, int H, int T, int S, int hd, long long q_bs,
                                long long q_rs, long long k_bs, long long k_rs, long long v_bs, long long v_rs,
                                long long o_bs, long long o_rs, long long do_bs, long long do_rs, long long dq_bs,
                                long long dq_rs, long long dk_bs, long long dk_rs, long long dv_bs, long long dv_rs,
                                int causal, float scale, float drop_p, const unsigned char* drop_mask,
                                const float* cos_t, const float* sin_t, int max_pos, int q_pos0, int k_pos0, int dtype,
                                void* stream) {
    PK_CHECK_ARG(cos_t && sin_t, "pk_attn_bwd_rope: null cos/sin table");
    const RopeArg r = {cos_t, sin_t, max_pos, q_pos0, k_pos0};
    return attn_bwd_impl(q, k, v, o, d_o, lse, delta, dq, dk, dv, key_pad, B, H, T, S, hd, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
                         o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, causal, scale, drop_p, drop_mask,
                         dtype, stream, &r);
}

extern "C" int pk_attn_probs(const void* q, const void* k, void* probs, const unsigned char* key_pad, int B, int H, int T,
                             int S, int hd, long long q_bs, long long q_rs, long long k_bs, long long k_rs, int causal,
                             float scale, int dtype, void* stream) {
    AttnParams p = {};
    p.B = B; p.H = H; p.T = T; p.S = S;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs;
    p.key_pad = key_pad; p.causal = causal; p.scale = scale;
    PK_CHECK_ARG(hd == 64 || hd == 128, "pk_attn_probs: head_dim %d not supported (64 or 128)", hd);
    PK_CHECK_ARG(dtype == PK_F32 || dtype == PK_BF16 || dtype == PK_F16, "pk_attn_probs: dtype %d not supported", dtype);
    PK_CHECK_ARG(B >= 0 && H > 0 && T >= 0 && S >= 0 && B <= 65535 && H <= 65535, "pk_attn_probs: bad sizes");
    if (B == 0 || T == 0 || S == 0) return 0;
    PK_CHECK_ARG(q && k && probs, "pk_attn_probs: null tensor");
    dim3 grid((T + 127) / 128, H, B);
    hipStream_t s = (hipStream_t)stream;
#define PK_PR(TT, D) hipLaunchKernelGGL((attn_probs_kernel<TT, D>), grid, dim3(128), 0, s, (const TT*)q, (const TT*)k, (TT*)probs, p)
    if (dtype == PK_BF16) { if (hd == 64) PK_PR(bf16, 64); else PK_PR(bf16, 128); }
    else if (dtype == PK_F16) { if (hd == 64) PK_PR(f16, 64); else PK_PR(f16, 128); }
    else { if (hd == 64) PK_PR(float, 64); else PK_PR(float, 128); }
#undef PK_PR
    PK_LAUNCH_CHECK();
    return 0;
}
