// Shared pieces of the attention kernels (attention.hip, attention_long.hip): parameters, LDS tile images, fragment loads,
// rotary helpers, row stores.  Everything but AttnParams lives in an anonymous namespace (one copy per translation unit).
#pragma once
#include "common.h"

// Register budgets: the minimum number of waves per SIMD the compiler must leave room for (512 / n registers per lane).
// Left to itself it takes 176-316 registers for these kernels and halves the occupancy for nothing: at these budgets
// none of them spills.  Measured (bf16, B = 256, H = 8, T = S = 128): forward 41.5 -> 33.6 us; Whisper encoder shape
// (T = S = 1500): forward 181 -> 148 us, backward (dQ + dK/dV kernels) 546 -> 407 us.  The fused backward spills 27
// registers at three waves and stays at two; the dQ kernel for heads of 128 spills at two and stays at one.
#ifndef PK_ATTN_FWD_WAVES
#define PK_ATTN_FWD_WAVES 3
#endif
#ifndef PK_ATTN_DQ_WAVES
#define PK_ATTN_DQ_WAVES 2
#endif
#ifndef PK_ATTN_FUSED_WAVES
#define PK_ATTN_FUSED_WAVES 2
#endif
constexpr int q_min_waves(int mode, int hd) { return mode == 0 ? (hd == 64 ? PK_ATTN_FWD_WAVES : 2) : (hd == 64 ? PK_ATTN_DQ_WAVES : 1); }
constexpr int dkv_min_waves(int hd) { return hd == 64 ? 2 : 1; }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
// 16-byte global loads / stores of the attention kernels; PKATT_NT (diagnostic builds): 1 = the fused backward's tile loads
// streaming, 2 = fragment / tile loads of every kernel, 4 = the row stores (o, dq, dk, dv)
#ifndef PKATT_NT
#define PKATT_NT 0
#endif
template <int BIT, typename T> __device__ __forceinline__ uint4 att_ld(const T* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    if constexpr ((PKATT_NT & BIT) != 0) return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
    else return *reinterpret_cast<const uint4*>(p);
}
// v or zeros (component by component: a `c ? a : b` on whole uint4 objects selects between their ADDRESSES and parks the
// operands in scratch memory)
__device__ __forceinline__ uint4 sel4(bool c, const uint4 v) { return make_uint4(c ? v.x : 0u, c ? v.y : 0u, c ? v.z : 0u, c ? v.w : 0u); }
template <typename T> __device__ __forceinline__ void att_st(T* p, uint4 v) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    if constexpr ((PKATT_NT & 4) != 0) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<uint4*>(p) = v;
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s4;

namespace pkattn {
struct AttnParams {
    int B, H, T, S;
    long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;  // batch / row strides in elements (head stride = 64)
    long long do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    const unsigned char* key_pad;  // [B][S] or null
    int causal;
    float scale;
    // attention-probability dropout (F.scaled_dot_product_attention(dropout_p=...), modules.py:707-720): the forward pass
    // draws keep bits from Philox (element index = row * 8 * mask_pitch + key) and stores them, one bit per (query, key),
    // rows of `mask_pitch` = 8 * ceil(S / 64) bytes; the backward kernels read the bits back
    unsigned drop_thr;             // 0: no dropout
    float drop_scale;              // 1 / (1 - p)
    unsigned long long seed, offset;
    unsigned char* drop_mask;      // [B][H][T][mask_pitch]
    long long mask_pitch;
    // rotary positions folded into the kernels (RotaryEmbedding.forward, pasero/models/modules.py:982-1025, applied to q and
    // k at modules.py:617-623): q and k arrive UNROTATED, every kernel rotates the rows it loads (query t by the angle of
    // position rope_q0 + t, key s by rope_k0 + s) and the backward kernels rotate dQ / dK back as they leave — the gradients
    // are those of the unrotated projection.  cos / sin: fp32 [rope_max][head_dim / 2]; rope_cos == NULL: no rotation.
    const float* rope_cos;
    const float* rope_sin;
    int rope_max, rope_q0, rope_k0;
    int nqb, nkb;  // 128-row query / key blocks per (batch, head) pair: the 1-D grids of the 16-bit kernels (pair_block)
};
}  // namespace pkattn
using pkattn::AttnParams;

namespace {

constexpr int HD64 = 64;  // the single-workgroup fused backward is built for this head dimension only
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;


// 1-D grid of the tiled 16-bit kernels: workgroup `lin` -> (batch b, head h, block blk of nblk); the hardware deals workgroups
// to the eight XCDs round-robin, so lin % 8 is the XCD and all nblk blocks of a pair sit on one (pairs beyond B H: false)
__device__ __forceinline__ bool pair_block(const AttnParams& p, int nblk, int& b, int& h, int& blk) {
    const int lin = blockIdx.x, j = lin >> 3;
    const int pair = (j / nblk) * 8 + (lin & 7);
    blk = j % nblk;
    if (pair >= p.B * p.H) return false;
    b = pair / p.H;
    h = pair % p.H;
    return true;
}

__device__ __forceinline__ bool drop_keep1(const AttnParams& p, long long row, int s) {
    const unsigned long long idx = (unsigned long long)row * 8ull * p.mask_pitch + s;
    return dropout_keep1(p.seed, p.offset, idx, p.drop_thr);
}
__device__ __forceinline__ bool drop_bit(const AttnParams& p, long long row, int s) {
    return (p.drop_mask[row * p.mask_pitch + (s >> 3)] >> (s & 7)) & 1;
}

__device__ __forceinline__ bool key_masked(const AttnParams& p, int b, int t, int s) {
    if (s >= p.S) return true;
    if (p.key_pad && p.key_pad[(long long)b * p.S + s]) return true;
    if (p.causal && s > t + (p.S - p.T)) return true;
    return false;
}

// ---- rotary helpers (GPT-J halves: y[i] = x[i] c_i - x[i + hd/2] s_i ; y[i + hd/2] = x[i + hd/2] c_i + x[i] s_i) ----
__device__ __forceinline__ int rope_row(const AttnParams& p, int pos) { return min(max(pos, 0), p.rope_max - 1); }
// a whole fp32 row in registers, in place; INV: the transposed rotation (gradients)
template <int HD, bool INV>
__device__ __forceinline__ void rope_row_f32(float (&x)[HD], const AttnParams& p, int pos) {
    const float* cs = p.rope_cos + (long long)rope_row(p, pos) * (HD / 2);
    const float* sn = p.rope_sin + (long long)rope_row(p, pos) * (HD / 2);
#pragma unroll
    for (int i = 0; i < HD / 2; ++i) {
        const float c = cs[i], s = INV ? -sn[i] : sn[i];
        const float a = x[i], b = x[i + HD / 2];
        x[i] = a * c - b * s;
        x[i + HD / 2] = b * c + a * s;
    }
}
// one element of a row read from memory: element d of head row `base` at position pos
template <int HD>
__device__ __forceinline__ float rope_elem_f32(const float* __restrict__ base, int d, const AttnParams& p, int pos) {
    const int i = d & (HD / 2 - 1);
    const long long r = (long long)rope_row(p, pos) * (HD / 2) + i;
    const float c = p.rope_cos[r], s = p.rope_sin[r];
    return d < HD / 2 ? base[d] * c - base[d + HD / 2] * s : base[d] * c + base[d - HD / 2] * s;
}
// 8 consecutive 16-bit elements of a head row (`own`) and the 8 of the other half (`oth`): own, rotated.  cs / sn point at
// the 8 angles; upper: own is the second half; inverse: rotate back.  fp32 arithmetic, rounded once (as pk_rope does).
template <typename T>
__device__ __forceinline__ uint4 rope8(uint4 own, uint4 oth, const float* __restrict__ cs, const float* __restrict__ sn,
                                        bool upper, bool inverse) {
    const float4 c0 = *reinterpret_cast<const float4*>(cs), c1 = *reinterpret_cast<const float4*>(cs + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sn), s1 = *reinterpret_cast<const float4*>(sn + 4);
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sg = (upper != inverse) ? 1.f : -1.f;  // forward: lower a c - b s, upper a c + b s
    const unsigned ow[4] = {own.x, own.y, own.z, own.w}, ot[4] = {oth.x, oth.y, oth.z, oth.w};
    unsigned out[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float a0 = H16<T>::val((unsigned short)(ow[w] & 0xffff)), a1 = H16<T>::val((unsigned short)(ow[w] >> 16));
        const float b0 = H16<T>::val((unsigned short)(ot[w] & 0xffff)), b1 = H16<T>::val((unsigned short)(ot[w] >> 16));
        const float y0 = a0 * c[2 * w] + sg * b0 * sv[2 * w], y1 = a1 * c[2 * w + 1] + sg * b1 * sv[2 * w + 1];
        out[w] = (unsigned)H16<T>::bits(y0) | ((unsigned)H16<T>::bits(y1) << 16);
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

// =====================================================================================================
// bf16 MFMA path
// =====================================================================================================
// LDS images of a [64 rows][64 x bf16] tile.  `P` > 0: plain rows of P bytes.
constexpr int PITCH = 144;   // +16 B pad: ds_read_b128 row reads conflict-free (tiles that are only read by rows)
constexpr int VPITCH = 192;  // V tile of the forward pass: only transposed reads (4 key rows on distinct bank quarters)
constexpr int DUAL = 0;      // tiles read BOTH by rows (ds_read_b128) and transposed (ds_read_b64_tr_b16): 8-row x 32-col
                             // subtiles of 512 B with the 16-B chunk XOR-swizzled by (row>>2)&3 — both kinds of read are
                             // conflict-free (cdna guide T10 image (a), cut down to 128-B rows)
template <int P> __device__ __forceinline__ int lds_off(int row, int ch) {
    if constexpr (P == DUAL)
        return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
    else
        return row * P + ch * 16;
}
constexpr int KT = 64;       // rows (keys or queries) staged per LDS tile
// T = bf16 or f16: fragments travel as raw 8 x 16-bit vectors; only the MFMA instruction and the conversions differ
template <typename T>
__device__ __forceinline__ f32x16 mm(bf16x8_t a, bf16x8_t b, f32x16 c) {
    typedef typename H16<T>::vec V;
    return H16<T>::mfma(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c);
}
// A [64 rows][head_dim] tile is head_dim / 64 images side by side (each 64 columns wide, laid out as above)
template <int P> constexpr int img_bytes() { return KT * (P == DUAL ? 128 : P); }

// stage a [64 rows][64 cols] bf16 tile (rows r0.., row limit `lim`, zero fill) into LDS with the given pitch
template <typename T>
__device__ __forceinline__ void stage_tile(char* lds, int pitch, const T* __restrict__ base, long long rs, int r0,
                                           int lim, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = tid + i * 256;  // 512 chunks of 16 B
        int r = c >> 3, cc = (c & 7) * 8;
        uint4 val = {0, 0, 0, 0};
        if (r0 + r < lim) val = *reinterpret_cast<const uint4*>(base + (long long)(r0 + r) * rs + cc);
        *reinterpret_cast<uint4*>(lds + r * pitch + cc * 2) = val;
    }
}

// the same in two halves, so the next tile's global loads fly under the current tile's MFMAs
template <int NR, typename T>  // NR = 2 * head_dim / 64 chunks of 16 B per thread
__device__ __forceinline__ void tile_g2r(uint4 (&regs)[NR], const T* __restrict__ base, long long rs, int r0, int lim,
                                         int tid) {
    constexpr int CPR = 4 * NR;  // 16-B chunks per row
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int c = tid + i * 256;
        int r = c / CPR, cc = (c % CPR) * 8;
        // (no branch around the load — rows past `lim` read the last row and are zeroed: at the join behind a branch hipcc
        // waits for EVERY load in flight, the caller's prefetches included)
        const uint4 got = att_ld<2>(base + (long long)min(r0 + r, lim - 1) * rs + cc);
        regs[i] = sel4(r0 + r < lim, got);
    }
}
template <int P, int NR>
__device__ __forceinline__ void tile_r2s(const uint4 (&regs)[NR], char* lds, int tid) {
    constexpr int CPR = 4 * NR;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int c = tid + i * 256, ch = c % CPR;
        *reinterpret_cast<uint4*>(lds + (ch >> 3) * img_bytes<P>() + lds_off<P>(c / CPR, ch & 7)) = regs[i];
    }
}

// row fragment: lane (r = l&31, h = l>>5) gets row row0 + r, elements d = 16*kk + 8*h .. +7
template <int P>
__device__ __forceinline__ bf16x8_t row_frag(const char* lds, int row0, int kk, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(lds + (kk >> 2) * img_bytes<P>() +
                                              lds_off<P>(row0 + (lane & 31), (kk & 3) * 2 + (lane >> 5)));
}
// transposed fragment for "accumulator tile as next operand" products (cdna guide §3): lane (r, h) gets column
// c0 + r of rows  row0 + 16*s + 8*(j>>2) + 4*h + (j&3),  j = 0..7
template <int P>
__device__ __forceinline__ bf16x8_t tr_frag(const char* lds, int row0, int s, int c0, int lane) {
    int q = (lane & 15) >> 2, p4 = lane & 3;
    int col = (c0 & 63) + 16 * ((lane >> 4) & 1) + 4 * p4;
    int row = row0 + 16 * s + 4 * (lane >> 5) + q;
    lds += (c0 >> 6) * img_bytes<P>();  // the 64-column image this d-tile lives in
    const char* ptr = lds + lds_off<P>(row, col >> 3) + (col & 7) * 2;
    const char* ptr8 = lds + lds_off<P>(row + 8, col >> 3) + (col & 7) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)ptr);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)ptr8);
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, f);
}
// accumulator registers 8s..8s+7 of a 32x32 tile -> bf16 B/A operand of k-step s
template <typename T>
__device__ __forceinline__ bf16x8_t acc_frag(const f32x16& a, int s) {
    s16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)H16<T>::bits(a[8 * s + j]);
    return __builtin_bit_cast(bf16x8_t, f);
}
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// global row fragment (rows beyond `lim` read as zero): lane (r, h) row row0 + r, d = 16kk + 8h .. +7
template <int NF, typename T>
__device__ __forceinline__ void load_row_frags(bf16x8_t (&f)[NF], const T* __restrict__ base, long long rs, int row,
                                               bool valid, int lane) {
#pragma unroll
    for (int kk = 0; kk < NF; ++kk) {
        const uint4 got = att_ld<2>(base + (long long)(valid ? row : 0) * rs + kk * 16 + 8 * (lane >> 5));  // (branch-free, as above)
        const uint4 val = sel4(valid, got);
        f[kk] = __builtin_bit_cast(bf16x8_t, val);
    }
}

// ---- rotary positions on the operands as they are loaded (AttnParams::rope_cos) ----
// row fragments (load_row_frags): lane (r, h) holds chunk 2 kk + h of its row for every kk; the partner chunk hd/2
// elements away is fragment kk + NF/2 of the SAME lane
template <int NF, typename T>
__device__ __forceinline__ void rope_frags(bf16x8_t (&f)[NF], const AttnParams& p, int pos, int lane) {
    constexpr int HALF = NF * 8;  // head_dim / 2
    const float* cs = p.rope_cos + (long long)rope_row(p, pos) * HALF + 8 * (lane >> 5);
    const float* sn = p.rope_sin + (long long)rope_row(p, pos) * HALF + 8 * (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < NF / 2; ++kk) {
        const uint4 lo = __builtin_bit_cast(uint4, f[kk]), hi = __builtin_bit_cast(uint4, f[kk + NF / 2]);
        f[kk] = __builtin_bit_cast(bf16x8_t, rope8<T>(lo, hi, cs + 16 * kk, sn + 16 * kk, false, false));
        f[kk + NF / 2] = __builtin_bit_cast(bf16x8_t, rope8<T>(hi, lo, cs + 16 * kk, sn + 16 * kk, true, false));
    }
}
// one 16-byte chunk `ch` (of CPR per row) of row `row` held by this lane: the partner chunk CPR / 2 away sits CPR / 2 lanes
// away (tile_g2r and the fused backward's loads walk a row with consecutive lanes)
template <int CPR, typename T>
__device__ __forceinline__ uint4 rope_chunk(uint4 v, int ch, const AttnParams& p, int pos) {
    constexpr int HC = CPR / 2;
    uint4 o;
    o.x = __shfl_xor(v.x, HC, 64); o.y = __shfl_xor(v.y, HC, 64); o.z = __shfl_xor(v.z, HC, 64); o.w = __shfl_xor(v.w, HC, 64);
    const long long r = (long long)rope_row(p, pos) * (CPR * 4) + 8 * (ch & (HC - 1));
    return rope8<T>(v, o, p.rope_cos + r, p.rope_sin + r, ch >= HC, false);
}
template <int NR, typename T>
__device__ __forceinline__ void rope_tile(uint4 (&regs)[NR], const AttnParams& p, int pos0, int tid) {
    constexpr int CPR = 4 * NR;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = tid + i * 256;
        regs[i] = rope_chunk<CPR, T>(regs[i], c % CPR, p, pos0 + c / CPR);
    }
}
// accumulators dQᵀ / dKᵀ [d-tile][d rows] x row-on-lane, rotated BACK in place (register 4g + j of d-tile dt is element
// d = 32 dt + 8 g + 4 (l >> 5) + j of the lane's row; its partner is the same register of d-tile dt + ND / 2)
template <int ND>
__device__ __forceinline__ void rope_acc_inverse(f32x16 (&acc)[ND], const AttnParams& p, int pos, int lane) {
    constexpr int HALF = ND * 16;  // head_dim / 2
    const float* cs = p.rope_cos + (long long)rope_row(p, pos) * HALF + 4 * (lane >> 5);
    const float* sn = p.rope_sin + (long long)rope_row(p, pos) * HALF + 4 * (lane >> 5);
#pragma unroll
    for (int dt = 0; dt < ND / 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 c4 = *reinterpret_cast<const float4*>(cs + 32 * dt + 8 * g);
            const float4 s4 = *reinterpret_cast<const float4*>(sn + 32 * dt + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = (&c4.x)[j], sv = (&s4.x)[j];
                const float a = acc[dt][4 * g + j], b = acc[dt + ND / 2][4 * g + j];
                acc[dt][4 * g + j] = a * c + b * sv;
                acc[dt + ND / 2][4 * g + j] = b * c - a * sv;
            }
        }
}

template <typename T>
__device__ __forceinline__ float frag_dot(const bf16x8_t& a, const bf16x8_t& b) {
    s16x8 x = __builtin_bit_cast(s16x8, a), y = __builtin_bit_cast(s16x8, b);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += H16<T>::val((unsigned short)x[j]) * H16<T>::val((unsigned short)y[j]);
    return s;
}

// write a transposed accumulator pair Xᵀ[d][row-on-lane] (2 d-tiles) as bf16 rows: lane (r, h) owns row `row`
template <int ND, typename T>
__device__ __forceinline__ void store_rowT(T* __restrict__ base, long long rs, int row, bool valid,
                                           const f32x16 (&acc)[ND], float mul, int lane) {
    // lanes l and l + 32 own the same row: columns [8g, 8g+4) and [8g+4, 8g+8) of every group g.  They swap one piece
    // per pair of groups so that each writes 16 contiguous bytes (lane l: group 2j whole, lane l + 32: group 2j + 1)
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned lo[2], hi[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int g = 2 * j + q;
                lo[q] = (unsigned)H16<T>::bits(acc[dt][4 * g] * mul) | ((unsigned)H16<T>::bits(acc[dt][4 * g + 1] * mul) << 16);
                hi[q] = (unsigned)H16<T>::bits(acc[dt][4 * g + 2] * mul) | ((unsigned)H16<T>::bits(acc[dt][4 * g + 3] * mul) << 16);
            }
            // h = 0: group 2j = {own cols 0-3, partner's cols 4-7};  h = 1: group 2j + 1 = {partner's cols 0-3, own cols 4-7}:
            // the upper half-wave's [0] pieces change places with the lower half-wave's [1] pieces (one swap instruction per
            // register; as two `__shfl_xor(.., 32)` this was 56 LDS round trips in the fused backward's epilogue)
            half_wave_swap(lo[0], lo[1]);
            half_wave_swap(hi[0], hi[1]);
            const uint4 v = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            const int d = dt * 32 + 8 * (2 * j + h);
            if (valid) att_st(base + (long long)row * rs + d, v);
        }
}

}  // namespace

// attention_long.hip: the long-key-sequence forward (heads of 64, 16-bit types, no causal mask, no rotation); returns a HIP error code
int pk_attn_fwd_long_launch(const void* q, const void* k, const void* v, void* o, float* lse, const pkattn::AttnParams& p, int dtype,
                            hipStream_t stream);
// the dQ pass of the same shapes (also writes delta for the dK / dV kernel that follows)
int pk_attn_dq_long_launch(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                           float* delta, void* dq, const pkattn::AttnParams& p, int dtype, hipStream_t stream);
// the dK / dV pass for long QUERY sequences (T >= the same threshold)
int pk_attn_dkv_long_launch(const void* q, const void* k, const void* v, const void* d_o, const float* lse, const float* delta,
                            void* dk, void* dv, const pkattn::AttnParams& p, int dtype, hipStream_t stream);
