// Shared device/host helpers for the pasero_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#define PK_F32 0
#define PK_BF16 1
#define PK_F16 2

// activation ids (pasero/models/modules.py:220-228)
// (include/pasero_hip.h: deferred LayerNorm parameter gradients)
#define PK_LN_GROUP_MAX 4
struct PkLnParamGrad {
    const void* workspace;
    void* dgamma;
    void* dbeta;
};
#define PK_GEMM_PAD_N 1 /* pk_gemm_ex promises (include/pasero_hip.h) */
#define PK_GEMM_PAD_K 2
#define PK_ACT_NONE 0
#define PK_ACT_RELU 1
#define PK_ACT_GELU 2       // erf
#define PK_ACT_GELU_TANH 3
#define PK_ACT_SILU 4

typedef __hip_bfloat16 bf16;
typedef _Float16 f16;  // IEEE half: the reference's default training dtype (config.py:518-523)
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA bf16 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" void pk_set_error(const char* fmt, ...);

#define PK_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            pk_set_error(__VA_ARGS__);          \
            return -1;                          \
        }                                       \
    } while (0)

#define PK_LAUNCH_CHECK()                                                        \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            pk_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return (int)e_;                                                      \
        }                                                                        \
    } while (0)

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
// round-to-nearest-even f32 -> bf16 through the compiler's cast (keeps NaN a NaN, v_cvt_pk_bf16_f32 at -O3)
__device__ __forceinline__ unsigned short f2bf(float f) {
    bf16 b = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short*>(&b);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }

// The two 16-bit storage types share every kernel: same bytes in memory, LDS and MFMA fragments; they differ in the
// float <-> 16-bit conversion and in the MFMA instruction (v_mfma_f32_32x32x16_bf16 / _f16).
template <typename T> struct H16;
template <> struct H16<bf16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 vec;
    static __device__ __forceinline__ unsigned short bits(float f) { return f2bf(f); }
    static __device__ __forceinline__ float val(unsigned short u) { return bf2f(u); }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct H16<f16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 vec;
    static __device__ __forceinline__ unsigned short bits(float f) { return __builtin_bit_cast(unsigned short, (f16)f); }
    static __device__ __forceinline__ float val(unsigned short u) { return (float)__builtin_bit_cast(f16, u); }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// ---- 16-byte vector access: VEC<T> elements per 16 B ----
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float4 raw;
    __device__ __forceinline__ float get(int i) const { return (&raw.x)[i]; }
    __device__ __forceinline__ void set(int i, float v) { (&raw.x)[i] = v; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ float get(int i) const {
        unsigned w = (&raw.x)[i >> 1];
        return bf2f((unsigned short)((i & 1) ? (w >> 16) : (w & 0xffff)));
    }
    __device__ __forceinline__ void set(int i, float v) {
        unsigned& w = (&raw.x)[i >> 1];
        unsigned b = f2bf(v);
        w = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
    }
};
template <> struct Vec16<f16> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ float get(int i) const {
        unsigned w = (&raw.x)[i >> 1];
        return H16<f16>::val((unsigned short)((i & 1) ? (w >> 16) : (w & 0xffff)));
    }
    __device__ __forceinline__ void set(int i, float v) {
        unsigned& w = (&raw.x)[i >> 1];
        unsigned b = H16<f16>::bits(v);
        w = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
    }
};
// a 16-byte chunk from its elements as floats.  (Vec16::set inserts ONE element: a conversion, a shift and a masked merge
// each — three instructions per element where v_cvt_pk_bf16_f32 converts and packs two; the cross-entropy gradient spent
// 390 instructions per 96 elements on it.)
template <typename T> __device__ __forceinline__ Vec16<T> vec16_pack(const float (&f)[16 / sizeof(T)]) {
    Vec16<T> o;
    if constexpr (sizeof(T) == 4) {
        o.raw = float4{f[0], f[1], f[2], f[3]};
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) (&o.raw.x)[j] = (unsigned)H16<T>::bits(f[2 * j]) | ((unsigned)H16<T>::bits(f[2 * j + 1]) << 16);
    }
    return o;
}
template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
    Vec16<T> v;
    v.raw = *reinterpret_cast<const decltype(v.raw)*>(p);
    return v;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
    *reinterpret_cast<decltype(v.raw)*>(p) = v.raw;
}
// streaming store (`nt`): for outputs no later instruction of this kernel reads — the next kernel finds them in HBM /
// Infinity Cache either way, and the L2 keeps the operand tiles that ARE re-read.  Measured on the 256-tile GEMM
// epilogue: C2 step 18.74 -> 18.41 ms (same box, 4 alternations).
template <typename T> __device__ __forceinline__ void store16_nt(T* p, const Vec16<T>& v) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v.raw), reinterpret_cast<u32x4*>(p));
}

// run `...` with T bound to the storage type of `dtype`
#define PK_DTYPE_SWITCH(dtype, who, ...)                                   \
    if (dtype == PK_BF16) { using T = bf16; __VA_ARGS__ }                  \
    else if (dtype == PK_F16) { using T = f16; __VA_ARGS__ }               \
    else if (dtype == PK_F32) { using T = float; __VA_ARGS__ }             \
    else { PK_CHECK_ARG(false, "%s: dtype %d not supported", who, dtype); }

// ---- wave (64 lanes) reductions ----
// A butterfly over the partners 32, 16, 8, 4, 2, 1 lanes away, in this order; every lane holds the result.  Written
// with `__shfl_xor` each step compiles to ds_bpermute_b32 plus a compare / select for its width argument — 4
// instructions and an LDS round trip, 12 dependent round trips for the mean and the variance of a LayerNorm row.
// gfx950's v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows of two registers; within
// a row of 16 lanes DPP rotations and quad permutes reach the partner (after the step "8" the values repeat with period
// 8, so a rotation by 4 finds the lane ^ 4 partner's value).  10 VALU instructions per reduction and bit for bit the
// `__shfl_xor` butterfly's result (tools/wave_reduce_check.hip, run by the GPU tests: 4096 random waves, no mismatch): the
// LayerNorm backward of C2 37.0 -> 28.5 us together with its specialisation (layernorm.hip).  All 64 lanes must be
// active.  (The swaps are inline asm: given the same register twice, the builtin's two results are folded into one by
// hipcc 7.2.)
template <int CTRL> __device__ __forceinline__ float dpp_mov_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
#define PK_WAVE_BUTTERFLY(OP)                                                         \
    float a = v, b = v;                                                               \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));       \
    v = OP(a, b);                                                                     \
    a = v; b = v;                                                                     \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));       \
    v = OP(a, b);                                                                     \
    v = OP(v, dpp_mov_f32<0x128>(v)); /* row_ror:8 */                                 \
    v = OP(v, dpp_mov_f32<0x124>(v)); /* row_ror:4 */                                 \
    v = OP(v, dpp_mov_f32<0x4E>(v));  /* quad_perm:[2,3,0,1] */                       \
    v = OP(v, dpp_mov_f32<0xB1>(v));  /* quad_perm:[1,0,3,2] */                       \
    return v;
__device__ __forceinline__ float pk_add_f32(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) { PK_WAVE_BUTTERFLY(pk_add_f32) }
__device__ __forceinline__ float wave_max(float v) { PK_WAVE_BUTTERFLY(fmaxf) }
#undef PK_WAVE_BUTTERFLY
// max over the lane and its partner 32 lanes away (the two half-waves of a 32 x 32 MFMA tile hold the same rows): one
// swap instead of the LDS round trip of `__shfl_xor(v, 32)` — it sits on the serial chain of an attention tile
__device__ __forceinline__ float half_wave_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
// sum over the 8 consecutive lanes a lane belongs to (partners 1, 2, 4 away, in this order, as three `__shfl_xor` steps
// would add them): quad permutes, then the half-row mirror (lane ^ 7 — its quad already holds the quad's sum)
__device__ __forceinline__ float lanes8_sum(float v) {
    v += dpp_mov_f32<0xB1>(v);   // quad_perm:[1,0,3,2]
    v += dpp_mov_f32<0x4E>(v);   // quad_perm:[2,3,0,1]
    v += dpp_mov_f32<0x141>(v);  // row_half_mirror
    return v;
}
// exchange between the two half-waves: afterwards lanes 32..63 of `a` hold what lanes 0..31 of `b` held and vice versa
// (v_permlane32_swap_b32; the other halves stay)
__device__ __forceinline__ void half_wave_swap(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// (the LDS-pipe form, kept for the kernels' tests of the above)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- activations ----
__device__ __forceinline__ float act_fwd(int act, float x) {
    switch (act) {
        case PK_ACT_RELU: return fmaxf(x, 0.f);
        case PK_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case PK_ACT_GELU_TANH: {
            float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.f + tanhf(u));
        }
        case PK_ACT_SILU: return x / (1.f + __expf(-x));
        default: return x;
    }
}
// derivative w.r.t. the pre-activation x (for RELU `x` may also be the post-activation: sign is the same)
__device__ __forceinline__ float act_bwd(int act, float x) {
    switch (act) {
        case PK_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case PK_ACT_GELU: {
            float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
            float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case PK_ACT_GELU_TANH: {
            float x2 = x * x;
            float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
            float t = tanhf(u);
            float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * x2);
            return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
        }
        case PK_ACT_SILU: {
            float s = 1.f / (1.f + __expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        default: return 1.f;
    }
}

// ---- the same activations for 16-bit epilogues: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// 2^-8 / 2^-11 of the storage types), tanh and the logistic through one v_exp_f32 and one v_rcp_f32.  The library erff /
// tanhf cost 40-60 VALU instructions per element: in the epilogue of the C4 feed-forward GEMMs (GELU, 24 000 x 2048
// outputs) that was as long as the GEMM itself (157 us at 308 TFLOP/s; gelu' in the dH GEMM: 196 us at 257).  The fp32
// kernels keep the exact forms above (parity with the reference to 1e-6).
// (v_rcp_f32, 1 ulp: `__frcp_rn` is a correctly rounded division — v_div_scale / v_rcp / four fma / v_div_fmas / v_div_fixup,
// ten instructions per call, a third of GELU's cost in an epilogue)
__device__ __forceinline__ float pk_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float act_fwd_fast(int act, float x) {
    switch (act) {
        case PK_ACT_RELU: return fmaxf(x, 0.f);
        case PK_ACT_GELU: {
            const float ax = fabsf(x) * 0.70710678118654752f;
            const float t = pk_rcp(fmaf(0.3275911f, ax, 1.f));
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            const float erf_abs = 1.f - poly * __expf(-ax * ax);
            return 0.5f * x * (1.f + copysignf(erf_abs, x));
        }
        case PK_ACT_GELU_TANH: {
            const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            const float th = 1.f - 2.f * pk_rcp(1.f + __expf(2.f * u));  // exp -> inf: 1; exp -> 0: -1
            return 0.5f * x * (1.f + th);
        }
        case PK_ACT_SILU: return x * pk_rcp(1.f + __expf(-x));
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd_fast(int act, float x) {
    switch (act) {
        case PK_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case PK_ACT_GELU: {  // cdf + x pdf: erf(x / sqrt 2) and the density share exp(-x^2 / 2)
            const float ax = fabsf(x) * 0.70710678118654752f;
            const float t = pk_rcp(fmaf(0.3275911f, ax, 1.f));
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            const float e = __expf(-ax * ax);
            const float cdf = 0.5f * (1.f + copysignf(1.f - poly * e, x));
            return cdf + x * 0.3989422804014327f * e;
        }
        case PK_ACT_GELU_TANH: {
            const float x2 = x * x;
            const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
            const float t = 1.f - 2.f * pk_rcp(1.f + __expf(2.f * u));
            const float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * x2);
            return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
        }
        case PK_ACT_SILU: {
            const float s = pk_rcp(1.f + __expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        default: return 1.f;
    }
}

// every kernel of a 16-bit model evaluates the activation with the SAME function (a GEMM may run on one kernel for a
// batch and on another for half of it: with two evaluations of GELU the halves of a batch no longer add up bit for bit)
template <typename T> __device__ __forceinline__ float act_fwd_t(int act, float x) {
    if constexpr (sizeof(T) == 2) return act_fwd_fast(act, x);
    else return act_fwd(act, x);
}
template <typename T> __device__ __forceinline__ float act_bwd_t(int act, float x) {
    if constexpr (sizeof(T) == 2) return act_bwd_fast(act, x);
    else return act_bwd(act, x);
}

// ---- Philox4x32-10 counter RNG for dropout: mask is a pure function of (seed, offset, element index), so the
// backward pass regenerates it instead of storing it ----
struct Philox4 {
    unsigned x, y, z, w;
};
// one round (of ten); k0 / k1: the round's keys = seed halves + round * (0x9E3779B9, 0xBB67AE85)
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    // (one v_bitop3_b32 each, truth table 0x96 = a ^ b ^ c: hipcc 7.2 emits two v_xor_b32 for the plain expression)
    const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c1, k0, 0x96);
    const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c3, k1, 0x96);
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
}
// rounds [R0, R1) on a state kept by the caller (gemmln.hip spreads the ten rounds of a draw over two phases of its K loop)
template <int R0, int R1>
__device__ __forceinline__ void philox_rounds(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned long long seed) {
#pragma unroll
    for (int r = R0; r < R1; ++r)
        philox_round(c0, c1, c2, c3, (unsigned)seed + (unsigned)r * 0x9E3779B9u, (unsigned)(seed >> 32) + (unsigned)r * 0xBB67AE85u);
}
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long seed, unsigned long long offset,
                                                 unsigned long long idx) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    philox_rounds<0, 10>(c0, c1, c2, c3, seed);
    return Philox4{c0, c1, c2, c3};
}
// keep-threshold for drop probability p: element kept iff rnd >= thr
__host__ __device__ __forceinline__ unsigned dropout_threshold(float p) {
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
}
// The mask: element i of a flat tensor is kept iff  u16[i & 7] >= thr >> 16,  where u16[0..7] are the eight 16-bit halves
// (low half first) of the four words of philox4x32_10(seed, offset, i >> 3).  One Philox evaluation serves eight elements
// = one 16-byte chunk of a 16-bit tensor (the evaluation is ~60 VALU instructions: with a 32-bit draw per element it
// was the largest single cost of the LayerNorm epilogues); the drop probability is quantised to 1 / 65536.
__device__ __forceinline__ void dropout_keep8(unsigned long long seed, unsigned long long offset,
                                              unsigned long long q8, unsigned thr, bool keep[8]) {
    const Philox4 r = philox4x32_10(seed, offset, q8);
    const unsigned t = thr >> 16;
    keep[0] = (r.x & 0xffffu) >= t; keep[1] = (r.x >> 16) >= t; keep[2] = (r.y & 0xffffu) >= t; keep[3] = (r.y >> 16) >= t;
    keep[4] = (r.z & 0xffffu) >= t; keep[5] = (r.z >> 16) >= t; keep[6] = (r.w & 0xffffu) >= t; keep[7] = (r.w >> 16) >= t;
}
// 4 keep flags for elements [4*q, 4*q+4) of a flat tensor (one half of dropout_keep8's draw)
__device__ __forceinline__ void dropout_keep4(unsigned long long seed, unsigned long long offset,
                                              unsigned long long q, unsigned thr, bool keep[4]) {
    const Philox4 r = philox4x32_10(seed, offset, q >> 1);
    const unsigned a = (q & 1) ? r.z : r.x, b = (q & 1) ? r.w : r.y, t = thr >> 16;
    keep[0] = (a & 0xffffu) >= t; keep[1] = (a >> 16) >= t; keep[2] = (b & 0xffffu) >= t; keep[3] = (b >> 16) >= t;
}
// one element
__device__ __forceinline__ bool dropout_keep1(unsigned long long seed, unsigned long long offset,
                                              unsigned long long i, unsigned thr) {
    const Philox4 r = philox4x32_10(seed, offset, i >> 3);
    const unsigned j = (unsigned)(i & 7), w = (j >> 1) == 0 ? r.x : (j >> 1) == 1 ? r.y : (j >> 1) == 2 ? r.z : r.w;
    return ((j & 1) ? (w >> 16) : (w & 0xffffu)) >= (thr >> 16);
}
// the EPV (4 or 8) elements of the 16-byte chunk that starts at element `elem0` (a multiple of EPV)
template <int EPV>
__device__ __forceinline__ void dropout_keep_chunk(unsigned long long seed, unsigned long long offset,
                                                   unsigned long long elem0, unsigned thr, bool* keep) {
    static_assert(EPV == 4 || EPV == 8, "16-byte chunks of 4- or 2-byte elements");
    if constexpr (EPV == 8) dropout_keep8(seed, offset, elem0 >> 3, thr, keep);
    else dropout_keep4(seed, offset, elem0 >> 2, thr, keep);
}
