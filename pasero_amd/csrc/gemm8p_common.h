// Building blocks shared by the phase-interleaved GEMM kernels (gemm8p.hip: 256 x 256 tile; gemmln.hip: 128 x 512 tile with
// the residual + dropout + LayerNorm epilogue): MFMA traits, the swizzled 16-KiB half-tile LDS image, the per-lane source
// offsets of its LDS-DMA pieces, row-form fragment reads, the XCD-contiguous workgroup remap.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <typename T> struct M16;
template <> struct M16<bf16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 vec;
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct M16<f16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 vec;
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// one half-tile image: row form [128 rows][64 k] (128-B rows), col form [64 k][128 m] (256-B rows)
template <bool COL> struct HT {
    static constexpr int ROWB = COL ? 256 : 128;
    __device__ static __forceinline__ int swz(int row) {
        return COL ? (((row & 3) | (((row >> 3) & 1) << 2)) << 1) : ((row >> 1) & 7);
    }
    __device__ static __forceinline__ int offset(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};

// per-lane byte offset (into the operand's buffer, K offset excluded) of the 16 bytes this lane's DMA piece `piece`
// (0..15, 1 KiB each) of half-tile `h` brings in; `ld` in elements
// Past the edge of the matrix (`lim` rows of a row-form operand, `lim` columns of a col-form one) a VALID row / column
// is re-read instead: it feeds only outputs that are never stored, and no access leaves the buffer whatever the
// descriptor's range check covers.
template <bool COL>
__device__ __forceinline__ unsigned src_offset(int piece, int lane, long long ld, long long r0, long long lim) {
    if constexpr (!COL) {  // 8 rows x 128 B per piece
        const int row = piece * 8 + (lane >> 3), chunk = (lane & 7) ^ HT<false>::swz(row);
        const long long gr = min(r0 + row, lim - 1);
        return (unsigned)((gr * ld + chunk * 8) * 2);
    } else {  // 4 k-rows x 256 B per piece
        const int krow = piece * 4 + (lane >> 4), chunk = (lane & 15) ^ HT<true>::swz(krow);
        long long gc = r0 + chunk * 8;
        if (gc + 8 > lim) gc = 0;
        return (unsigned)((krow * ld + gc) * 2);
    }
}

// row-form fragment of the 16 rows starting at local row `r0` of a half-tile image, k-step kk (32 deep): lane l holds
// X[r0 + (l & 15)][32 kk + 8 (l >> 4) + j], j = 0..7 — the A and B operand layout of v_mfma_16x16x32.  (Col-form images
// are read with ds_read_b64_tr_b16 inside the kernel: two 4-row x 16-column blocks per lane group, rows
// 32 kk + 8 (l >> 4) + {0..3} and + {4..7}.)
template <typename T, bool COL>
__device__ __forceinline__ typename M16<T>::vec frag(const char* img, int r0, int kk, int lane) {
    static_assert(!COL, "col-form fragments: col_frag in the kernel");
    typedef typename M16<T>::vec V;
    const int row = r0 + (lane & 15);
    return *reinterpret_cast<const V*>(img + HT<false>::offset(row, kk * 4 + (lane >> 4)));
}

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

