// Epilogue parameters shared by the GEMM kernels (gemm.hip, gemm256.hip)
#pragma once
struct EpiParams {
    const void* bias;   // [N] or null
    const void* aux;    // [M, ldaux] or null
    void* preact;       // optional second output: value before the activation
    long long ldaux, ldc, ldpre;
    int act;            // PK_ACT_*
    int mode;           // 0: act(v+bias)   1: act(v+bias) + aux   2: v * act'(aux)
    float alpha;
    // a vocabulary that is no multiple of 8 in buffers whose rows are padded to one (pk_gemm_ex, include/pasero_hip.h):
    long long nstore = 0;   // PK_GEMM_PAD_N: columns up to `nstore` = N rounded up to 8 may be STORED (0: N) — gemm8p's lean epilogue
    int half_m = 0;         // gemm8p.hip: 128 x 256 tiles (an output of 256 x 256 tiles would fill half the chip)
    long long kb_rows = 0;  // PK_GEMM_PAD_K: the rows a col-form B really has (the K passed on is rounded up to 8; 0: K)
};

// One weight-gradient problem of a grouped launch (include/pasero_hip.h: PkWgradProblem): C[M,N] = A^T B, both operands
// in col form (A = dY [K][lda], B = X [K][ldb]), asum_out[m] = sum_k A(m,k) (optional bias gradient)
#define PK_WGRAD_MAX 8
struct PkWgradProblem {
    const void* A;
    const void* B;
    void* C;
    void* asum_out;
    long long M, N, K, lda, ldb, ldc;
};
