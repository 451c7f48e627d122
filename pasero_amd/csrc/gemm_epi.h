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
    // the ReLU mask as one bit per element (pk_gemm_relu_bits): [M][ldbits] bytes, bit (n & 7) of byte n >> 3 of row m = (stored
    // C[m][n] > 0).  mode 0 + ReLU writes it beside C, mode 2 + ReLU reads it instead of aux.  gemmbs.hip takes it as a launch
    // argument of its own; gemm8p.hip (round 5: the d = 1024 feed-forward, K = 1024) through these fields
    unsigned char* bits = nullptr;
    long long ldbits = 0;
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
