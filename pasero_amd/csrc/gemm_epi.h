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
};
