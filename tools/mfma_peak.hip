// Sustained v_mfma_f32_32x32x16_bf16 rate of the card with nothing else in the way (no memory traffic): the practical
// ceiling the GEMM kernels are measured against in DESIGN.md, next to the 2.5 PFLOP/s datasheet peak.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC> void run(int waves_per_simd, float* d) {
    int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD)
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("waves/SIMD %d, %d independent accumulators: %.1f ms  %.0f TFLOP/s\n", waves_per_simd, NACC, ms, flops / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    run<4>(1, d);
    run<4>(2, d);
    run<8>(2, d);
    run<4>(1, d);
    return 0;
}
