// The DPP / permlane forms of the wave reductions (pasero_amd/csrc/common.h: wave_sum, wave_max, lanes8_sum,
// half_wave_max, half_wave_swap) against their `__shfl_xor` statements, bit for bit, on 4096 random waves.
//   hipcc -O3 --offload-arch=gfx950 tools/wave_reduce_check.hip -o /tmp/wave_reduce_check && /tmp/wave_reduce_check
// (run by tests/test_kernels_gpu.py::test_wave_reductions_match_their_shuffle_statements)
#include <cstdio>
#include <cstring>
#include <vector>
#include "../pasero_amd/csrc/common.h"

extern "C" void pk_set_error(const char*, ...) {}

__device__ __forceinline__ float shfl_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// out[0..4][i]: the new forms, out[5..9][i]: the shuffle statements
__global__ void check_kernel(const float* x, float* out, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float v = x[i];
    out[0 * n + i] = wave_sum(v);
    out[5 * n + i] = wave_sum_shfl(v);
    out[1 * n + i] = wave_max(v);
    out[6 * n + i] = shfl_max(v);
    out[2 * n + i] = lanes8_sum(v);
    float p = v;
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    p += __shfl_xor(p, 4);
    out[7 * n + i] = p;
    out[3 * n + i] = half_wave_max(v);
    out[8 * n + i] = fmaxf(v, __shfl_xor(v, 32, 64));
    // the attention kernels' exchange: the upper half-wave's `a` pieces change places with the lower half-wave's `b` pieces
    unsigned a = __float_as_uint(v), b = ~a;
    const bool hi = threadIdx.x >= 32;
    const unsigned got = __shfl_xor(hi ? a : b, 32, 64);
    const unsigned ra = hi ? got : a, rb = hi ? b : got;
    half_wave_swap(a, b);
    out[4 * n + i] = __uint_as_float(a ^ (b * 3u));
    out[9 * n + i] = __uint_as_float(ra ^ (rb * 3u));
}

int main() {
    const int waves = 4096, n = waves * 64;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((int)(s >> 8) - (1 << 23)) * 1e-3f * ((i % 7) + 1);
    }
    float *x, *out;
    if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&out, 10 * n * 4) != hipSuccess) return 2;
    if (hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(check_kernel, dim3(waves), dim3(64), 0, 0, x, out, n);
    std::vector<float> r(10 * n);
    if (hipMemcpy(r.data(), out, 10 * n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char* names[5] = {"wave_sum", "wave_max", "lanes8_sum", "half_wave_max", "half_wave_swap"};
    int bad_total = 0;
    for (int k = 0; k < 5; ++k) {
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += std::memcmp(&r[k * n + i], &r[(5 + k) * n + i], 4) != 0;
        std::printf("%s: %d mismatches of %d\n", names[k], bad, n);
        bad_total += bad;
    }
    return bad_total != 0;
}
