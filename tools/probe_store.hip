// probe: how fast can one CU (512 threads) drain 128 KiB of stores?  per-workgroup wall time by s_memrealtime
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int KIND, int SHAPE>
__global__ __launch_bounds__(512) void probe(unsigned* out, long long ld_bytes, unsigned long long* t, int reps) {
    const int tid = threadIdx.x;
    u32x4 v = {1u + tid, 2u, 3u, 4u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        // SHAPE 1/2: WG b owns tile (b / 8, b % 8) of an [M x 2048] bf16 matrix (row stride ld_bytes = 4096)
        const size_t vb = (size_t)blockIdx.x + (size_t)rep * gridDim.x;
        char* base = SHAPE == 0 ? (char*)out + vb * 131072 : (char*)out + (vb / 8) * 256 * ld_bytes + (vb % 8) * 512;
        // SHAPE 0: contiguous 128 KiB per WG.  SHAPE 1: 256 rows x 512 B, row stride ld_bytes (a 256x256 bf16 tile)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int chunk = it * 512 + tid;  // 8192 chunks of 16 B
            char* p;
            if (SHAPE == 0) p = base + (size_t)chunk * 16;
            else if (SHAPE == 1) p = base + (size_t)(chunk >> 5) * ld_bytes + (chunk & 31) * 16;  // 2 rows x 512 B per wave-instr
            else if (SHAPE == 3) {  // register epilogue with a wave owning 64 CONTIGUOUS columns and the two row halves of a
                // 16-row MFMA tile exchanging one chunk (DPP row_ror:8): instruction (mh, i, h2): 8 rows x 128 B (full lines)
                const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, mh = it >> 3, i = (it >> 1) & 3, h2 = it & 1;
                const int r = lane & 15, g = lane >> 4;
                const int row = 128 * mh + 64 * wr + 16 * i + 8 * h2 + (r & 7), col = 64 * wc + 32 * (r >> 3) + 8 * g;
                p = base + (size_t)row * ld_bytes + col * 2;
            }
            else {  // the register epilogue: wave (wr, wc), instruction (mh, i, nh): 16 rows x 64 B
                const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, mh = it >> 3, i = (it >> 1) & 3, nh = it & 1;
                const int row = 128 * mh + 64 * wr + 16 * i + (lane & 15), col = 128 * nh + 32 * wc + 8 * (lane >> 4);
                p = base + (size_t)row * ld_bytes + col * 2;
            }
            if (KIND == 0) *(u32x4*)p = v;
            else if (KIND == 1) __builtin_nontemporal_store(v, (u32x4*)p);
            else if (KIND == 2) { ((uint2*)p)[0] = uint2{v[0], v[1]}; ((uint2*)p)[1] = uint2{v[2], v[3]}; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) t[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}
template <int KIND, int SHAPE> void run(const char* name, unsigned* buf, unsigned long long* t, int nwg, long long ld_bytes, int reps) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<KIND, SHAPE>), dim3(nwg), dim3(512), 0, 0, buf, ld_bytes, t, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), t, nwg * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = h[nwg / 2] / 100.0;
    printf("%-34s nwg=%4d reps=%d: median %.2f us per WG (%.2f us / 128 KiB) = %.1f GB/s per CU, chip %.2f TB/s\n", name, nwg, reps, med, med / reps,
           131072.0 * reps / med / 1e3, 131072.0 * reps * nwg / med / 1e6);
}
int main() {
    unsigned* buf; unsigned long long* t;
    hipMalloc(&buf, (size_t)1 << 31); hipMalloc(&t, 4096 * 8);
    const long long ld = 2048 * 2;  // fc1 output row: 2048 bf16
    for (int nwg : {256, 128, 64, 32, 1}) {
        run<0, 0>("plain x4, contiguous", buf, t, nwg, ld, 4);
        run<1, 0>("nt x4, contiguous", buf, t, nwg, ld, 4);
        run<2, 0>("plain x2 x2, contiguous", buf, t, nwg, ld, 4);
        run<0, 1>("plain x4, tile rows 512 B", buf, t, nwg, ld, 4);
        run<1, 1>("nt x4, tile rows 512 B", buf, t, nwg, ld, 4);
        run<0, 2>("plain x4, 16 rows x 64 B / instr", buf, t, nwg, ld, 4);
        run<1, 2>("nt x4, 16 rows x 64 B / instr", buf, t, nwg, ld, 4);
        run<0, 3>("plain x4, 8 rows x 128 B / instr", buf, t, nwg, ld, 4);
        run<1, 3>("nt x4, 8 rows x 128 B / instr", buf, t, nwg, ld, 4);
    }
    return 0;
}
