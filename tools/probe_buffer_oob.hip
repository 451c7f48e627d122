// probe: range check of buffer_load ... lds with an SGPR offset, and what an out-of-range lane leaves in LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* g, unsigned* out, int nrec, unsigned soff) {
    __shared__ __attribute__((aligned(16))) unsigned smem[1024];
    typedef __attribute__((address_space(3))) void lds_void;
    for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = 0xDEADBEEF;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)smem, 16, threadIdx.x * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
    unsigned *g, *o;
    hipMalloc(&g, 1 << 20); hipMalloc(&o, 4096);
    std::vector<unsigned> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    hipMemcpy(g, h.data(), 1 << 20, hipMemcpyHostToDevice);
    struct { int nrec; unsigned soff; const char* what; } cases[] = {
        {512, 0, "nrec=512 soff=0: lanes 32.. out of range"},
        {1024, 512, "nrec=1024 soff=512: lanes 32.. beyond nrec only if soff counts"},
        {0, 0, "nrec=0: all out of range"}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, o, c.nrec, c.soff);
        std::vector<unsigned> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("%s\n  lane0: %08x lane31: %08x lane32: %08x lane63: %08x  (expected in-range lane L: %08x + L*4 + soff/4)\n", c.what,
               r[0], r[31 * 4], r[32 * 4], r[63 * 4], 0x1000000u);
    }
    return 0;
}
