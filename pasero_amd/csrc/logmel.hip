// K8 — Whisper log-mel features: wav (B, <=480000) fp32 -> (B, 3000, 80) fp32.
// The reference has no mel code of its own: features are produced offline by the third-party
// transformers.WhisperFeatureExtractor (examples/Whisper/extract-features.py:107-117; `transformers` is unpinned in
// setup.py:23).  This kernel restates that published algorithm (pinned to transformers 5.15.0 through
// tests/golden/logmel.npz and the CPU oracle): zero-pad / truncate to 30 s, reflect-pad n_fft/2, periodic Hann window,
// 400-point DFT power spectrum, 201 x 80 slaney mel filterbank, log10(max(., 1e-10)), drop the last frame,
// max(x, max - 8), (x + 4) / 4, laid out (frames, mel) as extract-features.py:116 transposes it.
//
// gfx950 design: one workgroup = 32 frames of one clip.  n_fft = 400 is not a power of two, so the DFT is a dense
// contraction on the exact-fp32 matrix cores: Re/Im[32 frames][32 bins] += frame[32][2] x twiddle[2][32] with
// v_mfma_f32_32x32x2_f32, 200 k-steps.  The twiddle matrix is never materialised: cos/sin(2*pi*j/400), j < 400, and
// the Hann window are staged in LDS once per workgroup ("LDS twiddle staging") and each lane walks its bin's phase
// index (n * bin mod 400) incrementally.  The windowed frames are built in LDS straight from the (reflect-indexed) wav,
// the power spectrum goes back to LDS, and the mel projection exploits the filterbank's sparsity (each triangular filter
// touches a short run of bins).  Per-clip max / clamp / scale is a second, elementwise kernel.
// HBM-bound by design: 1.92 MB in + 0.96 MB out per 30 s clip.
#include <math.h>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NMEL = 80, NSAMP = 480000, NFRAME = 3000;
constexpr int FT = 32;                  // frames per workgroup
constexpr int NBT = 7;                  // bin tiles of 32 (201 -> 224)
constexpr int APITCH = NFFT + 1;        // floats; column reads of the frame tile (lane = frame) are conflict-free
constexpr int PPITCH = NBT * 32 + 1;    // power tile pitch

struct Consts {  // device-resident, built once per device
    float* tw_cos;   // [400]
    float* tw_sin;   // [400]
    float* hann;     // [400]
    float* fb;       // [201][80]
    int* lo;         // [80] first bin with a non-zero weight
    int* hi;         // [80] last bin (inclusive)
};

double hz_to_mel(double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; }
double mel_to_hz(double m) { return m >= 15.0 ? 1000.0 * exp(log(6.4) / 27.0 * (m - 15.0)) : 200.0 * m / 3.0; }

std::mutex g_mu;
std::vector<Consts> g_consts(64);
std::vector<char> g_ready(64, 0);

int get_consts(Consts& out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) { pk_set_error("pk_logmel: hipGetDevice failed"); return -1; }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready[dev]) {
        std::vector<float> tc(NFFT), ts(NFFT), hw(NFFT), fb((size_t)NBIN * NMEL);
        std::vector<int> lo(NMEL), hi(NMEL);
        for (int j = 0; j < NFFT; ++j) {
            tc[j] = (float)cos(2.0 * M_PI * j / NFFT);
            ts[j] = (float)sin(2.0 * M_PI * j / NFFT);
            hw[j] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * j / NFFT));  // periodic Hann
        }
        // slaney-scale, slaney(area)-normalised triangular filters, 0..8000 Hz, 201 linear FFT bins
        std::vector<double> pts(NMEL + 2);
        const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
        for (int i = 0; i < NMEL + 2; ++i) pts[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (NMEL + 1));
        for (int m = 0; m < NMEL; ++m) {
            lo[m] = NBIN; hi[m] = -1;
            const double enorm = 2.0 / (pts[m + 2] - pts[m]);
            for (int b = 0; b < NBIN; ++b) {
                const double f = 8000.0 * b / (NBIN - 1);
                const double down = (f - pts[m]) / (pts[m + 1] - pts[m]);
                const double up = (pts[m + 2] - f) / (pts[m + 2] - pts[m + 1]);
                double w = fmax(0.0, fmin(down, up)) * enorm;
                fb[(size_t)b * NMEL + m] = (float)w;
                if (w > 0.0) { if (b < lo[m]) lo[m] = b; if (b > hi[m]) hi[m] = b; }
            }
            if (hi[m] < 0) { lo[m] = 0; hi[m] = -1; }
        }
        Consts c;
        auto up = [&](void** d, const void* h, size_t n) {
            return hipMalloc(d, n) == hipSuccess && hipMemcpy(*d, h, n, hipMemcpyHostToDevice) == hipSuccess;
        };
        bool ok = up((void**)&c.tw_cos, tc.data(), NFFT * 4) && up((void**)&c.tw_sin, ts.data(), NFFT * 4) &&
                  up((void**)&c.hann, hw.data(), NFFT * 4) && up((void**)&c.fb, fb.data(), fb.size() * 4) &&
                  up((void**)&c.lo, lo.data(), NMEL * 4) && up((void**)&c.hi, hi.data(), NMEL * 4);
        if (!ok) { pk_set_error("pk_logmel: constant upload failed"); return -1; }
        g_consts[dev] = c;
        g_ready[dev] = 1;
    }
    out = g_consts[dev];
    return 0;
}

__global__ __launch_bounds__(256, 3) void logmel_kernel(const float* __restrict__ wav, const long long* __restrict__ wav_len,
                                                     long long wav_stride, float* __restrict__ out,
                                                     float* __restrict__ blockmax, Consts cst) {
    // 52.9 KiB: THREE workgroups per CU (the power spectrum takes the frames' place once the DFT has read them, and the sine
    // is the cosine table read a quarter period earlier — as 81 KiB, one workgroup per CU with nothing beside it: 782 us
    // per call against an MFMA time of ~160; two per CU: 505)
    __shared__ __attribute__((aligned(16))) float smem[NFFT + FT * APITCH];
    static_assert(FT * PPITCH <= FT * APITCH, "the power tile aliases the frame tile");
    float* tw_c = smem;                                     // [400] cos(2 pi j / 400); sin(2 pi j / 400) = tw_c[(j + 300) % 400]
    float* a_t = tw_c + NFFT;                               // [32][401] windowed frames
    float* p_t = a_t;                                       // [32][225] power spectrum (after the DFT)
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = blockIdx.y, f0 = blockIdx.x * FT;
    const float* x = wav + (long long)clip * wav_stride;
    const long long nvalid = wav_len ? min(wav_len[clip], (long long)NSAMP) : NSAMP;

    for (int j = tid; j < NFFT; j += 256) tw_c[j] = cst.tw_cos[j];
    for (int i = tid; i < FT * NFFT; i += 256) {
        const int f = i / NFFT, n = i % NFFT;
        long long j = (long long)(f0 + f) * HOP + n - NFFT / 2;       // index into the 30 s zero-padded signal
        if (j < 0) j = -j;                                              // reflect padding (no edge repeat)
        if (j >= NSAMP) j = 2LL * (NSAMP - 1) - j;
        const float v = (f0 + f < NFRAME && j < nvalid) ? x[j] : 0.f;
        a_t[f * APITCH + n] = v * cst.hann[n];
    }
    __syncthreads();

    // DFT: wave w owns bin tiles w and w + 4 (tile 7 does not exist); Re and Im accumulators per tile
    f32x16 re[2], im[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) re[t][r] = im[t][r] = 0.f;
    const int frame_l = lane & 31, kh = lane >> 5;
    int bin[2], idx[2], step[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bin[t] = (wave + 4 * t) * 32 + frame_l;           // this lane's output column (bin) for tile t
        const int b = bin[t] % NFFT;
        idx[t] = (kh * b) % NFFT;                          // phase index of sample n = kh
        step[t] = (2 * b) % NFFT;                          // n advances by 2 per k-step
    }
    const bool two = wave + 4 < NBT;
    for (int s = 0; s < NFFT / 2; ++s) {
        const float av = a_t[frame_l * APITCH + 2 * s + kh];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two) break;
            const int is = idx[t] + 300 >= NFFT ? idx[t] - 100 : idx[t] + 300;
            const float c = tw_c[idx[t]], sn = tw_c[is];
            re[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c, re[t], 0, 0, 0);
            im[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sn, im[t], 0, 0, 0);
            idx[t] += step[t];
            if (idx[t] >= NFFT) idx[t] -= NFFT;
        }
    }
    __syncthreads();  // every wave has read its last frame samples: the tile may be overwritten
    // power spectrum -> LDS  (accumulator: column = lane & 31 = bin, rows = frames)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = (r & 3) + 8 * (r >> 2) + 4 * kh;
            p_t[fr * PPITCH + (wave + 4 * t) * 32 + frame_l] = re[t][r] * re[t][r] + im[t][r] * im[t][r];
        }
    }
    __syncthreads();

    // sparse mel projection + log10; (frame, mel) pairs over the workgroup
    float mx = -INFINITY;
    for (int i = tid; i < FT * NMEL; i += 256) {
        const int f = i / NMEL, m = i % NMEL;
        if (f0 + f >= NFRAME) continue;
        float acc = 0.f;
        const int lo = cst.lo[m], hi = cst.hi[m];
        for (int b = lo; b <= hi; ++b) acc += p_t[f * PPITCH + b] * cst.fb[b * NMEL + m];
        const float lg = log10f(fmaxf(acc, 1e-10f));
        out[((long long)clip * NFRAME + f0 + f) * NMEL + m] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = wave_max(mx);
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    if (tid == 0) blockmax[clip * gridDim.x + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

// x = (max(x, clipmax - 8) + 4) / 4
__global__ __launch_bounds__(256) void logmel_finalize_kernel(float* __restrict__ out, const float* __restrict__ blockmax,
                                                              int nblocks) {
    __shared__ float smax;
    const int clip = blockIdx.y;
    if (threadIdx.x < 64) {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < nblocks; i += 64) m = fmaxf(m, blockmax[clip * nblocks + i]);
        m = wave_max(m);
        if (threadIdx.x == 0) smax = m;
    }
    __syncthreads();
    const float floor_v = smax - 8.f;
    float4* p = reinterpret_cast<float4*>(out + (long long)clip * NFRAME * NMEL);
    const int n4 = NFRAME * NMEL / 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        float4 v = p[i];
        v.x = (fmaxf(v.x, floor_v) + 4.f) * 0.25f;
        v.y = (fmaxf(v.y, floor_v) + 4.f) * 0.25f;
        v.z = (fmaxf(v.z, floor_v) + 4.f) * 0.25f;
        v.w = (fmaxf(v.w, floor_v) + 4.f) * 0.25f;
        p[i] = v;
    }
}

}  // namespace

extern "C" size_t pk_logmel_workspace(int B) { return (size_t)B * ((NFRAME + FT - 1) / FT) * sizeof(float); }

extern "C" int pk_logmel(const float* wav, const long long* wav_len, long long wav_stride, float* out,
                         void* workspace, size_t ws_bytes, int B, void* stream) {
    PK_CHECK_ARG(wav && out, "pk_logmel: null tensor");
    PK_CHECK_ARG(B >= 0 && B <= 65535, "pk_logmel: bad batch size %d", B);
    PK_CHECK_ARG(workspace && ws_bytes >= pk_logmel_workspace(B), "pk_logmel: workspace too small");
    PK_CHECK_ARG(wav_len || wav_stride >= NSAMP, "pk_logmel: clips shorter than 30 s need wav_len");
    if (B == 0) return 0;
    Consts cst;
    if (int rc = get_consts(cst)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nfb = (NFRAME + FT - 1) / FT;
    hipLaunchKernelGGL(logmel_kernel, dim3(nfb, B), dim3(256), 0, s, wav, wav_len, wav_stride, out,
                       (float*)workspace, cst);
    PK_LAUNCH_CHECK();
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3(32, B), dim3(256), 0, s, out, (const float*)workspace, nfb);
    PK_LAUNCH_CHECK();
    return 0;
}
