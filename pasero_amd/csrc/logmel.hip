// K8 — Whisper log-mel features: wav (B, <=480000) fp32 -> (B, 3000, 80) fp32.
// The reference has no mel code of its own: features are produced offline by the third-party
// transformers.WhisperFeatureExtractor (examples/Whisper/extract-features.py:107-117; `transformers` is unpinned in
// setup.py:23).  This kernel restates that published algorithm (pinned to transformers 5.15.0 through
// tests/golden/logmel.npz and the CPU oracle): zero-pad / truncate to 30 s, reflect-pad n_fft/2, periodic Hann window,
// 400-point DFT power spectrum, 201 x 80 slaney mel filterbank, log10(max(., 1e-10)), drop the last frame,
// max(x, max - 8), (x + 4) / 4, laid out (frames, mel) as extract-features.py:116 transposes it.
//
// gfx950 design: one workgroup = 32 frames of one clip.  n_fft = 400 is not a power of two; the DFT is a dense contraction on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32: [32 frames][2] x [2][32 bins]) — but not over 400 samples x 201 bins
// (rounds 1-4: 2800 MFMAs per workgroup, 295 us per 16 clips, the kernel ran at a third of the fp32 matrix peak where
// SURVEY 8d budgets an HBM-bound one).  Round 5: the two symmetries of a real 400-point transform are folded into the INPUT
// before anything is multiplied —
//   x real:            Re X[k] =  sum_{n=0..200} s[n] cos(2 pi k n / 400),  s[n] = x[n] + x[400 - n]   (s[0] = x[0], s[200] = x[200])
//                      Im X[k] = -sum_{n=1..199} d[n] sin(2 pi k n / 400),  d[n] = x[n] - x[400 - n]
//   parity of the bin: cos(2 pi k (200 - n) / 400) = (-1)^k cos(2 pi k n / 400),  sin(...) = -(-1)^k sin(...), so
//     even k:  Re = sum_{n=0..100} ue[n] cos,  Im = -sum ve[n] sin,   ue[n] = s[n] + s[200 - n],  ve[n] = d[n] - d[200 - n]
//     odd  k:  Re = sum_{n=0..100} uo[n] cos,  Im = -sum vo[n] sin,   uo[n] = s[n] - s[200 - n],  vo[n] = d[n] + d[200 - n]
//     (n = 100 pairs with itself: u[100] = s[100], v[100] = d[100]; v[0] = 0)
// — four vectors of 101 (+ 1 zero) values per frame, built straight from the (reflect-indexed, windowed) wav with every sample
// read exactly once, and two bin classes (101 even, 100 odd bins) of four 32-bin tiles each with a contraction of 102: 816
// MFMAs per workgroup instead of 2800.  The twiddle matrix is never materialised: cos(2 pi j / 400), j < 400, is staged in LDS
// once per workgroup ("LDS twiddle staging"; the sine is the same table a quarter period earlier) and each lane walks its bin's
// phase index (n * bin mod 400) incrementally.  The power spectrum goes back to LDS over the dead input tile, and the mel
// projection exploits the filterbank's sparsity (each triangular filter touches a short run of bins).  Per-clip max / clamp /
// scale is a second, elementwise kernel.  HBM-bound by design: 1.92 MB in + 0.96 MB out per 30 s clip.
#include <math.h>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NMEL = 80, NSAMP = 480000, NFRAME = 3000;
constexpr int FT = 32;                  // frames per workgroup
constexpr int NFOLD = 102;              // folded input length: n = 0..100 and one zero (the contraction runs in steps of 2)
constexpr int APITCH = 4 * NFOLD + 1;   // floats per frame: ue | uo | ve | vo; odd, so column reads (lane = frame) are conflict-free
constexpr int PPITCH = 225;             // power tile pitch (bins 0..200)
constexpr int FBC_MAX = 1024;           // room for the packed non-zero filter weights (slaney, 80 filters over 201 bins: ~500)

struct Consts {  // device-resident, built once per device
    float* tw_cos;   // [400]
    float* tw_sin;   // [400]
    float* hann;     // [400]
    float* fb;       // [201][80]
    int* lo;         // [80] first bin with a non-zero weight
    int* hi;         // [80] last bin (inclusive)
    float* fbc;      // [nfbc] the weights fb[lo[m] .. hi[m]][m], filter after filter
    int* off;        // [80] first weight of filter m in fbc
    int* meta;       // [80][3] lo | number of taps | off, one run: staged in LDS with the weights
    int nfbc;
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
        std::vector<float> fbc;
        std::vector<int> off(NMEL);
        for (int m = 0; m < NMEL; ++m) {
            off[m] = (int)fbc.size();
            for (int b = lo[m]; b <= hi[m]; ++b) fbc.push_back(fb[(size_t)b * NMEL + m]);
        }
        if (fbc.size() > (size_t)FBC_MAX) { pk_set_error("pk_logmel: %zu packed filter weights exceed the kernel's room", fbc.size()); return -1; }
        if (fbc.empty()) fbc.push_back(0.f);
        Consts c;
        c.nfbc = (int)fbc.size();
        auto up = [&](void** d, const void* h, size_t n) {
            return hipMalloc(d, n) == hipSuccess && hipMemcpy(*d, h, n, hipMemcpyHostToDevice) == hipSuccess;
        };
        std::vector<int> meta(3 * NMEL);
        for (int m = 0; m < NMEL; ++m) { meta[3 * m] = lo[m]; meta[3 * m + 1] = hi[m] - lo[m] + 1; meta[3 * m + 2] = off[m]; }
        bool ok = up((void**)&c.meta, meta.data(), meta.size() * 4) && up((void**)&c.tw_cos, tc.data(), NFFT * 4) && up((void**)&c.tw_sin, ts.data(), NFFT * 4) &&
                  up((void**)&c.hann, hw.data(), NFFT * 4) && up((void**)&c.fb, fb.data(), fb.size() * 4) &&
                  up((void**)&c.lo, lo.data(), NMEL * 4) && up((void**)&c.hi, hi.data(), NMEL * 4) &&
                  up((void**)&c.fbc, fbc.data(), fbc.size() * 4) && up((void**)&c.off, off.data(), NMEL * 4);
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
    // 52.7 KiB: THREE workgroups per CU (the power spectrum takes the input tile's place once the DFT has read it, and the sine
    // is the cosine table read a quarter period earlier)
    __shared__ __attribute__((aligned(16))) float smem[NFFT + FT * APITCH];
    static_assert(FT * PPITCH <= FT * APITCH, "the power tile aliases the input tile");
    static_assert((NFFT + FT * APITCH) * 4 + 64 <= 160 * 1024 / 3, "three workgroups per CU");
    float* tw_c = smem;                                     // [400] cos(2 pi j / 400); sin(2 pi j / 400) = tw_c[(j + 300) % 400]
    float* a_t = tw_c + NFFT;                               // [32][4 x 102 + 1] folded, windowed frames: ue | uo | ve | vo
    float* p_t = a_t;                                       // [32][225] power spectrum (after the DFT)
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = blockIdx.y, f0 = blockIdx.x * FT;
    const float* x = wav + (long long)clip * wav_stride;
    const long long nvalid = wav_len ? min(wav_len[clip], (long long)NSAMP) : NSAMP;

    for (int j = tid; j < NFFT; j += 256) tw_c[j] = cst.tw_cos[j];
    // The four samples that meet in position n of the folded vectors: x[n], x[400 - n], x[200 - n], x[200 + n] — every sample of
    // the frame exactly once over n = 0..100.  Branch-free (clamped address + a 0 / 1 weight: behind a branch every load is a block
    // of its own and the compiler waits for all of them at the join), all four requested before the first use; the periodic Hann
    // window is symmetric — w[400 - n] = w[n], w[200 - n] = w[200 + n] = 1 - w[n] — so one table read serves the four.
    // n = 0: s[0] = x[0], s[200] = x[200], d = 0;  n = 100 pairs with itself;  n = 101 is the zero that rounds the contraction up.
    for (int i = tid; i < FT * NFOLD; i += 256) {
        const int f = i / NFOLD, n = i - f * NFOLD;
        const bool fv = f0 + f < NFRAME;
        const long long base = (long long)(f0 + f) * HOP - NFFT / 2;       // index of sample 0 in the 30 s zero-padded signal
        auto fetch = [&](int k, bool on) -> float {
            long long j = base + k;
            j = j < 0 ? -j : j;                                             // reflect padding (no edge repeat)
            j = j >= NSAMP ? 2LL * (NSAMP - 1) - j : j;
            const bool ok = on && fv && j < nvalid;
            const float v = x[ok ? j : 0];
            return ok ? v : 0.f;
        };
#ifdef LM_ABL_NOSTAGE
        const float xa = 0.25f * n, xb = 0.5f, xc = 0.125f * f, xe = 1.f;
#else
        const float xa = fetch(n, n <= 100), xb = fetch(NFFT - n, n >= 1 && n <= 100);
        const float xc = fetch(200 - n, n <= 99), xe = fetch(200 + n, n >= 1 && n <= 99);
#endif
        const float w = n <= 100 ? cst.hann[n] : 0.f, wc = 1.f - w;
        const float sn = (xa + xb) * w, dn = (xa - xb) * w, sm = (xc + xe) * wc, dm = (xc - xe) * wc;
        const float vm = n >= 1 ? 1.f : 0.f;
        float* row = a_t + f * APITCH;
        row[n] = sn + sm; row[NFOLD + n] = sn - sm; row[2 * NFOLD + n] = (dn - dm) * vm; row[3 * NFOLD + n] = (dn + dm) * vm;
    }
    __syncthreads();

    // DFT: wave w owns bin tile w of the even class (bins 2 (32 w + c)) and of the odd class (bins 2 (32 w + c) + 1), c = lane & 31;
    // Re and Im accumulators per class
    f32x16 re[2], im[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) re[t][r] = im[t][r] = 0.f;
    const int frame_l = lane & 31, kh = lane >> 5;
    int bin[2], idx[2], step[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bin[t] = 2 * (wave * 32 + frame_l) + t;           // this lane's output column (bin) in class t (may exceed 200: not stored)
        const int b = bin[t] % NFFT;
        idx[t] = (kh * b) % NFFT;                          // phase index of sample n = kh
        step[t] = (2 * b) % NFFT;                          // n advances by 2 per k-step
    }
    const float* arow = a_t + frame_l * APITCH + kh;
#ifdef LM_ABL_NODFT
    for (int s = 0; s < 1; ++s) {
#else
    for (int s = 0; s < NFOLD / 2; ++s) {
#endif
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float u = arow[t * NFOLD + 2 * s], v = arow[(2 + t) * NFOLD + 2 * s];
            const int is = idx[t] + 300 >= NFFT ? idx[t] - 100 : idx[t] + 300;
            const float c = tw_c[idx[t]], sn = tw_c[is];
            re[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(u, c, re[t], 0, 0, 0);
            im[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, sn, im[t], 0, 0, 0);
            idx[t] += step[t];
            if (idx[t] >= NFFT) idx[t] -= NFFT;
        }
    }
    __syncthreads();  // every wave has read its last input values: the tile may be overwritten
    // power spectrum -> LDS  (accumulator: column = lane & 31 = bin of the class, rows = frames); the non-zero filter weights,
    // packed filter by filter, into the part of the dead input tile behind it
    float* m_t = a_t + FT * PPITCH;          // [80][33] log-mel values of the workgroup (transposed: both accesses conflict-free)
    float* fb_t = m_t + NMEL * 33;           // [<= FBC_MAX] packed weights
    int* meta_t = reinterpret_cast<int*>(fb_t + FBC_MAX);  // [80][3] first bin | taps | first weight of every filter
    static_assert(FT * PPITCH + NMEL * 33 + FBC_MAX + 3 * NMEL <= FT * APITCH, "mel staging fits the dead input tile");
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (bin[t] >= NBIN) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = (r & 3) + 8 * (r >> 2) + 4 * kh;
            p_t[fr * PPITCH + bin[t]] = re[t][r] * re[t][r] + im[t][r] * im[t][r];
        }
    }
    for (int j = tid; j < cst.nfbc; j += 256) fb_t[j] = cst.fbc[j];
    if (tid < 3 * NMEL) meta_t[tid] = cst.meta[tid];
    __syncthreads();

    // sparse mel projection + log10.  A (frame, mel) pair per thread with the FRAME on the lane: the 64 lanes of a wave work on two
    // adjacent filters, so they all run the same number of steps (one filter per thread across lanes made every wave wait for its
    // widest filter: 3 taps at the bottom of the scale, 25 at the top), the weight is a broadcast LDS read and the power tile is read
    // down a column (pitch 225: conflict-free).
    float mx = -INFINITY;
    const int f = tid & 31;
    for (int m = tid >> 5; m < NMEL; m += 8) {
        // (from LDS: read from global memory here, the three were a dependent L2 round trip in front of every filter)
        const int lo = meta_t[3 * m], len = meta_t[3 * m + 1];
        const float* pw = p_t + f * PPITCH + lo;
        const float* fw = fb_t + meta_t[3 * m + 2];
        float acc = 0.f;
#ifdef LM_ABL_NOMEL
        acc = pw[0] * fw[0];
#else
        for (int j = 0; j < len; ++j) acc += pw[j] * fw[j];
#endif
        const float lg = log10f(fmaxf(acc, 1e-10f));
        m_t[m * 33 + f] = lg;
        if (f0 + f < NFRAME) mx = fmaxf(mx, lg);
    }
    mx = wave_max(mx);
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    // (frames, mel) rows of this workgroup are one contiguous run of the output: coalesced stores
    float* dst = out + ((long long)clip * NFRAME + f0) * NMEL;
    for (int o = tid; o < FT * NMEL; o += 256) {
        const int fo = o / NMEL, mo = o - fo * NMEL;
        if (f0 + fo < NFRAME) dst[o] = m_t[mo * 33 + fo];
    }
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
