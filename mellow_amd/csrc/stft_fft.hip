// A1 as what it algorithmically is: a 1024-point FFT per frame (SURVEY.md 8d prices the STFT at 0.051 GFLOP per clip; the DFT
// GEMM spends 2.10).  Reference: torchlibrosa Spectrogram(power=2) at htsat.py:647-649, called :864 -- two conv1d with the
// windowed DFT basis, re^2 + im^2.  The engine takes this path only in f32x3 mode and only when the checkpoint's conv weights
// ARE window[n] * cos / sin(2 pi k n / 1024) (verified element by element at load time, engine_weights.cpp); otherwise the GEMM runs.
//
// One wave per frame, n = 64 a + b, k = k1 + 16 e + 256 f:
//   A  lane b:           16-point DFT over a of window * x                       -> Y[b][k1],  times W_1024^(b k1)
//      (LDS transpose, per wave, conflict-free with a row stride of 68 floats)
//   B  lane (k1, d):     16-point DFT over j of Z[k1][d + 4 j]                   -> V[d][e],   times W_64^(d e)
//   C  quad (d = 0..3):  4-point DFT across the quad's lanes (DPP quad_perm)     -> X[k1 + 16 e + 256 f], f = lane & 3
//   power = re^2 + im^2 (separate roundings, like real ** 2 + imag ** 2), staged through LDS for 256-byte stores.
#include "common.h"
#include "kernels.h"

namespace mellow {

struct cf { float r, i; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.i, -a.r}; }      // a * (-i)

// 4-point forward DFT: X[q] = sum_p x[p] (-i)^(p q)
__device__ __forceinline__ void dft4(cf& a, cf& b, cf& c, cf& d) {
    const cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_mi(csub(b, d));
    a = cadd(t0, t2); b = cadd(t1, t3); c = csub(t0, t2); d = csub(t1, t3);
}
// 16-point forward DFT, natural order in and out: n = n1 + 4 n2, k = k2 + 4 k1
__device__ __forceinline__ void dft16(cf (&x)[16]) {
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft4(x[n1], x[n1 + 4], x[n1 + 8], x[n1 + 12]);     // over n2: x[n1 + 4 k2] = A[n1][k2]
    // twiddles W_16^(n1 k2), W_16 = exp(-2 pi i / 16)
    const cf w1 = {C1, -S1}, w2 = {R, -R}, w3 = {S1, -C1}, w6 = {-R, -R}, w9 = {-C1, S1};
    x[1 + 4] = cmul(x[1 + 4], w1); x[1 + 8] = cmul(x[1 + 8], w2); x[1 + 12] = cmul(x[1 + 12], w3);
    x[2 + 4] = cmul(x[2 + 4], w2); x[2 + 8] = mul_mi(x[2 + 8]);   x[2 + 12] = cmul(x[2 + 12], w6);
    x[3 + 4] = cmul(x[3 + 4], w3); x[3 + 8] = cmul(x[3 + 8], w6); x[3 + 12] = cmul(x[3 + 12], w9);
    cf y[16];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {                                                   // over n1: X[k2 + 4 k1]
        cf a = x[0 + 4 * k2], b = x[1 + 4 * k2], c = x[2 + 4 * k2], d = x[3 + 4 * k2];
        dft4(a, b, c, d);
        y[k2] = a; y[k2 + 4] = b; y[k2 + 8] = c; y[k2 + 12] = d;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = y[k];
}

constexpr int FFT_LS = 68;          // LDS row stride (floats): bank = (4 k1 + d + 4 j) mod 64 is distinct for the 64 lanes of step B

__global__ __launch_bounds__(256) void stft_fft_power_kernel(const float* __restrict__ wpad, int fpc, int64_t clip_stride, int hop,
                                                             int M, const float* __restrict__ win, const float2* __restrict__ tw1,
                                                             const float2* __restrict__ tw2, float* __restrict__ power) {
    __shared__ float zr[4][16 * FFT_LS], zi[4][16 * FFT_LS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;                                    // wave-uniform; no workgroup barrier below
    const float* x = wpad + (int64_t)(m / fpc) * clip_stride + (int64_t)(m % fpc) * hop;
    float* sr = zr[wave];
    float* si = zi[wave];
    // ---- A ----
    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = {x[64 * a + lane] * win[64 * a + lane], 0.f};
    dft16(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const float2 t = tw1[k1 * 64 + lane];
        const cf z = cmul(v[k1], cf{t.x, t.y});
        sr[k1 * FFT_LS + lane] = z.r;
        si[k1 * FFT_LS + lane] = z.i;
    }
    __builtin_amdgcn_wave_barrier();                       // one wave owns the region: LDS operations of a wave complete in order
    // ---- B ----
    const int k1 = lane >> 2, d = lane & 3;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = {sr[k1 * FFT_LS + d + 4 * j], si[k1 * FFT_LS + d + 4 * j]};
    dft16(v);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float2 t = tw2[d * 16 + e];
        v[e] = cmul(v[e], cf{t.x, t.y});
    }
    __builtin_amdgcn_wave_barrier();
    // ---- C: X[f] = sum_d V_d (-i)^(d f) across the quad; this lane keeps f = d.  Only k <= 512 is stored: f = 0, 1 and bin 512
    float* pw = sr;                                        // the staging row reuses the wave's region (all reads above are done)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const cf v0 = {dpp_mov<0x00>(v[e].r), dpp_mov<0x00>(v[e].i)}, v1 = {dpp_mov<0x55>(v[e].r), dpp_mov<0x55>(v[e].i)};
        const cf v2 = {dpp_mov<0xAA>(v[e].r), dpp_mov<0xAA>(v[e].i)}, v3 = {dpp_mov<0xFF>(v[e].r), dpp_mov<0xFF>(v[e].i)};
        const cf t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = mul_mi(csub(v1, v3));
        cf X;
        if (d == 0) X = cadd(t0, t2);
        else if (d == 1) X = cadd(t1, t3);
        else if (d == 2) X = csub(t0, t2);
        else X = csub(t1, t3);
        const int k = k1 + 16 * e + 256 * d;
        if (k <= 512) pw[k] = __fadd_rn(__fmul_rn(X.r, X.r), __fmul_rn(X.i, X.i));
    }
    __builtin_amdgcn_wave_barrier();
    float* out = power + (int64_t)m * 544;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int k = lane + 64 * t;
        if (k < 544) out[k] = k <= 512 ? pw[k] : 0.f;
    }
}

void launch_stft_fft_power(const float* wpad, int fpc, int64_t clip_stride, int hop, int M, const float* win, const float* tw1,
                           const float* tw2, float* power, hipStream_t s) {
    hipLaunchKernelGGL(stft_fft_power_kernel, dim3((M + 3) / 4), dim3(256), 0, s, wpad, fpc, clip_stride, hop, M, win,
                       reinterpret_cast<const float2*>(tw1), reinterpret_cast<const float2*>(tw2), power);
}

}  // namespace mellow
