// Non-GEMM kernels of the audio encoder.  All are HBM-bound
// index-math kernels: permutations of the reference (reflect pad, bicubic resample + 4x256 fold,
// cyclic shift + window partition, patch-merging gather, un-fold for the token-semantic head) are
// address arithmetic on loads, never materialised copies.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace mellow {

// ---- A1 reflect pad (torchlibrosa STFT center=True, pad_mode='reflect'; htsat.py:640-641) -----------
__global__ void reflect_pad_kernel(const float* __restrict__ wav, int64_t n, float* __restrict__ out, int64_t plen,
                                   int pad) {
    const int c = blockIdx.y;
    const float* src = wav + (int64_t)c * n;
    float* dst = out + (int64_t)c * plen;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plen; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = i - pad;
        if (j < 0) j = -j;
        if (j >= n) j = 2 * (n - 1) - j;
        if (j < 0) j = 0;           // only reachable for the tail padding beyond the reflected region
        if (j >= n) j = n - 1;
        dst[i] = src[j];
    }
}
void launch_reflect_pad(const float* wav, int n_clips, int64_t n_samples, float* out, int64_t padded_len, int pad,
                        hipStream_t s) {
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(512, n_clips), dim3(256), 0, s, wav, n_samples, out, padded_len, pad);
}

// ---- A4 + A5: bicubic time resample (align_corners) + fold + 4x4 patch conv + LayerNorm(96) -----------
// reference htsat.py:830-845 (reshape_wav2img) and :86-116 (PatchEmbed).
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// A wave keeps its lanes' conv weights (channels lane and lane + 64: 2 x 16 taps), bias and LayerNorm parameters in registers and
// walks over 16 consecutive tokens, four at a time (64 lanes = 4 tokens x 16 pixels fetch the patch pixels of a group together);
// one token per wave re-read the 6 KB of weights for every token (2 GB through the L1 per call: 138 us for 64 clips).
constexpr int PE_TPW = 16;                        // tokens per wave (n_tokens is a multiple of 4096)
__global__ __launch_bounds__(256) void fold_patch_embed_kernel(const float* __restrict__ lm, int src_frames,
                                                               int n_crops, int crop_hop, int crop_len,
                                                               const float* __restrict__ cw,
                                                               const float* __restrict__ cb,
                                                               const float* __restrict__ lw,
                                                               const float* __restrict__ lb, float* __restrict__ x0,
                                                               int64_t n_tokens) {
    __shared__ __attribute__((aligned(16))) float px[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tok0 = ((int64_t)blockIdx.x * 4 + wave) * PE_TPW;
    if (tok0 >= n_tokens) return;                 // (whole waves only; the waves of a block share no LDS region and no barrier)
    const bool has1 = lane < 32;
    float w0[16], w1[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        w0[p] = cw[lane * 16 + p];
        w1[p] = has1 ? cw[(lane + 64) * 16 + p] : 0.f;
    }
    const float b0 = cb[lane], b1 = has1 ? cb[lane + 64] : 0.f;
    const float lw0 = lw[lane], lb0 = lb[lane], lw1 = has1 ? lw[lane + 64] : 0.f, lb1 = has1 ? lb[lane + 64] : 0.f;
    for (int g4 = 0; g4 < PE_TPW; g4 += 4) {
        {   // lane = (token of the group, pixel): dy = pixel / 4 (row of the folded image), dx = pixel % 4 (time)
            const int64_t tok = tok0 + g4 + (lane >> 4);
            const int pix = lane & 15;
            const int v = (int)(tok >> 12);           // virtual clip
            const int t = (int)(tok & 4095);
            const int th = t >> 6, tw = t & 63;       // token row (freq-folded), token column (time)
            const int src = v / n_crops, crop = v % n_crops;
            const float* base = lm + ((int64_t)src * src_frames + (int64_t)crop * crop_hop) * 64;
            const int dy = pix >> 2, dx = pix & 3;
            const int row = th * 4 + dy;              // 0..255 = chunk*64 + mel
            const int chunk = row >> 6, mel = row & 63;
            const int tt = chunk * 256 + tw * 4 + dx; // destination frame 0..1023
            float val;
            if (crop_len == 1024) {
                val = base[(int64_t)tt * 64 + mel];
            } else {
                // upsample_bicubic2d, align_corners=True: scale = (in-1)/(out-1) in fp32, A = -0.75
                const float scale = (float)(crop_len - 1) / (float)(1024 - 1);
                const float real = scale * (float)tt;
                int i0 = (int)floorf(real);
                i0 = i0 < crop_len - 1 ? i0 : crop_len - 1;
                float lam = real - (float)i0;
                lam = fminf(fmaxf(lam, 0.f), 1.f);
                const float A = -0.75f;
                const float c0 = cubic2(lam + 1.f, A), c1 = cubic1(lam, A);
                const float c2 = cubic1(1.f - lam, A), c3 = cubic2(2.f - lam, A);
                int i[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int ii = i0 - 1 + k;
                    ii = ii < 0 ? 0 : (ii > crop_len - 1 ? crop_len - 1 : ii);
                    i[k] = ii;
                }
                val = __fmul_rn(base[(int64_t)i[0] * 64 + mel], c0);
                val = __fadd_rn(val, __fmul_rn(base[(int64_t)i[1] * 64 + mel], c1));
                val = __fadd_rn(val, __fmul_rn(base[(int64_t)i[2] * 64 + mel], c2));
                val = __fadd_rn(val, __fmul_rn(base[(int64_t)i[3] * 64 + mel], c3));
            }
            __builtin_amdgcn_wave_barrier();          // (the previous group's reads of this wave's region are issued)
            px[wave][lane] = val;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float pv[16];
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const float4 t4 = *reinterpret_cast<const float4*>(&px[wave][q * 16 + p4 * 4]);
                pv[p4 * 4] = t4.x; pv[p4 * 4 + 1] = t4.y; pv[p4 * 4 + 2] = t4.z; pv[p4 * 4 + 3] = t4.w;
            }
            float a0 = b0, a1 = b1;
#pragma unroll
            for (int p = 0; p < 16; ++p) { a0 += pv[p] * w0[p]; a1 += pv[p] * w1[p]; }
            const float v0 = a0, v1 = has1 ? a1 : 0.f;
            const float mean = wave_sum(v0 + v1) * (1.0f / 96.0f);
            const float d0 = v0 - mean, d1 = has1 ? v1 - mean : 0.f;
            const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / 96.0f);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            float* dst = x0 + (tok0 + g4 + q) * 96;
            dst[lane] = d0 * rstd * lw0 + lb0;
            if (has1) dst[lane + 64] = d1 * rstd * lw1 + lb1;
        }
    }
}
void launch_fold_patch_embed(const float* logmel_bn, int n_src, int src_frames, int n_crops, int crop_hop,
                             int crop_len, const float* conv_w, const float* conv_b, const float* ln_w,
                             const float* ln_b, float* x0, hipStream_t s) {
    const int64_t n_tokens = (int64_t)n_src * n_crops * 4096;
    hipLaunchKernelGGL(fold_patch_embed_kernel, dim3((unsigned)((n_tokens + 4 * PE_TPW - 1) / (4 * PE_TPW))), dim3(256), 0, s, logmel_bn,
                       src_frames, n_crops, crop_hop, crop_len, conv_w, conv_b, ln_w, ln_b, x0, n_tokens);
}

// ---- LayerNorm family: one wave per row, row kept in registers (C <= 1536) ------------------------------
constexpr int LN_MAXV = 6;  // float4 per lane

template <int MODE>  // 0: plain / row-map gather, 1: patch-merging gather
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int64_t M, int C, const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        const int32_t* __restrict__ row_map, int ntok, int R,
                                                        int Cin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + wave;
    if (m >= M) return;
    const int nv = C >> 2;
    const float* src_row = nullptr;
    int64_t seg_base[4];
    if (MODE == 0) {
        int64_t src = m;
        if (row_map) src = (m / ntok) * ntok + row_map[m % ntok];
        src_row = in + src * C;
    } else {
        // output token (clip, i, j) of the (R/2)^2 grid gathers tokens (2i,2j), (2i+1,2j), (2i,2j+1), (2i+1,2j+1)
        // in that order (reference htsat.py:488-492)
        const int R2 = R >> 1;
        const int64_t clip = m / (R2 * R2);
        const int ij = (int)(m % (R2 * R2));
        const int i = ij / R2, j = ij % R2;
        const int64_t cb = clip * R * R;
        seg_base[0] = (cb + (int64_t)(2 * i) * R + 2 * j) * Cin;
        seg_base[1] = (cb + (int64_t)(2 * i + 1) * R + 2 * j) * Cin;
        seg_base[2] = (cb + (int64_t)(2 * i) * R + 2 * j + 1) * Cin;
        seg_base[3] = (cb + (int64_t)(2 * i + 1) * R + 2 * j + 1) * Cin;
    }
    float4 x[LN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXV; ++q) {
        const int v = lane + 64 * q;
        x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < nv) {
            if (MODE == 0) {
                x[q] = reinterpret_cast<const float4*>(src_row)[v];
            } else {
                const int e = v * 4;
                const int seg = e / Cin, c = e % Cin;
                x[q] = *reinterpret_cast<const float4*>(in + seg_base[seg] + c);
            }
            sum += (x[q].x + x[q].y) + (x[q].z + x[q].w);
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXV; ++q) {
        const int v = lane + 64 * q;
        if (v < nv) {
            const float a = x[q].x - mean, bb = x[q].y - mean, c = x[q].z - mean, d = x[q].w - mean;
            sq += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + 1e-5f);
    float4* dst = reinterpret_cast<float4*>(out + m * C);
#pragma unroll
    for (int q = 0; q < LN_MAXV; ++q) {
        const int v = lane + 64 * q;
        if (v < nv) {
            const float4 ww = reinterpret_cast<const float4*>(w)[v];
            const float4 bv = reinterpret_cast<const float4*>(b)[v];
            float4 y;
            y.x = (x[q].x - mean) * rstd * ww.x + bv.x;
            y.y = (x[q].y - mean) * rstd * ww.y + bv.y;
            y.z = (x[q].z - mean) * rstd * ww.z + bv.z;
            y.w = (x[q].w - mean) * rstd * ww.w + bv.w;
            dst[v] = y;
        }
    }
}
void launch_layernorm(const float* in, float* out, int M, int C, const float* w, const float* b,
                      const int32_t* row_map, int ntok, hipStream_t s) {
    hipLaunchKernelGGL((layernorm_kernel<0>), dim3((M + 3) / 4), dim3(256), 0, s, in, out, (int64_t)M, C, w, b,
                       row_map, ntok, 0, 0);
}
void launch_merge_layernorm(const float* in, float* out, int n, int R, int C, const float* w, const float* b,
                            hipStream_t s) {
    const int64_t M = (int64_t)n * (R / 2) * (R / 2);
    hipLaunchKernelGGL((layernorm_kernel<1>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, in, out, M, 4 * C, w, b,
                       (const int32_t*)nullptr, 0, R, C);
}

// The row-map LayerNorm with its output pre-split in APB order (f32x3 mode: the consumer is an x3q GEMM, common.h).  Same
// statistics as layernorm_kernel<0> bit for bit (same lane -> column mapping, same reduction order, same expression for y);
// a workgroup owns 32 consecutive OUTPUT rows: pass 1 gathers them (one row per wave at a time, coalesced), keeps them in LDS
// and forms mean / rstd; pass 2 re-reads them from LDS in the (row % 32, k-half) order of an APB slot run, so that every store
// instruction of a wave writes 1 KiB contiguous (the scheme of rmsnorm_apb_kernel below).  NQ = float4 per lane (C <= 256 NQ).
// AMX = true (fp8 mode): the same rows, quantised to MXFP8 in the AMX image order of gemm_mx8_kernel (common.h: one scale per 32
// columns; a lane of pass 2 holds 16 consecutive columns, its partner lane the other 16 of the block; pad blocks up to a whole
// k64 step are written as zeros).
template <int NQ, bool AMX>
__global__ __launch_bounds__(256) void layernorm_apb_kernel(const float* __restrict__ in, i32x4* __restrict__ out, int64_t M, int C,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const int32_t* __restrict__ row_map, int ntok, uint8_t* __restrict__ sc) {
    extern __shared__ __attribute__((aligned(16))) float ln_rows_s[];
    __shared__ float mu_s[32], rs_s[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2, KT = C >> 4, RS = C + 4;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    float4 x[8][NQ];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t m = m0 + wave * 8 + i;
        m = m < M ? m : M - 1;
        int64_t src = m;
        if (row_map) src = (m / ntok) * ntok + row_map[m % ntok];
        const float4* sp = reinterpret_cast<const float4*>(in + src * C);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = lane + 64 * q;
            x[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < nv) x[i][q] = sp[v];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float4* dst = reinterpret_cast<float4*>(ln_rows_s + (wave * 8 + i) * RS);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = lane + 64 * q;
            if (v < nv) {
                dst[v] = x[i][q];
                sum += (x[i][q].x + x[i][q].y) + (x[i][q].z + x[i][q].w);
            }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = lane + 64 * q;
            if (v < nv) {
                const float a = x[i][q].x - mean, bb = x[i][q].y - mean, c = x[i][q].z - mean, d = x[i][q].w - mean;
                sq += (a * a + bb * bb) + (c * c + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + 1e-5f);
        if (lane == 0) { mu_s[wave * 8 + i] = mean; rs_s[wave * 8 + i] = rstd; }
    }
    __syncthreads();
    const int ml = lane & 31;
    const int64_t m = m0 + ml;
    if (m >= M) return;
    const float mean = mu_s[ml], rstd = rs_s[ml];
    const int kh = lane >> 5;
    if constexpr (AMX) {
        const int KT64 = (C + 63) >> 6, KQ = (KT64 + 3) >> 2;
        for (int kb = wave; kb < 2 * KT64; kb += 4) {
            const int col = kb * 32 + kh * 16;
            float y[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col + 4 * q < C) {
                    const float4 x0 = *reinterpret_cast<const float4*>(ln_rows_s + ml * RS + col + 4 * q);
                    const float4 w0 = *reinterpret_cast<const float4*>(w + col + 4 * q), b0 = *reinterpret_cast<const float4*>(b + col + 4 * q);
                    v.x = (x0.x - mean) * rstd * w0.x + b0.x;
                    v.y = (x0.y - mean) * rstd * w0.y + b0.y;
                    v.z = (x0.z - mean) * rstd * w0.z + b0.z;
                    v.w = (x0.w - mean) * rstd * w0.w + b0.w;
                }
                y[4 * q] = v.x; y[4 * q + 1] = v.y; y[4 * q + 2] = v.z; y[4 * q + 3] = v.w;
            }
            amx_store16(out, sc, m, kb, KT64, KQ, y, kh);
        }
        return;
    }
    for (int kt = wave; kt < KT; kt += 4) {
        const int col = kt * 16 + kh * 8;
        const float4 x0 = *reinterpret_cast<const float4*>(ln_rows_s + ml * RS + col), x1 = *reinterpret_cast<const float4*>(ln_rows_s + ml * RS + col + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(w + col), w1 = *reinterpret_cast<const float4*>(w + col + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + col), b1 = *reinterpret_cast<const float4*>(b + col + 4);
        float y[8];
        y[0] = (x0.x - mean) * rstd * w0.x + b0.x;
        y[1] = (x0.y - mean) * rstd * w0.y + b0.y;
        y[2] = (x0.z - mean) * rstd * w0.z + b0.z;
        y[3] = (x0.w - mean) * rstd * w0.w + b0.w;
        y[4] = (x1.x - mean) * rstd * w1.x + b1.x;
        y[5] = (x1.y - mean) * rstd * w1.y + b1.y;
        y[6] = (x1.z - mean) * rstd * w1.z + b1.z;
        y[7] = (x1.w - mean) * rstd * w1.w + b1.w;
        apb_store8(out, m, kt * 2 + kh, KT, y);
    }
}
// C % 16 == 0, C <= 768; out_apb holds roundup(M, 128) rows (rows >= M are left as they are: their products are never stored)
void launch_layernorm_apb(const float* in, void* out_apb, int M, int C, const float* w, const float* b, const int32_t* row_map,
                          int ntok, hipStream_t s, void* out_scales) {
    const size_t lds = (size_t)32 * (C + 4) * sizeof(float);          // 98.8 KB at C = 768
    const dim3 grid((M + 31) / 32), block(256);
    i32x4* o = reinterpret_cast<i32x4*>(out_apb);
    uint8_t* sc = reinterpret_cast<uint8_t*>(out_scales);
#define MELLOW_LN_LAUNCH(NQ, AMX)                                                                                        \
    do {                                                                                                                 \
        set_max_dynamic_lds(reinterpret_cast<const void*>(&layernorm_apb_kernel<NQ, AMX>), 100 * 1024);                  \
        hipLaunchKernelGGL((layernorm_apb_kernel<NQ, AMX>), grid, block, lds, s, in, o, (int64_t)M, C, w, b, row_map, ntok, sc); \
    } while (0)
    if (out_scales) {        // AMX image + scale bytes (fp8 mode)
        if (C <= 256) MELLOW_LN_LAUNCH(1, true); else if (C <= 512) MELLOW_LN_LAUNCH(2, true); else MELLOW_LN_LAUNCH(3, true);
    } else {
        if (C <= 256) MELLOW_LN_LAUNCH(1, false); else if (C <= 512) MELLOW_LN_LAUNCH(2, false); else MELLOW_LN_LAUNCH(3, false);
    }
#undef MELLOW_LN_LAUNCH
}

// LlamaRMSNorm (fp32): w * (x * rsqrt(mean(x^2) + eps))
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t M,
                                                      int C, const float* __restrict__ w, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + wave;
    if (m >= M) return;
    const int nv = C >> 2;
    const float4* src = reinterpret_cast<const float4*>(in + m * C);
    float4 x[LN_MAXV];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXV; ++q) {
        const int v = lane + 64 * q;
        x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < nv) {
            x[q] = src[v];
            ss += (x[q].x * x[q].x + x[q].y * x[q].y) + (x[q].z * x[q].z + x[q].w * x[q].w);
        }
    }
    const float r = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
    float4* dst = reinterpret_cast<float4*>(out + m * C);
#pragma unroll
    for (int q = 0; q < LN_MAXV; ++q) {
        const int v = lane + 64 * q;
        if (v < nv) {
            const float4 ww = reinterpret_cast<const float4*>(w)[v];
            float4 y;
            y.x = __fmul_rn(ww.x, __fmul_rn(x[q].x, r));
            y.y = __fmul_rn(ww.y, __fmul_rn(x[q].y, r));
            y.z = __fmul_rn(ww.z, __fmul_rn(x[q].z, r));
            y.w = __fmul_rn(ww.w, __fmul_rn(x[q].w, r));
            dst[v] = y;
        }
    }
}
// the same arithmetic, output pre-split in APB order.  A workgroup owns 32 consecutive rows.  Pass 1 is the plain kernel's
// statistic (same lane -> column mapping and reduction order: the same 1/rms bit for bit), one row per wave at a time; pass 2
// re-reads the rows (L2-hot) with lane = (row % 32, k-half), the order of an APB slot run, so every store instruction of a
// wave writes 1 KiB contiguous (a first version that stored from the row-per-wave mapping scattered 16-byte pieces 12 KiB
// apart and took 2.3x the time of the plain kernel)
template <bool AMX>
__global__ __launch_bounds__(256) void rmsnorm_apb_kernel(const float* __restrict__ in, i32x4* __restrict__ out, int64_t M,
                                                          int C, const float* __restrict__ w, float eps, uint8_t* __restrict__ sc) {
    // The 32 rows of the workgroup are staged in LDS by pass 1 (coalesced row reads) and re-read from there by pass 2 in the
    // (row % 32, k-half) order of an APB slot run: a second pass over global memory in that order touches 32 cache lines per load
    // instruction (22 us per launch at M = 12448; from LDS: see profiles/).  Row stride C + 4 floats: the sixteen lanes of a
    // ds_read_b128 group then sit on sixteen different bank quads.
    extern __shared__ __attribute__((aligned(16))) float rows_s[];
    __shared__ float rs[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2, KT = C >> 4, RS = C + 4;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    // all eight rows of the wave are loaded before the first reduction (rows past M: a clamped re-read, result unused)
    float ss[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t m = m0 + wave * 8 + i;
        m = m < M ? m : M - 1;
        const float4* src = reinterpret_cast<const float4*>(in + m * C);
        float4* dst = reinterpret_cast<float4*>(rows_s + (wave * 8 + i) * RS);
        ss[i] = 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAXV; ++q) {
            const int v = lane + 64 * q;
            if (v < nv) {
                const float4 x = src[v];
                dst[v] = x;
                ss[i] += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float r = 1.0f / sqrtf(wave_sum(ss[i]) / (float)C + eps);
        if (lane == 0) rs[wave * 8 + i] = r;
    }
    __syncthreads();
    const int ml = lane & 31;
    const int64_t m = m0 + ml;
    if (m >= M) return;
    const float r = rs[ml];
    const int kh = lane >> 5;
    if constexpr (AMX) {          // fp8 mode: the same values as MXFP8 in AMX order (see layernorm_apb_kernel)
        const int KT64 = (C + 63) >> 6, KQ = (KT64 + 3) >> 2;
        for (int kb = wave; kb < 2 * KT64; kb += 4) {
            const int col = kb * 32 + kh * 16;
            float y[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col + 4 * q < C) {
                    const float4 x0 = *reinterpret_cast<const float4*>(rows_s + ml * RS + col + 4 * q);
                    const float4 w0 = *reinterpret_cast<const float4*>(w + col + 4 * q);
                    v = make_float4(__fmul_rn(w0.x, __fmul_rn(x0.x, r)), __fmul_rn(w0.y, __fmul_rn(x0.y, r)), __fmul_rn(w0.z, __fmul_rn(x0.z, r)),
                                    __fmul_rn(w0.w, __fmul_rn(x0.w, r)));
                }
                y[4 * q] = v.x; y[4 * q + 1] = v.y; y[4 * q + 2] = v.z; y[4 * q + 3] = v.w;
            }
            amx_store16(out, sc, m, kb, KT64, KQ, y, kh);
        }
        return;
    }
    for (int kt = wave; kt < KT; kt += 4) {
        const int col = kt * 16 + kh * 8;
        const float4 x0 = *reinterpret_cast<const float4*>(rows_s + ml * RS + col), x1 = *reinterpret_cast<const float4*>(rows_s + ml * RS + col + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(w + col), w1 = *reinterpret_cast<const float4*>(w + col + 4);
        const float y[8] = {__fmul_rn(w0.x, __fmul_rn(x0.x, r)), __fmul_rn(w0.y, __fmul_rn(x0.y, r)), __fmul_rn(w0.z, __fmul_rn(x0.z, r)),
                            __fmul_rn(w0.w, __fmul_rn(x0.w, r)), __fmul_rn(w1.x, __fmul_rn(x1.x, r)), __fmul_rn(w1.y, __fmul_rn(x1.y, r)),
                            __fmul_rn(w1.z, __fmul_rn(x1.z, r)), __fmul_rn(w1.w, __fmul_rn(x1.w, r))};
        apb_store8(out, m, kt * 2 + kh, KT, y);
    }
}
void launch_rmsnorm_apb(const float* in, void* out_apb, int M, int C, const float* w, float eps, hipStream_t s, void* out_scales) {
    const size_t lds = (size_t)32 * (C + 4) * sizeof(float);          // 74 KB at C = 576: two workgroups per CU
    set_max_dynamic_lds(reinterpret_cast<const void*>(&rmsnorm_apb_kernel<false>), 96 * 1024);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&rmsnorm_apb_kernel<true>), 96 * 1024);
    if (out_scales) hipLaunchKernelGGL(rmsnorm_apb_kernel<true>, dim3((M + 31) / 32), dim3(256), lds, s, in, reinterpret_cast<i32x4*>(out_apb), (int64_t)M, C, w, eps, reinterpret_cast<uint8_t*>(out_scales));
    else hipLaunchKernelGGL(rmsnorm_apb_kernel<false>, dim3((M + 31) / 32), dim3(256), lds, s, in, reinterpret_cast<i32x4*>(out_apb), (int64_t)M, C, w, eps, (uint8_t*)nullptr);
}
void launch_rmsnorm(const float* in, float* out, int M, int C, const float* w, float eps, hipStream_t s) {
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, in, out, (int64_t)M, C, w, eps);
}

// ---- A7 window attention core on the matrix pipe (v_mfma_f32_32x32x2_f32, exact fp32); reference htsat.py:301-327.  Inputs arrive in
// window order (the LN kernel did roll + partition), so a window is 64 consecutive rows. ----------------------------------------
// One wave per (window, head).  Scores are computed TRANSPOSED, S^T[key][query] = sum_d K[key][d] Q[query][d] (K tile = MFMA A
// operand, Q = B operand), so a lane owns ONE query column and 16 keys of each 32-key tile: the softmax over the 64 keys of a
// query is in-lane plus one half-wave exchange, and the P^T accumulator registers are directly the B operand of
// O^T[d][query] += V^T[d][key] P^T[key][query] (k-step r pairs the keys the two half-waves hold in register r) -- the scheme of
// prefill_attn.hip, without the causal mask, with the relative-position bias and the shift mask added to S.
//   * head_dim 24: k-step s of the QK^T product uses d = s + 12 h (h = lane / 32), so a lane's operand values are 12 contiguous
//     floats of its q / k row (3 float4 loads each, straight from global memory: no LDS for Q or K);
//   * V^T operand: V[key][d = lane % 32] from a row-major LDS copy of the wave's 64 x 24 V tile (lanes 24..31 multiply zeros:
//     the 32-row output tile is 25 % padding);  48 + 64 = 112 MFMAs per tile instead of 3072 FMAs + 768 LDS broadcasts per lane.
//   * out_apb (f32x3 mode, the proj GEMM runs on the x3q kernel): the output goes out pre-split in APB order instead of fp32 --
//     a lane's quads are the epilogue quads of the GEMMs (common.h: apb_store_quads), head_dim 24 = three 8-column groups.
//   * IN16 (fp8 mode, round 6): q / k / v arrive as bf16 rows (stored rounded once by the qkv GEMM's epilogue: half the bytes of the
//     kernel's dominant stream) and are widened on load; the arithmetic below is unchanged.
__device__ __forceinline__ void wa_ld4(const float* base, int64_t elem, bool in16, float (&o)[4]) {
    if (in16) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ r = *reinterpret_cast<const u32x2_*>(reinterpret_cast<const uint16_t*>(base) + elem);
        o[0] = __uint_as_float(r[0] << 16); o[1] = __uint_as_float(r[0] & 0xffff0000u); o[2] = __uint_as_float(r[1] << 16); o[3] = __uint_as_float(r[1] & 0xffff0000u);
    } else {
        const float4 r = *reinterpret_cast<const float4*>(base + elem);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
}
template <bool IN16>
__global__ __launch_bounds__(256) void window_attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                    int C, int nH, const float* __restrict__ bias_exp,
                                                                    const float* __restrict__ mask, int nW,
                                                                    int64_t n_tiles, i32x4* __restrict__ out_apb) {
    __shared__ __attribute__((aligned(16))) float Vs[4][64 * 24];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;   // tile = window * nH + head
    const bool active = tile < n_tiles;
    const int64_t win = active ? tile / nH : 0;
    const int hd = active ? (int)(tile % nH) : 0;
    const int h = lane >> 5, ml = lane & 31;
    const float scale = 0.20412414523193150818f;            // 24^-0.5 (python float -> fp32 scalar multiply)
    // operands of S^T: token rows ml and ml + 32, d = 12 h .. 12 h + 11
    float kf[2][12], qf[2][12];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t row = (win * 64 + t * 32 + ml) * (3 * C) + hd * 24 + 12 * h;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            float q4[4], k4[4];
            wa_ld4(qkv, row + v * 4, IN16, q4);
            wa_ld4(qkv, row + C + v * 4, IN16, k4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qf[t][4 * v + j] = q4[j] * scale; kf[t][4 * v + j] = k4[j]; }
        }
    }
    {   // V tile of this wave -> LDS, row-major [key][24]
        const int64_t vrow = (win * 64 + lane) * (3 * C) + 2 * C + hd * 24;
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            float v4[4];
            wa_ld4(qkv, vrow + v * 4, IN16, v4);
            *reinterpret_cast<float4*>(&Vs[wave][lane * 24 + v * 4]) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
    }
    // S^T tiles [key tile kt][query tile qt]
    f32x16 S[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][qt][r] = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < 12; ++s_) S[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kt][s_], qf[qt][s_], S[kt][qt], 0, 0, 0);
        }
    // lane: query qt*32 + ml, keys kt*32 + (r&3) + 8(r>>2) + 4h.  bias / mask rows are [query][key]: 4 consecutive keys per float4
    float inv[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = qt * 32 + ml;
        const float* brow = bias_exp + ((int64_t)hd * 64 + qi) * 64;
        const float* mrow = mask ? mask + ((win % nW) * 64 + qi) * 64 : nullptr;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int k0 = kt * 32 + 8 * g4 + 4 * h;
                const float4 b4 = *reinterpret_cast<const float4*>(brow + k0);
                float add[4] = {b4.x, b4.y, b4.z, b4.w};
                if (mrow) {
                    const float4 m4 = *reinterpret_cast<const float4*>(mrow + k0);
                    // (s + bias) + mask: the reference's order (htsat.py:317, 321)
                    S[kt][qt][4 * g4 + 0] = (S[kt][qt][4 * g4 + 0] + add[0]) + m4.x;
                    S[kt][qt][4 * g4 + 1] = (S[kt][qt][4 * g4 + 1] + add[1]) + m4.y;
                    S[kt][qt][4 * g4 + 2] = (S[kt][qt][4 * g4 + 2] + add[2]) + m4.z;
                    S[kt][qt][4 * g4 + 3] = (S[kt][qt][4 * g4 + 3] + add[3]) + m4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) S[kt][qt][4 * g4 + j] += add[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, S[kt][qt][4 * g4 + j]);
            }
        mx = half_max(mx);                                   // the other 32 keys of this query live in lane ^ 32
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f((S[kt][qt][r] - mx) * 1.44269504088896340736f);
                S[kt][qt][r] = pv;
                sum += pv;
            }
        inv[qt] = 1.0f / half_sum(sum);
    }
    __syncthreads();                                         // V tiles staged (each wave reads only its own)
    // O^T[d][query] += V^T[d][key] P^T[key][query]
    f32x16 O[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[qt][r] = 0.f;
    const float* Vc = Vs[wave];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float a = ml < 24 ? Vc[key * 24 + ml] : 0.f;
            O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S[kt][0][r], O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S[kt][1][r], O[1], 0, 0, 0);
        }
    if (active && out_apb) {
        const int KT = C >> 4;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int64_t m = win * 64 + qt * 32 + ml;
            float X[4], Y[4], Z[4], w[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { X[j] = O[qt][j] * inv[qt]; Y[j] = O[qt][4 + j] * inv[qt]; Z[j] = O[qt][8 + j] * inv[qt]; }
            apb_store_quads(out_apb, m, hd * 3, KT, X, Y, h);           // d = 0..7 (lower half-wave), 8..15 (upper)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(Z[j]), __float_as_uint(Z[j]), false, false);
                w[j] = __uint_as_float(r[0]);
                w[4 + j] = __uint_as_float(r[1]);
            }
            if (h == 0) apb_store8(out_apb, m, hd * 3 + 2, KT, w);      // d = 16..23
        }
    } else if (active) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float* dst = out + (win * 64 + qt * 32 + ml) * C + hd * 24;
#pragma unroll
            for (int g4 = 0; g4 < 3; ++g4) {                 // d = 8 g4 + 4 h + (0..3) < 24
                const int d = 8 * g4 + 4 * h;
                *reinterpret_cast<float4*>(dst + d) = make_float4(O[qt][4 * g4] * inv[qt], O[qt][4 * g4 + 1] * inv[qt],
                                                                  O[qt][4 * g4 + 2] * inv[qt], O[qt][4 * g4 + 3] * inv[qt]);
            }
        }
    }
}
void launch_window_attention(const float* qkv, float* out, int M, int C, int nH, const float* bias_exp,
                             const float* mask, int nW, hipStream_t s, void* out_apb, bool qkv16) {
    const int64_t n_tiles = (int64_t)(M / 64) * nH;
    if (qkv16) {
        hipLaunchKernelGGL(window_attention_mfma_kernel<true>, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, qkv, out, C, nH,
                           bias_exp, mask, nW, n_tiles, reinterpret_cast<i32x4*>(out_apb));
        return;
    }
    hipLaunchKernelGGL(window_attention_mfma_kernel<false>, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, qkv, out, C, nH,
                       bias_exp, mask, nW, n_tiles, reinterpret_cast<i32x4*>(out_apb));
}

// ---- A10 tail: latent mean + im2col for the token-semantic conv (htsat.py:742-775) ---------------------------
// y [n][64][768], token = F*8 + T with F = g*2 + cf (g = time group 0..3, cf = freq bin 0..1).
// regrouped time index tt = g*8 + T (0..31).  a_ts row (clip, tt), column (cf*3 + dt)*768 + ch holds
// y[clip][token(g', cf, T')][ch] with tt' = tt + dt - 1 (zero outside 0..31).
__global__ __launch_bounds__(256) void tail_latent_im2col_kernel(const float* __restrict__ y, float* __restrict__ latent,
                                                                 int64_t latent_stride, float* __restrict__ a_ts) {
    const int clip = blockIdx.x;
    const float* yc = y + (int64_t)clip * 64 * 768;
    // latent: mean over the 64 positions in (cf, tt) order, like avgpool(flatten(x, 2))
    for (int ch = threadIdx.x; ch < 768; ch += 256) {
        float sacc = 0.f;
        for (int cf = 0; cf < 2; ++cf)
            for (int tt = 0; tt < 32; ++tt) {
                const int tok = ((tt >> 3) * 2 + cf) * 8 + (tt & 7);
                sacc += yc[tok * 768 + ch];
            }
        latent[(int64_t)clip * latent_stride + ch] = sacc * (1.0f / 64.0f);
    }
    // im2col
    float4* dst = reinterpret_cast<float4*>(a_ts + (int64_t)clip * 32 * 4608);
    for (int i = threadIdx.x; i < 32 * 6 * 192; i += 256) {
        const int c4 = i % 192;
        const int seg = (i / 192) % 6;
        const int tt = i / (192 * 6);
        const int cf = seg / 3, dt = seg % 3;
        const int ts = tt + dt - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ts >= 0 && ts < 32) {
            const int tok = ((ts >> 3) * 2 + cf) * 8 + (ts & 7);
            v = reinterpret_cast<const float4*>(yc + tok * 768)[c4];
        }
        dst[(int64_t)tt * 1152 + seg * 192 + c4] = v;
    }
}
void launch_tail_latent_im2col(const float* y, int n, float* latent, int64_t latent_stride, float* a_ts,
                               hipStream_t s) {
    hipLaunchKernelGGL(tail_latent_im2col_kernel, dim3(n), dim3(256), 0, s, y, latent, latent_stride, a_ts);
}

// mean over crops, summed sequentially from zero then divided (htsat.py:922-931)
__global__ void crop_average_kernel(const float* __restrict__ in, int n_crops, int64_t len, int64_t in_stride,
                                    float* __restrict__ out, int64_t out_stride) {
    const int c = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int k = 0; k < n_crops; ++k) a = __fadd_rn(a, in[((int64_t)c * n_crops + k) * in_stride + i]);
        out[(int64_t)c * out_stride + i] = a / (float)n_crops;
    }
}
void launch_crop_average(const float* in, int n, int n_crops, int64_t len, int64_t in_stride, float* out,
                         int64_t out_stride, hipStream_t s) {
    hipLaunchKernelGGL(crop_average_kernel, dim3(64, n), dim3(256), 0, s, in, n_crops, len, in_stride, out, out_stride);
}

// ---- host-harness helper A0 on the device: sinc-hann polyphase resampler (reference wrapper.py:146 ->
// torchaudio.transforms.Resample defaults; host twin: mellow_amd/audio.py resample()).  out[c][f*new + p] =
// sum_k x[c][f*orig + k - width] * wT[k][p], k < 2*width + orig, zero outside the clip; one thread per output sample,
// the frame's threads share x (broadcast) and read wT rows coalesced.
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t n_in, const float* __restrict__ wT,
                                                       int orig, int nw, int klen, int width, float* __restrict__ out,
                                                       int64_t n_out) {
    const int c = blockIdx.y;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int64_t f = o / nw;
    const int p = (int)(o % nw);
    const float* xc = x + (int64_t)c * n_in;
    const int64_t base = f * orig - width;
    float acc = 0.f;
    for (int k = 0; k < klen; ++k) {
        const int64_t t = base + k;
        const float xv = (t >= 0 && t < n_in) ? xc[t] : 0.f;
        acc = fmaf(xv, wT[(int64_t)k * nw + p], acc);
    }
    out[(int64_t)c * n_out + o] = acc;
}
void launch_resample(const float* x, int n_clips, int64_t n_in, const float* wT, int orig, int nw, int klen, int width,
                     float* out, int64_t n_out, hipStream_t s) {
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256), n_clips), dim3(256), 0, s, x, n_in, wT, orig, nw,
                       klen, width, out, n_out);
}

__global__ void gelu_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = gelu_erf(in[i]);
}
void launch_gelu(const float* in, float* out, int64_t n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, s, in, out, n);
}

// ---- A13 + A14: downsample + prefix assembly (decoder.py:14-18, 36-55) ----------------------------------------
// pooled row j (0..127) averages framewise rows 8j..8j+7 = eight copies of distinct row j/4; the 8-term
// sequential sum + divide is reproduced so the rounding matches avg_pool2d.
__device__ __forceinline__ float pool8(float a) {
    float s = a;
#pragma unroll
    for (int k = 0; k < 7; ++k) s = __fadd_rn(s, a);
    return s * 0.125f;
}
__device__ __forceinline__ float audio_row_value(const float* __restrict__ p33, int r, int c) {
    // r in 0..128 of the 129-row downsampled audio embedding; p33 -> [33][576] of one clip
    if (r == 0) return p33[c];
    return pool8(p33[(1 + ((r - 1) >> 2)) * 576 + c]);
}
__global__ __launch_bounds__(192) void prefix_assemble_kernel(const float* __restrict__ proj33,
                                                              const float* __restrict__ embed,
                                                              const int32_t* __restrict__ ids, int B, int text_len,
                                                              int sep_id, int vocab, float* __restrict__ prefix,
                                                              unsigned long long* __restrict__ bad_id_word) {
    const int pos = blockIdx.x, b = blockIdx.y;
    const int P = 2 * 129 + 2 + text_len;
    float* dst = prefix + ((int64_t)b * P + pos) * 576;
    // a prompt id outside the vocabulary: the reference's embedding lookup raises IndexError (decoder.py:47).  The row is read
    // clamped (never a wild read) and the call's error word -- mapped host memory, read by the host with the results -- is set:
    // the range check costs no kernel and no synchronisation of its own (it was a torch min/max + two host syncs per call)
    if (pos >= 260 && threadIdx.x == 0 && bad_id_word) {
        const int id = ids[(int64_t)b * text_len + (pos - 260)];
        if (id < 0 || id >= vocab)
            __hip_atomic_store(bad_id_word, (1ull << 63) | ((unsigned long long)(unsigned)b << 32) | (unsigned)id, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int c = threadIdx.x; c < 576; c += 192) {
        float v;
        if (pos < 129) v = audio_row_value(proj33 + (int64_t)b * 33 * 576, pos, c);
        else if (pos == 129) v = embed[(int64_t)sep_id * 576 + c];
        else if (pos < 259) v = audio_row_value(proj33 + (int64_t)(B + b) * 33 * 576, pos - 130, c);
        else if (pos == 259) v = embed[(int64_t)sep_id * 576 + c];
        else v = embed[(int64_t)min(max(ids[(int64_t)b * text_len + (pos - 260)], 0), vocab - 1) * 576 + c];   // clamped: see engine.py
        dst[c] = v;
    }
}
void launch_prefix_assemble(const float* proj33, const float* embed, const int32_t* ids, int B, int text_len,
                            int sep_id, int vocab, float* prefix, unsigned long long* bad_id_word, hipStream_t s) {
    hipLaunchKernelGGL(prefix_assemble_kernel, dim3(260 + text_len, B), dim3(192), 0, s, proj33, embed, ids, B,
                       text_len, sep_id, vocab, prefix, bad_id_word);
}
__global__ __launch_bounds__(192) void downsample33_kernel(const float* __restrict__ proj33, float* __restrict__ out) {
    const int r = blockIdx.x, clip = blockIdx.y;
    for (int c = threadIdx.x; c < 576; c += 192)
        out[((int64_t)clip * 129 + r) * 576 + c] = audio_row_value(proj33 + (int64_t)clip * 33 * 576, r, c);
}
__global__ __launch_bounds__(192) void gather_rows_kernel(const float* __restrict__ table, int width4, const int32_t* __restrict__ ids,
                                                          int n_rows, float* __restrict__ out) {
    const int i = blockIdx.x;
    const int64_t src = min(max(ids[i], 0), n_rows - 1);
    for (int c = threadIdx.x; c < width4; c += 192)
        reinterpret_cast<float4*>(out)[(int64_t)i * width4 + c] = reinterpret_cast<const float4*>(table)[src * width4 + c];
}
void launch_gather_rows(const float* table, int width, const int32_t* ids, int n, int n_rows, float* out, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(192), 0, s, table, width / 4, ids, n_rows, out);
}
__global__ __launch_bounds__(192) void gather_span_kernel(const float* __restrict__ in, int T, int from_pos, int n, float* __restrict__ out) {
    const int j = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x < 144)
        reinterpret_cast<float4*>(out)[((int64_t)b * n + j) * 144 + threadIdx.x] =
            reinterpret_cast<const float4*>(in)[((int64_t)b * T + from_pos + j) * 144 + threadIdx.x];
}
void launch_gather_span(const float* in, int B, int T, int from_pos, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(gather_span_kernel, dim3(n, B), dim3(192), 0, s, in, T, from_pos, n, out);
}
// zero the slots [t0, t1) of every KV page (pages x Tmax x 64 floats): plain coalesced float4 stores
// (f4 = float4 per slot: 16 for fp32 pages, 8 for the bf16 pages of the fp8 mode)
__global__ __launch_bounds__(256) void clear_page_slots_kernel(float* __restrict__ cache, int64_t pages, int Tmax, int t0, int t1, int f4) {
    const int64_t per = (int64_t)(t1 - t0) * f4;                   // float4 per page
    const int64_t total = pages * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pg = i / per, r = i % per;
        reinterpret_cast<float4*>(cache + (pg * Tmax + t0) * (f4 * 4))[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
void launch_clear_page_slots(float* cache, int64_t pages, int Tmax, int t0, int t1, hipStream_t s, bool pages16) {
    if (t1 <= t0 || pages <= 0) return;
    const int f4 = pages16 ? 8 : 16;
    const int64_t total = pages * (int64_t)(t1 - t0) * f4;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(clear_page_slots_kernel, dim3(blocks), dim3(256), 0, s, cache, pages, Tmax, t0, t1, f4);
}
void launch_downsample33(const float* proj33, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(downsample33_kernel, dim3(129, n), dim3(192), 0, s, proj33, out);
}

}  // namespace mellow
