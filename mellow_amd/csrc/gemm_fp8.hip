// fp8 (OCP e4m3) GEMM for BASELINE config 5: activations quantised per row, weights per output channel, products on
// v_mfma_f32_32x32x16_fp8_fp8 (2.46 PFLOP/s measured, tools/microbench/mfma_peak.hip) with fp32 accumulation, and the
// same fused epilogues as the fp32 kernel (gemm_epilogue.h).  Not bit-exact by construction: tests report token
// agreement against the fp32 path.  The reference has no counterpart (it runs fp32 ATen matmuls).
//
//   quant_rows   A fp32 [M][K] -> A8 e4m3 [M][K64] + a_scale[m] = amax_m / 448        (one wave per row)
//   pack_fp8     W fp32 P-layout -> P8-layout + w_scale[n] = amax_n / 448             (once per tensor at load time)
//   gemm_fp8     128 x 128 tile, 4 waves (64 x 64 each), 64 k-bytes per stage, double-buffered LDS in fragment order:
//                stage = A [2 k32][4 m-tiles][64 lanes][16 B] + W [2 k32][4 n-tiles][64][16 B] = 16 KiB; one
//                ds_read_b128 feeds two MFMAs; issued as D[n][m] like the fp32 kernel, so the epilogues are shared.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace mellow {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// ---- per-row quantisation ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_rows_kernel(const float* __restrict__ A, int64_t lda, int M, int K,
                                                         uint8_t* __restrict__ A8, int64_t lda8, float* __restrict__ a_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    const f32x4* row = reinterpret_cast<const f32x4*>(A + (int64_t)m * lda);
    const int K4 = K >> 2;
    float amax = 0.f;
    for (int i = lane; i < K4; i += 64) {
        const f32x4 v = row[i];
        amax = fmaxf(fmaxf(amax, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.0f / amax : 0.f;
    if (lane == 0) a_scale[m] = amax > 0.f ? amax / 448.0f : 1.0f;
    uint32_t* out = reinterpret_cast<uint32_t*>(A8 + (int64_t)m * lda8);
    const int K4p = (int)(lda8 >> 2);
    for (int i = lane; i < K4p; i += 64) {
        int w = 0;
        if (i < K4) {
            const f32x4 v = row[i];       // second pass over the row: L2-resident (<= 18 KB)
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * inv, v.y * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * inv, v.w * inv, w, true);
        }
        out[i] = (uint32_t)w;             // zero padding up to a multiple of 64 bytes
    }
}
void launch_quant_rows(const float* A, int64_t lda, int M, int K, uint8_t* A8, int64_t lda8, float* a_scale, hipStream_t s) {
    hipLaunchKernelGGL(quant_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, A, lda, M, K, A8, lda8, a_scale);
}

// ---- weight packing ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float p_layout_at(const float* __restrict__ Wp, int K8, int n, int k) {
    return Wp[(((int64_t)(n >> 5) * K8 + (k >> 3)) * 64 + (n & 31) + 32 * ((k >> 2) & 1)) * 4 + (k & 3)];
}
__global__ __launch_bounds__(256) void fp8_row_scale_kernel(const float* __restrict__ Wp, int NP, int KP, float* __restrict__ w_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= NP) return;
    float amax = 0.f;
    for (int k = lane; k < KP; k += 64) amax = fmaxf(amax, fabsf(p_layout_at(Wp, KP >> 3, n, k)));
    amax = wave_max(amax);
    if (lane == 0) w_scale[n] = amax > 0.f ? amax / 448.0f : 1.0f;
}
__global__ __launch_bounds__(256) void pack_fp8_kernel(const float* __restrict__ Wp, int NP, int KP, const float* __restrict__ w_scale,
                                                       uint8_t* __restrict__ W8) {
    const int K32 = KP >> 5;
    const int64_t total = (int64_t)(NP >> 5) * K32 * 64;            // one thread per 16-byte lane slot
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t tile = i >> 6;
        const int s32 = (int)(tile % K32), nt = (int)(tile / K32);
        const int n = nt * 32 + (lane & 31), kh = lane >> 5;
        const float inv = 1.0f / w_scale[n];
        int w[4];
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            const int k0 = s32 * 32 + k16 * 16 + kh * 8;
            float v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = p_layout_at(Wp, KP >> 3, n, k0 + b) * inv;
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
            w[2 * k16] = lo;
            w[2 * k16 + 1] = hi;
        }
        reinterpret_cast<i32x4*>(W8)[i] = i32x4{w[0], w[1], w[2], w[3]};
    }
}
void launch_pack_fp8(const float* Wp, int NP, int KP, uint8_t* W8, float* w_scale, hipStream_t s) {
    hipLaunchKernelGGL(fp8_row_scale_kernel, dim3((NP + 3) / 4), dim3(256), 0, s, Wp, NP, KP, w_scale);
    const int64_t total = (int64_t)(NP >> 5) * (KP >> 5) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_fp8_kernel, dim3(blocks), dim3(256), 0, s, Wp, NP, KP, w_scale, W8);
}

// ---- GEMM ------------------------------------------------------------------------------------------------------------
struct Gemm8Dev {
    GemmArgs a;
    int gm, gn;
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const Gemm8Dev p) {
    constexpr int BM = 128, BN = 128, WN = 2;
    constexpr int STAGE = 2 * 4 * 64;                  // 16-byte slots per operand per stage (2 k32 x 4 tiles x 64 lanes)
    __shared__ __attribute__((aligned(16))) i32x4 As[2 * STAGE];
    __shared__ __attribute__((aligned(16))) i32x4 Ws[2 * STAGE];
    const GemmArgs& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int L = xcd_remap((int)blockIdx.x, p.gm * p.gn);
    const int pm = L / p.gn, pn = L % p.gn;
    const int K32 = (int)(g.lda8 >> 5);                // k32 steps of the whole K
    const int KT = (int)(g.lda8 >> 6);                 // 64-byte stages
    const bool wave_active = (pn * BN + wn * 64) < g.Nw;

    // staging roles: 512 chunks of 16 B per operand per stage, 2 per thread
    const uint8_t* a_ptr[2];
    int a_lds[2];                                      // i32x2 (8-byte) index of the chunk's first half
    const i32x4* w_ptr[2];
    int w_lds[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = q * 256 + tid;
        const int row = c >> 2, ck = c & 3;            // 4 chunks per row: k = 16 ck .. 16 ck + 15 of the stage
        int m = pm * BM + row;
        m = m < g.M ? m : g.M - 1;
        a_ptr[q] = g.A8 + (int64_t)m * g.lda8 + ck * 16;
        // slot (k32 = ck>>1, m-tile = row>>5, lane = row&31 [+32 for the second half]), 8-byte offset k16 = ck&1
        a_lds[q] = ((((ck >> 1) * 4 + (row >> 5)) * 64 + (row & 31)) << 1) + (ck & 1);
        const int ln = c & 63, ntl = (c >> 6) & 3, k32l = c >> 8;
        w_ptr[q] = reinterpret_cast<const i32x4*>(g.W8) + ((int64_t)(pn * 4 + ntl) * K32 + k32l) * 64 + ln;
        w_lds[q] = (k32l * 4 + ntl) * 64 + ln;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    i32x4 ra[2], rw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { ra[q] = *reinterpret_cast<const i32x4*>(a_ptr[q]); rw[q] = w_ptr[q][0]; }
    {
        i32x2* A2 = reinterpret_cast<i32x2*>(As);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            A2[a_lds[q]] = i32x2{ra[q].x, ra[q].y};              // k bytes 0..7  -> lanes 0..31
            A2[a_lds[q] + 64] = i32x2{ra[q].z, ra[q].w};         // k bytes 8..15 -> lanes 32..63 (+32 slots of 16 B)
            Ws[w_lds[q]] = rw[q];
        }
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        const int ktn = kt + 1 < KT ? kt + 1 : kt;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[q] = *reinterpret_cast<const i32x4*>(a_ptr[q] + (int64_t)ktn * 64);
            rw[q] = w_ptr[q][(int64_t)ktn * 2 * 64];
        }
        if (wave_active) {
            const i32x4* Ac = As + cur * STAGE + (2 * wm) * 64 + lane;
            const i32x4* Wc = Ws + cur * STAGE + (2 * wn) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const i32x4 a0 = Ac[s * 4 * 64], a1 = Ac[(s * 4 + 1) * 64];
                const i32x4 w0 = Wc[s * 4 * 64], w1 = Wc[(s * 4 + 1) * 64];
                const long a0l = ((long)(unsigned)a0.y << 32) | (unsigned)a0.x, a0h = ((long)(unsigned)a0.w << 32) | (unsigned)a0.z;
                const long a1l = ((long)(unsigned)a1.y << 32) | (unsigned)a1.x, a1h = ((long)(unsigned)a1.w << 32) | (unsigned)a1.z;
                const long w0l = ((long)(unsigned)w0.y << 32) | (unsigned)w0.x, w0h = ((long)(unsigned)w0.w << 32) | (unsigned)w0.z;
                const long w1l = ((long)(unsigned)w1.y << 32) | (unsigned)w1.x, w1h = ((long)(unsigned)w1.w << 32) | (unsigned)w1.z;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w0l, a0l, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w0l, a1l, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w1l, a0l, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w1l, a1l, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w0h, a0h, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w0h, a1h, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w1h, a0h, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w1h, a1h, acc[1][1], 0, 0, 0);
            }
        }
        {
            i32x2* A2 = reinterpret_cast<i32x2*>(As + (cur ^ 1) * STAGE);
            i32x4* Wn = Ws + (cur ^ 1) * STAGE;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                A2[a_lds[q]] = i32x2{ra[q].x, ra[q].y};
                A2[a_lds[q] + 64] = i32x2{ra[q].z, ra[q].w};
                Wn[w_lds[q]] = rw[q];
            }
        }
        __syncthreads();
    }
    // dequantise: acc[ni][mi][r] belongs to row m(mi, lane & 31) and column n(ni, r) = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
    const int h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int m = pm * BM + wm * 64 + mi * 32 + (lane & 31);
        m = m < g.M ? m : g.M - 1;
        const float sa = g.a_scale[m];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const float* sw = g.w_scale + pn * BN + wn * 64 + ni * 32 + 4 * h;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sw + 8 * gq);
                acc[ni][mi][4 * gq + 0] *= sa * s4.x;
                acc[ni][mi][4 * gq + 1] *= sa * s4.y;
                acc[ni][mi][4 * gq + 2] *= sa * s4.z;
                acc[ni][mi][4 * gq + 3] *= sa * s4.w;
            }
        }
    }
    gemm_epilogue<WN, EPI>(g, acc, pm, pn, wm, wn, lane, BM, BN);
}

template <int EPI>
static void launch8(const GemmArgs& a, hipStream_t s) {
    Gemm8Dev d;
    d.a = a;
    d.gm = (a.M + 127) / 128;
    d.gn = (a.Nw + 127) / 128;
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI>), dim3(d.gm * d.gn), dim3(256), 0, s, d);
}
void launch_gemm_fp8(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_LINEAR: launch8<EPI_LINEAR>(a, s); break;
        case EPI_SWIGLU: launch8<EPI_SWIGLU>(a, s); break;
        case EPI_QKV_ROPE: launch8<EPI_QKV_ROPE>(a, s); break;
        default: break;   // EPI_POWER / EPI_LOGMEL stay on the fp32 kernel (front-end precision)
    }
}

}  // namespace mellow
