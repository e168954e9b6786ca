// fp8 (OCP e4m3) GEMMs for BASELINE config 5, rebuilt in round 6 on the block-scaled matrix instruction of gfx950:
// v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, 64 cycles per SIMD: twice the rate of the K = 16 fp8 form the round-1
// kernel issued).  The reference has no counterpart (it runs fp32 ATen matmuls).  Not bit-exact by construction: the tests hold
// it to an emulation of the same quantisation and report token agreement against the fp32 engine.
//
//   activations  MXFP8: e4m3 elements + one E8M0 scale (a power of two) per 32 consecutive k of a row, scale = 2^ceil(log2(amax /
//                448)) -- no element clips, no cross-workgroup statistic, so EVERY producer (LayerNorm / RMSNorm, GEMM epilogues,
//                the attention epilogue) can emit its output already quantised and in the consumer's LDS order ("AMX" image);
//                the hardware applies the scale inside the MFMA: no dequantisation arithmetic in the loop.
//   weights      e4m3 per output channel (fp32 scale applied to the accumulators in the epilogue), "WMX" image, packed once at load.
//
// Operand order of the instruction, measured with tools/microbench/mx8_probe.hip and mx8_semantics.hip (the ISA text is not in
// this image): lane l = (row l & 31, half h = l >> 5) supplies 32 bytes; byte t is k = 32 (t / 16) + 16 h + t % 16, and byte
// `opsel` of lane (row, h)'s scale VGPR scales the row's k-block h (k = 32 h .. 32 h + 31), i.e. bytes 16 h .. 16 h + 15 of BOTH
// lanes of the row.  Hence the images (16-byte slots):
//   AMX data    [panel = m / 128][s = k / 64][m-tile = (m / 32) % 4][j = (k / 32) % 2][lane = m % 32 + 32 ((k / 16) % 2)]   16 e4m3
//   AMX scales  [panel][s / 4][m-tile][lane = m % 32 + 32 j] uint32, byte s % 4 = E8M0 scale of block (2 s + j) of row m
//   WMX data    [n-tile = n / 32][s][j][lane = n % 32 + 32 ((k / 16) % 2)]
// so a k64 stage of a 128-row panel (8 KiB) or of a 128-column weight tile is eight contiguous 1-KiB pieces: both operands go
// global -> LDS by LDS-DMA exactly like the f32x3 kernel's (gemm_bf16x3.hip, x3q), and the scale words of a tile are staged once.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace mellow {

typedef int i32x8 __attribute__((ext_vector_type(8)));

// ---- standalone quantiser: fp32 row-major -> AMX (for GEMM inputs whose producer does not emit AMX itself) ----------------------
// one lane per (row, 16-k half of a 32-block): lane = (row % 32, h); a wave covers one (m-tile, block), a workgroup four m-tiles
__global__ __launch_bounds__(256) void quant_mx8_kernel(const float* __restrict__ A, int64_t lda, int M, int K, int KT64, int KQ,
                                                        i32x4* __restrict__ img, uint8_t* __restrict__ sc) {
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6;
    const int64_t panel = blockIdx.y;
    const int64_t m = panel * 128 + mt * 32 + (lane & 31);
    const int h = lane >> 5;
    for (int kb = blockIdx.x; kb < 2 * KT64; kb += gridDim.x) {
        const int k0 = kb * 32 + h * 16;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (m < M && k0 + 4 * q < K) x = *reinterpret_cast<const f32x4*>(A + m * lda + k0 + 4 * q);      // K % 4 == 0
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
        amx_store16(img, sc, m, kb, KT64, KQ, v, h);
    }
}
void launch_quant_mx8(const float* A, int64_t lda, int M, int K, void* img, void* sc, hipStream_t s) {
    const int KT64 = (K + 63) / 64, KQ = (KT64 + 3) / 4, panels = (M + 127) / 128;
    int gx = 2 * KT64;
    while (gx > 1 && (int64_t)gx * panels > 16384) gx = (gx + 1) / 2;
    hipLaunchKernelGGL(quant_mx8_kernel, dim3(gx, panels), dim3(256), 0, s, A, lda, M, K, KT64, KQ, reinterpret_cast<i32x4*>(img),
                       reinterpret_cast<uint8_t*>(sc));
}

// ---- weight packing: fp32 P-layout -> WMX + one fp32 scale per packed weight row -------------------------------------------------
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
__global__ __launch_bounds__(256) void pack_wmx_kernel(const float* __restrict__ Wp, int NP, int KP, int KT64, const float* __restrict__ w_scale,
                                                       i32x4* __restrict__ W8) {
    const int64_t total = (int64_t)(NP >> 5) * KT64 * 2 * 64;            // one thread per 16-byte slot
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), j = (int)((i >> 6) & 1);
        const int64_t t = i >> 7;
        const int s = (int)(t % KT64), nt = (int)(t / KT64);
        const int n = nt * 32 + (lane & 31), k0 = 64 * s + 32 * j + 16 * (lane >> 5);
        const float inv = 1.0f / w_scale[n];
        i32x4 w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) v[b] = (k0 + 4 * q + b < KP) ? p_layout_at(Wp, KP >> 3, n, k0 + 4 * q + b) * inv : 0.f;
            w[q] = mx8_pack4(v[0], v[1], v[2], v[3]);
        }
        W8[i] = w;
    }
}
// W8 holds (NP / 32) * ceil(KP / 64) * 2 KiB
void launch_pack_fp8(const float* Wp, int NP, int KP, uint8_t* W8, float* w_scale, hipStream_t s) {
    hipLaunchKernelGGL(fp8_row_scale_kernel, dim3((NP + 3) / 4), dim3(256), 0, s, Wp, NP, KP, w_scale);
    const int KT64 = (KP + 63) / 64;
    const int64_t total = (int64_t)(NP >> 5) * KT64 * 128;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_wmx_kernel, dim3(blocks), dim3(256), 0, s, Wp, NP, KP, KT64, w_scale, reinterpret_cast<i32x4*>(W8));
}

// ---- GEMM ------------------------------------------------------------------------------------------------------------------------
struct GemmMxDev {
    GemmArgs a;
    int gm, gn, ng;
};
__device__ __forceinline__ void glds16_mx(const void* base, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(base) : "memory");
}
// 128 x 128 tile, 4 waves (64 x 64 each: 4 accumulator tiles), one k64 step per stage (8 KiB per operand), ring of 3 stages filled
// by LDS-DMA (waves 0, 1: the activation stage, waves 2, 3: the weight stage; four 1-KiB pieces per wave and step).  Per step and
// wave: 8 ds_read_b128 feed 4 MFMAs of 64 cycles.  D[n][m] like every GEMM of the engine, so the epilogues are shared.
template <int EPI, bool C16>
__global__ __launch_bounds__(256, 2) void gemm_mx8_kernel(const GemmMxDev p) {
    constexpr int BM = 128, BN = 128, WN = 2, NST = 3;
    constexpr int STAGE = 2 * 4 * 64;                                      // 16-byte slots per operand per stage
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_mx[];
    i32x4* As = smem_mx;
    i32x4* Ws = smem_mx + NST * STAGE;
    uint32_t* Sc = reinterpret_cast<uint32_t*>(smem_mx + 2 * NST * STAGE);  // [KQ][4 m-tiles][64 lanes]
    __shared__ float rs_rows[BM];
    const GemmArgs& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int pm, pn;
    if (p.ng) {         // weight-stationary per XCD for wide outputs (gemm_bf16x3.hip: gemm_x3q_kernel)
        const int npx = p.gn >> 3, b = (int)blockIdx.x, j = b >> 3;
        pm = j / npx; pn = (b & 7) * npx + j % npx;
    } else {
        const int L = xcd_remap((int)blockIdx.x, p.gm * p.gn);
        pm = L / p.gn; pn = L % p.gn;
    }
    const int KT = (int)(g.lda8 >> 6);                                     // k64 steps
    const int KQ = (KT + 3) >> 2;
    // scale words of this panel -> LDS (ordinary loads: they are complete before the first LDS-DMA piece is issued, so the
    // hand-counted vmcnt of the loop sees LDS-DMA pieces only); row scales of the norm-free chaining likewise
    {
        const uint32_t* src = g.a_sc + (int64_t)pm * KQ * 256;
        for (int i = tid; i < KQ * 256; i += 256) Sc[i] = src[i];
    }
    if (g.rs_ssq && tid < BM) {
        int64_t m = (int64_t)pm * BM + tid;
        m = m < g.M ? m : g.M - 1;
        const float* sp = g.rs_ssq + m * g.rs_parts;
        float ss = 0.f;
        for (int q = 0; q < g.rs_parts; ++q) ss += sp[q];                  // fixed order
        rs_rows[tid] = 1.0f / sqrtf(ss / g.rs_dim + g.rs_eps);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool isA = wave < 2;
    const int c0 = (wave & 1) * 4;                                         // pieces c0 .. c0 + 3 of the 8 of a stage
    const char* base = isA ? reinterpret_cast<const char*>(g.A8) + ((int64_t)pm * KT * 8 + c0) * 1024
                           : reinterpret_cast<const char*>(g.W8);
    const uint32_t kstep = isA ? 8192u : 2048u;
    uint32_t voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;                                              // W piece c = n-tile (c >> 1), half j (c & 1)
        voff[j] = isA ? (uint32_t)(j * 1024 + lane * 16)
                      : (uint32_t)((((int64_t)(pn * 4 + (c >> 1)) * KT * 2 + (c & 1)) * 64 + lane) * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_mx + (isA ? 0u : (uint32_t)(NST * STAGE * 16)) + (uint32_t)c0 * 1024u;
#define MX_ISSUE(T, ST)                                                                          \
    {                                                                                            \
        const int t_ = (T) < KT ? (T) : KT - 1;                                                  \
        const char* b_ = base + (int64_t)t_ * kstep;                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) glds16_mx(b_, voff[j], lds0 + (uint32_t)((ST) * STAGE * 16 + j * 1024)); \
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool idle = pn * BN + wn * 64 >= g.Nw;            // wave-uniform: its 64 columns lie beyond the weight's rows
    MX_ISSUE(0, 0)
    MX_ISSUE(1, 1)
    int sa0 = 0, sa1 = 0;
    // step T in stage ST: byte OPS of the scale words; the weight's scale operand is the constant 127 = 2^0
#define MX_STEP(T, ST, OPS)                                                                      \
    if ((T) < KT) {                                                                              \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      /* the pieces of step T have landed (those of T + 1 may fly) */ \
        __builtin_amdgcn_s_barrier();                         /* ... for every wave; stage (T + 2) % 3 = (T - 1) % 3 is free */ \
        MX_ISSUE((T) + 2, ((ST) + 2) % NST)                                                      \
        if (!idle) {                                                                             \
            const i32x4* Ac = As + (ST) * STAGE + (2 * wm) * 128 + lane;                         \
            const i32x4* Wc = Ws + (ST) * STAGE + (2 * wn) * 128 + lane;                         \
            i32x8 fa[2], fw[2];                                                                  \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                      \
                const i32x4 a0 = Ac[t * 128], a1 = Ac[t * 128 + 64], w0 = Wc[t * 128], w1 = Wc[t * 128 + 64]; \
                fa[t] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};           \
                fw[t] = i32x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};           \
            }                                                                                    \
            if ((OPS) == 0) { sa0 = (int)Sc[(((T) >> 2) * 4 + 2 * wm) * 64 + lane]; sa1 = (int)Sc[(((T) >> 2) * 4 + 2 * wm + 1) * 64 + lane]; } \
            acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[0], fa[0], acc[0][0], 0, 0, 0, 0x7F7F7F7F, OPS, sa0); \
            acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[0], fa[1], acc[0][1], 0, 0, 0, 0x7F7F7F7F, OPS, sa1); \
            acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[1], fa[0], acc[1][0], 0, 0, 0, 0x7F7F7F7F, OPS, sa0); \
            acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[1], fa[1], acc[1][1], 0, 0, 0, 0x7F7F7F7F, OPS, sa1); \
        }                                                                                        \
    }
    // 12 steps per trip: the stage index (T % 3) and the scale byte (T % 4) are compile-time constants
    for (int kt = 0; kt < KT; kt += 12) {
        MX_STEP(kt + 0, 0, 0)  MX_STEP(kt + 1, 1, 1)  MX_STEP(kt + 2, 2, 2)   MX_STEP(kt + 3, 0, 3)
        MX_STEP(kt + 4, 1, 0)  MX_STEP(kt + 5, 2, 1)  MX_STEP(kt + 6, 0, 2)   MX_STEP(kt + 7, 1, 3)
        MX_STEP(kt + 8, 2, 0)  MX_STEP(kt + 9, 0, 1)  MX_STEP(kt + 10, 1, 2)  MX_STEP(kt + 11, 2, 3)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the clamped re-loads behind the last step
#undef MX_STEP
#undef MX_ISSUE
    if (idle) return;
    // per-output-channel weight scale: acc[ni][mi][r] is column n(ni, r) = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of its n-tile
    const int h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const float* sw = g.w_scale + pn * BN + wn * 64 + ni * 32 + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sw + 8 * gq);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                acc[ni][mi][4 * gq + 0] *= s4.x;
                acc[ni][mi][4 * gq + 1] *= s4.y;
                acc[ni][mi][4 * gq + 2] *= s4.z;
                acc[ni][mi][4 * gq + 3] *= s4.w;
            }
        }
    }
    gemm_epilogue<WN, EPI, C16, true>(g, acc, pm, pn, wm, wn, lane, BM, BN, g.rs_ssq ? rs_rows : nullptr);
}

template <int EPI, bool C16 = false>
static void launch8(const GemmArgs& a, hipStream_t s) {
    GemmMxDev d;
    d.a = a;
    d.gm = (a.M + 127) / 128;
    d.gn = (a.Nw + 127) / 128;
    d.ng = (d.gn % 8 == 0 && d.gn >= 16) ? 1 : 0;
    const int KT = (int)(a.lda8 >> 6), KQ = (KT + 3) / 4;
    const size_t lds = (size_t)2 * 3 * (2 * 4 * 64) * 16 + (size_t)KQ * 1024;          // 48 KiB of stages + the panel's scale words
    set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_mx8_kernel<EPI, C16>), 48 * 1024 + 24 * 1024);
    hipLaunchKernelGGL((gemm_mx8_kernel<EPI, C16>), dim3(d.gm * d.gn), dim3(256), lds, s, d);
}
// A8 / a_sc = AMX image + scale words of the activation (lda8 = K rounded up to 64), W8 / w_scale = WMX image + row scales
void launch_gemm_fp8(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_LINEAR: if (a.c16) launch8<EPI_LINEAR, true>(a, s); else launch8<EPI_LINEAR>(a, s); break;
        case EPI_SWIGLU: launch8<EPI_SWIGLU>(a, s); break;
        case EPI_QKV_ROPE: launch8<EPI_QKV_ROPE>(a, s); break;
        default: break;   // EPI_POWER / EPI_LOGMEL stay on the fp32 kernel (front-end precision)
    }
}

}  // namespace mellow
