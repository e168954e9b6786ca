// fp32-in / fp32-accumulate GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), the workhorse of
// the encoder and of LM prefill.  Replaces every nn.Linear / conv-as-matmul ATen call of the reference
// hot path: htsat.py:130-136, 304, 330, 496-497, 774, 952;
// mellow.py:49-50; transformers LlamaAttention / LlamaMLP projections.
//
// Design (MI355X-first, not a CUDA tiling):
//  * exact fp32: gfx950 has no TF32; v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain at the full
//    157 TFLOP/s fp32 rate, so the "bit-closeness to the fp32 reference" mode costs nothing vs VALU.
//  * both operands live in LDS in MFMA *fragment order*: one lane-linear float4 per lane feeds four
//    MFMAs (k-pairs (k0+j, k0+4+j)), so every LDS read is a conflict-free ds_read_b128 and a wave
//    issues one LDS read per four 64-cycle MFMAs.
//  * weights are pre-tiled once at load time into that order (P-layout, kernels.h), so the W stream is
//    a plain contiguous 4 KiB copy per (n-tile, k-tile); activations are staged from row-major global
//    memory as full 128-byte lines (8 rows x 128 B per wave instruction).
//  * the MFMA is issued as D[n][m] (weight = A operand, activation = B operand): a lane then owns one
//    output ROW and four consecutive columns per accumulator quad -> float4 epilogue stores, and the
//    two halves of a (re,im) / (gate,up) / RoPE pair sit in the same lane.
//  * 4 waves per workgroup, 64x64 per wave (4 accumulators), block 128x128 or 256x64, BK = 32,
//    double-buffered LDS fed by a two-tile-deep register prefetch (one barrier per k-tile),
//    MFMA fragments double-buffered in registers.
//  * XCD-aware block order: all n-blocks of an m-panel run on one XCD (A panel read once per L2).
#include <cstdlib>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace mellow {

struct GemmDev {
    GemmArgs a;
    int gm, gn;
};
// developer instrumentation (-DMELLOW_KDEBUG): slot 7 of the debug buffer = main-loop {shader clocks, 100 MHz ticks,
// k-tiles, grid} of a mid-grid workgroup of the last GEMM launch
__device__ uint64_t* g_kdbg = nullptr;
void set_gemm_debug_buffer(uint64_t* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kdbg), &p, sizeof(p)); }

// pin a wave-uniform pointer in scalar registers so that loads through it use the `saddr + 32-bit voffset` form
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

template <int WM, int WN, int EPI, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmDev p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int MT = BM / 32, NTB = BN / 32;      // 32-row tiles per block
    constexpr int KS = BK / 8;                      // 8-wide k sub-tiles per k-tile
    constexpr int CH = BK / 4;                      // float4 chunks per row per k-tile
    constexpr int A_F4 = BM * CH / 256;             // float4 per thread per k-tile (A)
    constexpr int W_F4 = BN * CH / 256;             // float4 per thread per k-tile (W)
    constexpr int A_STAGE = BM * CH;                // float4 per stage
    constexpr int W_STAGE = BN * CH;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* As = smem;                              // [2][A_STAGE]
    float4* Ws = smem + 2 * A_STAGE;                // [2][W_STAGE]

    const GemmArgs& g = p.a;
    const int tid = threadIdx.x;
#ifdef MELLOW_KDEBUG
    const uint64_t dbg_c0 = __builtin_readcyclecounter(), dbg_r0 = wall_clock64();
#endif
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: xcd_remap gives every XCD (private L2) a contiguous range of logical ranks, m-major, so
    // the n-blocks of one activation panel run together on one XCD.  (An n-major walk inside per-XCD panel groups was
    // measured too: no fewer memory-side fetches, 2 % slower.)
    const int L = xcd_remap((int)blockIdx.x, p.gm * p.gn);
    const int pm = L / p.gn, pn = L % p.gn;
    const int nt0 = pn * NTB;
    const int K8 = g.K >> 3;
    const int KT = g.K / BK;
    // a wave whose 64 output columns are all beyond the weight's rows only helps with staging (frees the MFMA pipe)
    const bool wave_active = (pn * BN + wn * 64) < g.Nw;

    // ---- per-thread load descriptors -------------------------------------------------------------
    // global addresses = wave-uniform base (SGPR pair, advanced per k-tile on the scalar unit) + a 32-bit per-lane byte
    // offset: the loads use the saddr form and the loop carries no 64-bit vector address arithmetic
    // (offsets are relative to the workgroup's first row, so they stay far below 4 GiB however large the matrix is)
    uint32_t a_off[A_F4];
    int a_lds[A_F4];
    auto row_off = [&](int m) -> int64_t {
        m = m < g.M ? m : g.M - 1;
        return g.a_mode == A_FRAMES ? (int64_t)(m / g.fpc) * g.clip_stride + (int64_t)(m % g.fpc) * g.hop : (int64_t)m * g.lda;
    };
    const int64_t a_row0 = row_off(pm * BM);            // monotonic in m for both addressing modes
#pragma unroll
    for (int q = 0; q < A_F4; ++q) {
        const int idx = q * 256 + tid;
        // 8 consecutive lanes -> 8 rows (conflict-free 128 B LDS write), next lane bits -> the CH chunks of a row
        const int row = (idx / (8 * CH)) * 8 + (idx & 7);
        const int chunk = (idx >> 3) % CH;
        a_off[q] = (uint32_t)((row_off(pm * BM + row) - a_row0 + chunk * 4) * (int64_t)sizeof(float));
        a_lds[q] = ((chunk >> 1) * MT + (row >> 5)) * 64 + (row & 31) + 32 * (chunk & 1);
    }
    uint32_t w_off[W_F4];
    int w_lds[W_F4];
#pragma unroll
    for (int q = 0; q < W_F4; ++q) {
        const int idx = q * 256 + tid;
        const int ln = idx & 63, k8 = (idx >> 6) % KS, ntl = idx / (64 * KS);
        w_off[q] = (uint32_t)((((int64_t)ntl * K8 + k8) * 64 + ln) * 16);
        w_lds[q] = (k8 * NTB + ntl) * 64 + ln;
    }

    f32x16 acc[2][2];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop: 3-deep software pipeline -------------------------------------------------------------------
    //   G(t): global -> registers,  S(t): registers -> LDS[t & 1],  C(t): MFMAs from LDS[t & 1]
    //   iteration t:  G(t+2) ; C(t) ; S(t+1) ; barrier
    // A tile's global loads are issued two compute phases before they are needed: with one workgroup per CU (small
    // grids: encoder stage 3, tscam) nothing else hides the L2/HBM latency, +20..30 % there; neutral on the large LM
    // shapes, where the second workgroup of the CU already covered it.  Two register sets, k-loop unrolled by two so
    // that every array index is static.  Prefetches past the last k-tile are skipped (uniform branch).
    // native vector type, not HIP's float4 struct: struct copies become llvm.memcpy through a private alloca that the
    // compiler did not promote for this loop shape (every prefetch went through scratch memory)
    const char* a_base = reinterpret_cast<const char*>(g.A + a_row0);
    const char* w_base = reinterpret_cast<const char*>(g.Wp) + (size_t)nt0 * K8 * 64 * 16;
    f32x4 ra0[A_F4], rw0[W_F4], ra1[A_F4], rw1[W_F4];
#define MELLOW_GLOAD(RA, RW, T)                                                                      \
    if ((T) < KT) {                                                                                  \
        const int t_ = (T);                                                                          \
        const char* ab_ = sgpr_ptr(a_base + (size_t)t_ * (BK * 4));                                  \
        const char* wb_ = sgpr_ptr(w_base + (size_t)t_ * (KS * 64 * 16));                            \
        _Pragma("unroll") for (int q = 0; q < A_F4; ++q)                                             \
            RA[q] = *reinterpret_cast<const f32x4*>(ab_ + a_off[q]);                                      \
        _Pragma("unroll") for (int q = 0; q < W_F4; ++q)                                             \
            RW[q] = *reinterpret_cast<const f32x4*>(wb_ + w_off[q]);                                      \
    }
#define MELLOW_LSTORE(RA, RW, STAGE)                                                                 \
    {                                                                                                \
        f32x4* An = reinterpret_cast<f32x4*>(As + (STAGE) * A_STAGE);                                \
        f32x4* Wn = reinterpret_cast<f32x4*>(Ws + (STAGE) * W_STAGE);                                \
        _Pragma("unroll") for (int q = 0; q < A_F4; ++q) An[a_lds[q]] = RA[q];                       \
        _Pragma("unroll") for (int q = 0; q < W_F4; ++q) Wn[w_lds[q]] = RW[q];                       \
    }
#define MELLOW_MFMA16(w0, w1, a0, a1)                                                                \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, a0.x, acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, a1.x, acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, a0.x, acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, a1.x, acc[1][1], 0, 0, 0);                \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, a0.y, acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, a1.y, acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, a0.y, acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, a1.y, acc[1][1], 0, 0, 0);                \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, a0.z, acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, a1.z, acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, a0.z, acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, a1.z, acc[1][1], 0, 0, 0);                \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, a0.w, acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, a1.w, acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, a0.w, acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, a1.w, acc[1][1], 0, 0, 0);
    // fragments double-buffered in registers: the LDS reads of sub-step k8+1 are issued before the 16 MFMAs of k8.
    // STORE (the LDS refill of the other stage, or nothing) is placed between the two MFMA groups of the LAST sub-step
    // pair: its ds_writes drain while the matrix pipe works, so the exposed part of the refill is the barrier and the
    // first fragment reads.  Where the remaining ~20 % of the loop goes was measured with tools/microbench/gemm_ablate.hip
    // (same loop, switchable parts): MFMA + LDS reads + barrier alone sustain 148-151 TFLOP/s; adding the
    // LDS refill or the global prefetch costs ~5 % EACH even when a different, idle wave issues them (8-wave ping-pong
    // variant), i.e. it is contention inside the CU, not exposed latency, and no instruction order removes it.
#define MELLOW_COMPUTE(STAGE, STORE)                                                                 \
    if (wave_active) {                                                                               \
        const f32x4* Ac = reinterpret_cast<const f32x4*>(As + (STAGE) * A_STAGE + (2 * wm) * 64 + lane); \
        const f32x4* Wc = reinterpret_cast<const f32x4*>(Ws + (STAGE) * W_STAGE + (2 * wn) * 64 + lane); \
        f32x4 xa0 = Ac[0], xa1 = Ac[64], xw0 = Wc[0], xw1 = Wc[64];                                  \
        __builtin_amdgcn_s_setprio(2);                                                               \
        _Pragma("unroll") for (int k8 = 0; k8 < KS; k8 += 2) {                                       \
            const f32x4 ya0 = Ac[((k8 + 1) * MT) * 64], ya1 = Ac[((k8 + 1) * MT + 1) * 64];          \
            const f32x4 yw0 = Wc[((k8 + 1) * NTB) * 64], yw1 = Wc[((k8 + 1) * NTB + 1) * 64];        \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            MELLOW_MFMA16(xw0, xw1, xa0, xa1)                                                        \
            if (k8 + 2 < KS) {                                                                       \
                xa0 = Ac[((k8 + 2) * MT) * 64]; xa1 = Ac[((k8 + 2) * MT + 1) * 64];                  \
                xw0 = Wc[((k8 + 2) * NTB) * 64]; xw1 = Wc[((k8 + 2) * NTB + 1) * 64];                \
            } else {                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                   \
                STORE                                                                                \
            }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            MELLOW_MFMA16(yw0, yw1, ya0, ya1)                                                        \
        }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } else {                                                                                         \
        STORE                                                                                        \
    }
    static_assert(KS % 2 == 0, "k sub-steps are processed in pairs");

    MELLOW_GLOAD(ra0, rw0, 0)
    MELLOW_GLOAD(ra1, rw1, 1)
    MELLOW_LSTORE(ra0, rw0, 0)
    __syncthreads();
#ifdef MELLOW_KDEBUG
    uint64_t dbg_tc = 0, dbg_ts = 0, dbg_t = __builtin_readcyclecounter();
#define MELLOW_DBG_MARK(ACC) { const uint64_t n_ = __builtin_readcyclecounter(); ACC += n_ - dbg_t; dbg_t = n_; }
#else
#define MELLOW_DBG_MARK(ACC)
#endif
    for (int kt = 0; kt + 1 < KT; kt += 2) {       // no conditional inside: the register arrays must stay unconditional
        MELLOW_GLOAD(ra0, rw0, kt + 2)
        MELLOW_DBG_MARK(dbg_ts)
        MELLOW_COMPUTE(0, MELLOW_LSTORE(ra1, rw1, 1))
        MELLOW_DBG_MARK(dbg_tc)
        __syncthreads();
        MELLOW_GLOAD(ra1, rw1, kt + 3)
        MELLOW_DBG_MARK(dbg_ts)
        MELLOW_COMPUTE(1, MELLOW_LSTORE(ra0, rw0, 0))
        MELLOW_DBG_MARK(dbg_tc)
        __syncthreads();
    }
    if (KT & 1) { MELLOW_COMPUTE(0, ) }             // odd tile count: the last tile sits in stage 0
#undef MELLOW_GLOAD
#undef MELLOW_LSTORE
#undef MELLOW_MFMA16
#undef MELLOW_COMPUTE

#ifdef MELLOW_KDEBUG
    if (g_kdbg && tid == 0 && blockIdx.x == gridDim.x / 2 && acc[0][0][0] == acc[0][0][0]) {
        g_kdbg[56] = __builtin_readcyclecounter() - dbg_c0;
        g_kdbg[57] = wall_clock64() - dbg_r0;
        g_kdbg[58] = (uint64_t)KT;
        g_kdbg[59] = (uint64_t)gridDim.x;
        g_kdbg[60] = dbg_tc;
        g_kdbg[61] = dbg_ts;
        g_kdbg[62] = __builtin_amdgcn_s_getreg(0xf804);    // HW_REG_HW_ID bits 15:0
    }
#endif
    gemm_epilogue<WN, EPI>(g, acc, pm, pn, wm, wn, lane, BM, BN);
}

template <int WM, int WN, int EPI, int BK>
static void launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    GemmDev d;
    d.a = a;
    d.gm = (a.M + BM - 1) / BM;
    d.gn = (a.Nw + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_f32_kernel<WM, WN, EPI, BK>), lds);      // per (function, device)
    hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI, BK>), dim3(d.gm * d.gn), dim3(256), lds, s, d);
}

template <int EPI>
static void launch_epi(const GemmArgs& a, hipStream_t s) {
    // 128x128 everywhere except the narrow outputs (mel: 64 columns), where 256x64 wastes no padded columns.
    // (Measured and not kept: 256x64 whenever it pads less, and a 16-wide k-tile with 32 KB of LDS -- both slower.)
    if (a.Nw <= 64 && a.M >= 256) launch_cfg<4, 1, EPI, 32>(a, s);
    else launch_cfg<2, 2, EPI, 32>(a, s);
}

void launch_gemm(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_LINEAR: launch_epi<EPI_LINEAR>(a, s); break;
        case EPI_POWER: launch_epi<EPI_POWER>(a, s); break;
        case EPI_LOGMEL: launch_epi<EPI_LOGMEL>(a, s); break;
        case EPI_SWIGLU: launch_epi<EPI_SWIGLU>(a, s); break;
        case EPI_QKV_ROPE: launch_epi<EPI_QKV_ROPE>(a, s); break;
    }
}

double gemm_flops(const GemmArgs& a) { return 2.0 * (double)a.M * (double)a.K * (double)a.Nw; }

// ---- weight packing (runs once per tensor at load time) ---------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int N, int K,
                                   int64_t ldw, float* __restrict__ out, int NP, int KP) {
    // one thread per packed float4
    const int64_t total = (int64_t)(NP / 32) * (KP / 8) * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t tile = i >> 6;
        const int k8 = (int)(tile % (KP / 8));
        const int nt = (int)(tile / (KP / 8));
        const int n = nt * 32 + (lane & 31);
        const int k0 = k8 * 8 + 4 * (lane >> 5);
        const float* src = nullptr;
        int row = -1;
        if (w1 == nullptr) {
            if (n < N) { src = w0; row = n; }
        } else {
            // pairs-interleaved: 64-row group j = n/64; first 32 rows from w0, next 32 from w1
            const int j = n >> 6, i32 = n & 31, r = j * 32 + i32;
            if (r < N) { src = (n & 32) ? w1 : w0; row = r; }
        }
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (src) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < K) v[j] = src[(int64_t)row * ldw + k0 + j];
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

void launch_pack_weight(const float* w, int N, int K, int64_t ldw, float* out, int NP, int KP, hipStream_t s) {
    const int64_t total = (int64_t)(NP / 32) * (KP / 8) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, s, w, (const float*)nullptr, N, K, ldw, out, NP, KP);
}
void launch_pack_weight_pairs(const float* w0, const float* w1, int N, int K, int64_t ldw, float* out, int NP, int KP,
                              hipStream_t s) {
    const int64_t total = (int64_t)(NP / 32) * (KP / 8) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, s, w0, w1, N, K, ldw, out, NP, KP);
}

}  // namespace mellow
