// fp32 GEMM on the bf16 matrix pipe by exact operand splitting ("bf16x3"): every fp32 operand value is the exact sum of
// three bf16 numbers, a = a1 + a2 + a3 (8 + 8 + 8 significand bits), so a*b = sum_ij ai*bj with every partial product
// exact in fp32; the products run on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) with fp32 accumulation.
// Six partial products: a2*b3, a3*b2, a3*b3 are dropped (< 2^-23 |a*b| in total, the order of one fp32 rounding; the nine-term
// form measured the same error against fp64 and was removed in round 6 together with its pre-split debug kernel).
// Accumulation order differs from the k-ordered fmaf chain of v_mfma_f32_32x32x2_f32, so results are fp32-accurate, not
// bit-identical to that kernel; tools/f32x3_check.py measures all three against an fp64 product.
// This is the engine's default numeric mode (MELLOW_PRECISION_F32X3, include/mellow_hip.h); the parity suite runs in it and in
// the exact fp32 MFMA mode (MELLOW_PRECISION_F32, gemm_f32.hip) with the same tolerances and exact tokens.
//
//   pack_bf16x3  W fp32 P-layout -> PB [n/32][k/16][3][lane][8 bf16]               (once per tensor at load time)
//   gemm         128 x 128 tile, 4 waves, one k16 step per stage (12 KiB per operand), double-buffered LDS in
//                fragment order, 12 ds_read_b128 feed 24 MFMAs; shared epilogues (gemm_epilogue.h).
#include <cstdlib>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace mellow {


__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const float* __restrict__ Wp, int NP, int KP, i32x4* __restrict__ PB) {
    const int K16 = KP >> 4, K8 = KP >> 3;
    const int64_t total = (int64_t)(NP >> 5) * K16 * 64;            // one thread per (n-tile, k16, lane): 3 x 16 bytes
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t tile = i >> 6;
        const int s16 = (int)(tile % K16), nt = (int)(tile / K16);
        const int n = nt * 32 + (lane & 31), k0 = s16 * 16 + (lane >> 5) * 8;
        float v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int k = k0 + b;
            v[b] = Wp[(((int64_t)(n >> 5) * K8 + (k >> 3)) * 64 + (n & 31) + 32 * ((k >> 2) & 1)) * 4 + (k & 3)];
        }
        i32x4 p0, p1, p2;
        split8(v, p0, p1, p2);
        i32x4* o = PB + (tile * 3) * 64 + lane;
        o[0] = p0; o[64] = p1; o[128] = p2;
    }
}
void launch_pack_bf16x3(const float* Wp, int NP, int KP, void* PB, hipStream_t s) {
    const int64_t total = (int64_t)(NP >> 5) * (KP >> 4) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(blocks), dim3(256), 0, s, Wp, NP, KP, reinterpret_cast<i32x4*>(PB));
}

struct GemmBDev {
    GemmArgs a;
    int gm, gn;
    int ks;                 // split-K: workgroups per output tile (1 = none); tile = L % (gm gn), split = L / (gm gn)
    int ng = 0;             // x3q, wide outputs (gn % 8 == 0, gn >= 16, no split-K): XCD x owns the n-tiles [x gn/8, (x+1) gn/8) for EVERY row
                            // panel -- its gn/8 weight tiles stay in its 4 MiB L2 while the panels stream through (below)
};

// Split-K for launches that would leave most of the chip idle (<= 256 output tiles: the tail GEMMs of the encoder).  Two
// launches: every (tile, split) workgroup of the GEMM kernel writes its 128 x 128 fp32 partial tile to the workspace
// (thread-major 16-byte pieces, coalesced) and leaves; splitk_finish_kernel (one workgroup per tile, the same thread -> element
// mapping) sums the partials in split order -- a fixed summation order -- and runs the epilogue.  The kernel boundary is the
// hand-over: the in-kernel forms (the last split to arrive at a counter finishes the tile) were either slow (__threadfence() on
// both sides: 80-100 us per launch) or wrong (write-through stores + L2-bypassing loads), tools/experiments/r04_encoder_notes.md.
__device__ __forceinline__ void splitk_store(const GemmBDev& p, const f32x16 (&acc)[2][2], int tile, int split, int tid) {
    f32x4* ws = reinterpret_cast<f32x4*>(p.a.sk_ws) + ((int64_t)(tile * p.ks + split) * 16) * 256 + tid;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ws[((i * 2 + j) * 4 + r) * 256] = f32x4{acc[i][j][4 * r], acc[i][j][4 * r + 1], acc[i][j][4 * r + 2], acc[i][j][4 * r + 3]};
}
template <int EPI>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const GemmBDev p) {
    constexpr int BM = 128, BN = 128, WN = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int pm = tile / p.gn, pn = tile % p.gn;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const f32x4* rs = reinterpret_cast<const f32x4*>(p.a.sk_ws) + ((int64_t)tile * p.ks * 16) * 256 + tid;
    for (int sp = 0; sp < p.ks; ++sp, rs += 16 * 256) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 v = __builtin_nontemporal_load(rs + ((i * 2 + j) * 4 + r) * 256);
                    acc[i][j][4 * r] += v.x; acc[i][j][4 * r + 1] += v.y; acc[i][j][4 * r + 2] += v.z; acc[i][j][4 * r + 3] += v.w;
                }
    }
    gemm_epilogue<WN, EPI>(p.a, acc, pm, pn, wave >> 1, wave & 1, lane, BM, BN);
}
// host side: splits for a launch of `tiles` output tiles and KT k16 steps: at most 512 workgroups, at least 32 steps per split,
// and launches that already cover half of the chip are split only when they are long (measured at B = 32, round 4: the
// token-semantic conv 80 tiles x 288 steps 194 -> 89 us, stage-3 fc2 192 x 192 142 -> 113; stage-3 proj 192 x 48 and the
// patch-merging reduction 192 x 96: +-0 / +5 us, so those stay whole)
static int splitk_for(const GemmArgs& a, int tiles, int KT) {
    if (!a.sk_ws || tiles > 256 || tiles > a.sk_tiles) return 1;
    if (tiles > 128 && KT < 128) return 1;
    int s = 512 / tiles;
    if (s > KT / 32) s = KT / 32;
    if (s > a.sk_max) s = a.sk_max;
    return s < 2 ? 1 : s;
}

#define MELLOW_BF(W, A, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, A), ACC, 0, 0, 0);

// ---- fused variant: A stays fp32 in global memory and in LDS and is split in registers after the fragment read -----------
// No pre-pass, A traffic identical to the fp32 kernel; only the (pre-split, PB-layout) weight costs 6 bytes per element.
// stage = one k32: A [k16 2][float4 half 2][m-tile 4][64 lanes] x 16 B = 16 KiB (lane-linear b128 fragment reads),
//                  W [k16 2][piece 3][n-tile 4][64] x 16 B = 24 KiB; two stages = 80 KiB (dynamic LDS), 2 workgroups per CU.
template <int EPI, int KS16>
__global__ __launch_bounds__(256) void gemm_bf16x3f_kernel(const GemmBDev p) {
    constexpr int BM = 128, BN = 128, WN = 2;
    constexpr int A_STAGE = KS16 * 2 * 4 * 64, W_STAGE = KS16 * 3 * 4 * 64;     // 16-byte slots
    constexpr int NA = KS16 * 2, NW = KS16 * 3;                             // float4 / 16-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_bf[];
    f32x4* As = reinterpret_cast<f32x4*>(smem_bf);                          // [2][A_STAGE]
    i32x4* Ws = smem_bf + 2 * A_STAGE;                                      // [2][W_STAGE]
    const GemmArgs& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int L = xcd_remap((int)blockIdx.x, p.gm * p.gn);
    const int pm = L / p.gn, pn = L % p.gn;
    const int K16 = g.K >> 4, KT = K16 / KS16;
    const bool wave_active = (pn * BN + wn * 64) < g.Nw;
    const i32x4* PB = reinterpret_cast<const i32x4*>(g.W8);

    const float* a_ptr[NA];
    int a_lds[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int idx = q * 256 + tid;
        constexpr int CHK = 4 * KS16;                    // float4 chunks per row per stage
        const int row = (idx / (8 * CHK)) * 8 + (idx & 7);   // 8 consecutive lanes -> 8 rows, next lane bits -> the chunks of a row
        const int chunk = (idx >> 3) % CHK;              // k = 4 chunk .. 4 chunk + 3 of the stage
        int m = pm * BM + row;
        m = m < g.M ? m : g.M - 1;
        a_ptr[q] = g.A + (int64_t)m * g.lda + chunk * 4;
        const int s16 = chunk >> 2, kh = (chunk >> 1) & 1, half4 = chunk & 1;
        a_lds[q] = ((s16 * 2 + half4) * 4 + (row >> 5)) * 64 + (row & 31) + 32 * kh;
    }
    const i32x4* w_ptr[NW];
    int w_lds[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int c = q * 256 + tid;
        const int ntl = c / (192 * KS16), rem = c % (192 * KS16), s16 = rem / 192, r2 = rem % 192;
        w_ptr[q] = PB + ((int64_t)(pn * 4 + ntl) * K16 + s16) * 192 + r2;
        w_lds[q] = s16 * (3 * 4 * 64) + ((r2 >> 6) * 4 + ntl) * 64 + (r2 & 63);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA];
    i32x4 rw[NW];
#pragma unroll
    for (int q = 0; q < NA; ++q) ra[q] = *reinterpret_cast<const f32x4*>(a_ptr[q]);
#pragma unroll
    for (int q = 0; q < NW; ++q) rw[q] = w_ptr[q][0];
#pragma unroll
    for (int q = 0; q < NA; ++q) As[a_lds[q]] = ra[q];
#pragma unroll
    for (int q = 0; q < NW; ++q) Ws[w_lds[q]] = rw[q];
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        const int ktn = kt + 1 < KT ? kt + 1 : kt;
#pragma unroll
        for (int q = 0; q < NA; ++q) ra[q] = *reinterpret_cast<const f32x4*>(a_ptr[q] + ktn * 16 * KS16);
#pragma unroll
        for (int q = 0; q < NW; ++q) rw[q] = w_ptr[q][(int64_t)ktn * KS16 * 192];
        if (wave_active) {
            const f32x4* Ac = As + cur * A_STAGE + (2 * wm) * 64 + lane;
            const i32x4* Wc = Ws + cur * W_STAGE + (2 * wn) * 64 + lane;
#pragma unroll
            for (int s16 = 0; s16 < KS16; ++s16) {
                i32x4 a[2][3], w[2][3];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 lo = Ac[((s16 * 2 + 0) * 4 + t) * 64], hi = Ac[((s16 * 2 + 1) * 4 + t) * 64];
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    split8(v, a[t][0], a[t][1], a[t][2]);
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) w[t][pc] = Wc[s16 * (3 * 4 * 64) + (pc * 4 + t) * 64];
                }
                // term-major: consecutive MFMAs go to four different accumulators (no back-to-back dependence);
                // per accumulator the order is still smallest partial product first
#define MELLOW_TERM(PW, PA)                                   \
                MELLOW_BF(w[0][PW], a[0][PA], acc[0][0])      \
                MELLOW_BF(w[0][PW], a[1][PA], acc[0][1])      \
                MELLOW_BF(w[1][PW], a[0][PA], acc[1][0])      \
                MELLOW_BF(w[1][PW], a[1][PA], acc[1][1])
                MELLOW_TERM(2, 0)
                MELLOW_TERM(0, 2)
                MELLOW_TERM(1, 1)
                MELLOW_TERM(1, 0)
                MELLOW_TERM(0, 1)
                MELLOW_TERM(0, 0)
#undef MELLOW_TERM
            }
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) As[(cur ^ 1) * A_STAGE + a_lds[q]] = ra[q];
#pragma unroll
        for (int q = 0; q < NW; ++q) Ws[(cur ^ 1) * W_STAGE + w_lds[q]] = rw[q];
        __syncthreads();
    }
    gemm_epilogue<WN, EPI>(g, acc, pm, pn, wm, wn, lane, BM, BN);
}

// ---- software-pipelined variant of the fused kernel ("x3p") ---------------------------------------------------------------
// What the counters of the kernel above say (rocprofv3 --pmc on the K = 4608 steady state, profiles/r02_pmc_x3_loop.txt): the
// matrix pipe is busy 39 % of the time, the VALU (operand split) 25 %, and the two co-execute in only 4 % of the cycles: a wave
// reads its fragments, splits them, issues its 24 MFMAs and then stores / waits at the barrier, so with ~2 waves per SIMD the
// pipe idles whenever both are outside their MFMA burst.  Here the bursts are made continuous inside ONE wave:
//   * three LDS stages (k16 each): tile t+2 is being written while tile t is multiplied, so tile t+1 is already complete and
//     its fragments are read + split DURING the MFMAs of tile t (fragments double-buffered in registers);
//   * sched_group_barrier interleaves those LDS reads / VALU ops / global loads between the MFMA issues;
//   * still one barrier per k16, but nothing waits on it: after the barrier the next tile's fragments are in registers;
//   * A is split ONCE per workgroup when it is staged into LDS (the fused kernel split it in every consuming wave);
//   * a tile's global loads are issued two whole iterations before its LDS store.
// Measured (tools/f32x3_bench.py, random data): LM shapes 131/119/131/128 -> 148/144/160/167 TFLOP/s-equivalent, K = 4608
// steady state 167 -> 189.  Also measured and NOT kept: loading the weight fragments straight from global memory in fragment
// order (no LDS for W, half the LDS operations): 188.7 vs 189.2 -- LDS issue is not what bounds the loop; a free (wrong)
// operand split: 206 -- the VALU is not either.  What remains is the one barrier per 24 MFMAs with two waves per SIMD.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_x3p_kernel(const GemmBDev p) {
    constexpr int BM = 128, BN = 128, WN = 2, NST = 3;
    constexpr int STAGE = 3 * 4 * 64;                                       // 16-byte slots per operand per stage: [piece][tile][lane]
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_bf[];
    i32x4* As = smem_bf;                                                    // [NST][STAGE]  A already split (3 bf16 pieces)
    i32x4* Ws = smem_bf + NST * STAGE;                                      // [NST][STAGE]
    const GemmArgs& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = p.gm * p.gn;
    const int L = xcd_remap((int)blockIdx.x, tiles * p.ks);
    const int tile = L % tiles, split = L / tiles;
    const int pm = tile / p.gn, pn = tile % p.gn;
    const int KTf = g.K >> 4;
    const int kt0 = split * KTf / p.ks;
    const int KT = (split + 1) * KTf / p.ks - kt0;                          // this workgroup's k16 steps (all of them without split-K)
    const i32x4* PB = reinterpret_cast<const i32x4*>(g.W8);

    // A staging: thread t owns the 8 consecutive k of (row t/2, k-half t%2) of a k16 tile = exactly one lane's share of an MFMA
    // operand: it loads 32 contiguous bytes, splits them ONCE for the whole workgroup and writes the three bf16 pieces
    const int arow = tid >> 1, akh = tid & 1;
    int am = pm * BM + arow;
    am = am < g.M ? am : g.M - 1;
    // A_FRAMES (the STFT): row m = frame (m % fpc) of clip (m / fpc), a 1024-sample window starting at hop * frame
    const int64_t a_off = g.a_mode == A_FRAMES ? (int64_t)(am / g.fpc) * g.clip_stride + (int64_t)(am % g.fpc) * g.hop : (int64_t)am * g.lda;
    const float* a_ptr = g.A + a_off + akh * 8 + kt0 * 16;
    const int a_lds = (arow >> 5) * 64 + (arow & 31) + 32 * akh;            // + piece * 256
    const i32x4* w_ptr[3];
    int w_lds[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = q * 256 + tid;
        const int ntl = c / 192, r2 = c % 192;           // W: [n-tile][piece][lane] per k16
        w_ptr[q] = PB + ((int64_t)(pn * 4 + ntl) * KTf + kt0) * 192 + r2;
        w_lds[q] = ((r2 >> 6) * 4 + ntl) * 64 + (r2 & 63);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: a tile's global loads are issued two whole iterations before its LDS store (an L2 round trip is
    // longer than one 24-MFMA iteration)
    f32x4 raA[2], raB[2];
    i32x4 rwA[3], rwB[3];
#define X3_GLOAD(T, ra, rw)                                                                      \
    {                                                                                            \
        const int t_ = (T) < KT ? (T) : KT - 1;                                                  \
        ra[0] = *reinterpret_cast<const f32x4*>(a_ptr + t_ * 16);                                \
        ra[1] = *reinterpret_cast<const f32x4*>(a_ptr + t_ * 16 + 4);                            \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) rw[q] = w_ptr[q][(int64_t)t_ * 192];       \
    }
#define X3_LSTORE(ST, ra, rw)                                                                    \
    {                                                                                            \
        const float v_[8] = {ra[0].x, ra[0].y, ra[0].z, ra[0].w, ra[1].x, ra[1].y, ra[1].z, ra[1].w}; \
        i32x4 p0_, p1_, p2_;                                                                     \
        split8(v_, p0_, p1_, p2_);                                                               \
        As[(ST) * STAGE + a_lds] = p0_;                                                          \
        As[(ST) * STAGE + 256 + a_lds] = p1_;                                                    \
        As[(ST) * STAGE + 512 + a_lds] = p2_;                                                    \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) Ws[(ST) * STAGE + w_lds[q]] = rw[q];        \
    }
    // fragments of one k16 tile: a[m-tile][piece], w[n-tile][piece]
#define X3_FRAGS(ST, FA, FW)                                                                     \
    {                                                                                            \
        const i32x4* Ac = As + (ST) * STAGE + (2 * wm) * 64 + lane;                              \
        const i32x4* Wc = Ws + (ST) * STAGE + (2 * wn) * 64 + lane;                              \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                            \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                   \
                FA[t][pc] = Ac[(pc * 4 + t) * 64];                                               \
                FW[t][pc] = Wc[(pc * 4 + t) * 64];                                               \
            }                                                                                    \
    }
#define X3_TERM(FA, FW, PW, PA)                         \
    MELLOW_BF(FW[0][PW], FA[0][PA], acc[0][0])          \
    MELLOW_BF(FW[0][PW], FA[1][PA], acc[0][1])          \
    MELLOW_BF(FW[1][PW], FA[0][PA], acc[1][0])          \
    MELLOW_BF(FW[1][PW], FA[1][PA], acc[1][1])
#define X3_MFMAS(FA, FW) X3_TERM(FA, FW, 2, 0) X3_TERM(FA, FW, 0, 2) X3_TERM(FA, FW, 1, 1) X3_TERM(FA, FW, 1, 0) X3_TERM(FA, FW, 0, 1) X3_TERM(FA, FW, 0, 0)
    // interleave between the MFMA issues: the 12 fragment reads of the next tile first, then the split of the staged tile
    // (VALU) with its 6 LDS writes, then the 5 global loads of the tile after that
#define X3_SCHED()                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) {                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        if (i_ < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
        if (i_ >= 2 && i_ < 14) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);               \
        if (i_ >= 14 && i_ < 20) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);              \
        if (i_ >= 19 && i_ < 24) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              \
    }

    // the wave raises its priority for the MFMA burst so that the SIMD's other wave does not steal issue slots in the middle
    // of it (LM shapes +1..5 %, tools/ab_build.sh -DMELLOW_X3_NO_SETPRIO for the A/B)
#ifdef MELLOW_X3_NO_SETPRIO
#define X3_PRIO(P)
#else
#define X3_PRIO(P) __builtin_amdgcn_s_setprio(P);
#endif
    i32x4 fa0[2][3], fw0[2][3], fa1[2][3], fw1[2][3];
    X3_GLOAD(0, raA, rwA)
    X3_GLOAD(1, raB, rwB)
    X3_LSTORE(0, raA, rwA)
    X3_LSTORE(1, raB, rwB)
    X3_GLOAD(2, raA, rwA)
    X3_GLOAD(3, raB, rwB)
    __syncthreads();
    X3_FRAGS(0, fa0, fw0)
    // iteration t: fragments of tile t+1 (complete since the last barrier) -> registers; MFMAs of tile t; registers (tile t+2,
    // loaded one whole iteration ago) split -> stage (t+2) % 3, whose last readers finished in iteration t-2; loads of tile
    // t+4 -> the same registers (two sets alternate); barrier.  Unrolled by 6: stage indices and register sets are static.
    // MELLOW_X3_ABL (developer ablation, wrong results): bit 0 no global loads, 1 no barrier, 2 no LDS stores, 3 no fragment
    // reads in the loop -- what each part costs on top of the 24 MFMAs (tools/README.md)
#ifndef MELLOW_X3_ABL
#define MELLOW_X3_ABL 0
#endif
#define X3_ITER(T, SN, SW, FA, FW, FAN, FWN, ra, rw)                                             \
    if ((T) < KT) {                                                                              \
        X3_PRIO(2)                                                                               \
        if (!(MELLOW_X3_ABL & 8)) X3_FRAGS(SN, FAN, FWN)                                         \
        X3_MFMAS(FA, FW)                                                                         \
        if (!(MELLOW_X3_ABL & 4)) X3_LSTORE(SW, ra, rw)                                          \
        if (!(MELLOW_X3_ABL & 1)) X3_GLOAD((T) + 4, ra, rw)                                      \
        if (!MELLOW_X3_ABL) X3_SCHED()                                                           \
        X3_PRIO(0)                                                                               \
        if (!(MELLOW_X3_ABL & 2)) __syncthreads();                                               \
    }
    for (int kt = 0; kt < KT; kt += 6) {
        X3_ITER(kt + 0, 1, 2, fa0, fw0, fa1, fw1, raA, rwA)
        X3_ITER(kt + 1, 2, 0, fa1, fw1, fa0, fw0, raB, rwB)
        X3_ITER(kt + 2, 0, 1, fa0, fw0, fa1, fw1, raA, rwA)
        X3_ITER(kt + 3, 1, 2, fa1, fw1, fa0, fw0, raB, rwB)
        X3_ITER(kt + 4, 2, 0, fa0, fw0, fa1, fw1, raA, rwA)
        X3_ITER(kt + 5, 0, 1, fa1, fw1, fa0, fw0, raB, rwB)
    }
#undef X3_GLOAD
#undef X3_LSTORE
#undef X3_FRAGS
#undef X3_TERM
#undef X3_MFMAS
#undef X3_SCHED
#undef X3_ITER
#undef X3_PRIO
    if (p.ks > 1) return splitk_store(p, acc, tile, split, tid);
    gemm_epilogue<WN, EPI>(g, acc, pm, pn, wm, wn, lane, BM, BN);
}

// ---- "x3q": both operands arrive pre-split in fragment order and go global -> LDS by LDS-DMA ------------------------------
// The ablations of the kernel above (MELLOW_X3_ABL, K = 4608: all parts 184 TF-eq; without the LDS stores + split 249; without
// the global loads 228; MFMAs alone 320) say the register staging is what the loop pays for.  Here A is split by its PRODUCER
// into the same [piece][32-row tile][lane] 16-byte order the weight already has ("APB": per 128-row panel and k16 tile 12 KiB
// contiguous), so a stage of either operand is twelve 1-KiB `global_load_lds_dwordx4` (wave-uniform LDS base + lane * 16): no
// staging registers, no VALU, no ds_write.  Three stages: during iteration t the fragments of tile t+1 are read, tile t+2 is
// landing and tile t+3 is issued into the stage tile t occupied (its fragments have been in registers since iteration t-1).
// The LDS-DMA loads are issued from inline asm, which hipcc does not count: the waits are explicit (vmcnt(6): the six loads of
// tile t+3 may stay in flight across the barrier).
__device__ __forceinline__ void glds16(const void* base, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(base) : "memory");
}
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_x3q_kernel(const GemmBDev p) {
    constexpr int BM = 128, BN = 128, WN = 2, NST = 3;
    constexpr int STAGE = 3 * 4 * 64;                                       // i32x4 per operand per stage
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_bf[];
    i32x4* As = smem_bf;
    i32x4* Ws = smem_bf + NST * STAGE;
    const GemmArgs& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = p.gm * p.gn;
    int tile, split, pm, pn;
    if (p.ng) {
        // Weight-stationary per XCD (block b runs on XCD b % 8: observed, speed only).  With the panel-major order below, an XCD
        // walks ~12 row panels and needs ALL gn weight tiles for each: the LM's gate/up operand is 10.6 MB against a 4 MiB L2, so
        // every panel re-fetched the whole weight -- measured 850 MB of reads per launch for 54 MB of operands, 45 % of the
        // family's memory-side traffic, and the launch ran at the fabric's rate (profiles/r05_pmc_traffic_by_shape_*.txt).  Here an
        // XCD keeps gn / 8 weight tiles (1.3 MB) hot and every panel is read once per XCD: 8 x 43 + 10.6 MB.
        const int npx = p.gn >> 3, b = (int)blockIdx.x, j = b >> 3;
        pm = j / npx; pn = (b & 7) * npx + j % npx;
        tile = pm * p.gn + pn; split = 0;
    } else {
        const int L = xcd_remap((int)blockIdx.x, tiles * p.ks);
        tile = L % tiles; split = L / tiles;
        pm = tile / p.gn; pn = tile % p.gn;
    }
    const int KTf = g.K >> 4;
    const int kt0 = split * KTf / p.ks;
    const int KT = (split + 1) * KTf / p.ks - kt0;                          // this workgroup's k16 steps (split-K, see splitk_store)
    // waves 0,1 carry the A stage (chunks wave*6 .. +5 of 12), waves 2,3 the W stage
    // norm-free chaining (kernels.h, rs_*): the row scales of this workgroup's 128 rows are formed here, while the first
    // LDS-DMA stages are in flight, so that the epilogue finds them in LDS instead of starting with nine dependent loads per row
    __shared__ float rs_rows[BM];
    if (g.rs_ssq && tid < BM) {
        int64_t m = (int64_t)pm * BM + tid;
        m = m < g.M ? m : g.M - 1;
        const float* sp = g.rs_ssq + m * g.rs_parts;
        float ss = 0.f;
        for (int p = 0; p < g.rs_parts; ++p) ss += sp[p];             // fixed order
        rs_rows[tid] = 1.0f / sqrtf(ss / g.rs_dim + g.rs_eps);
    }
    const bool isA = wave < 2;
    const int c0 = (wave & 1) * 6;
    const char* base = isA ? reinterpret_cast<const char*>(g.A8) + (((int64_t)pm * KTf + kt0) * 12 + c0) * 1024
                           : reinterpret_cast<const char*>(g.W8) + ((int64_t)pn * 4 * KTf + kt0) * 3072;
    const uint32_t kstep = isA ? 12288u : 3072u;
    uint32_t voff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = c0 + j;                                               // W chunk c = piece * 4 + n-tile
        voff[j] = isA ? (uint32_t)(j * 1024 + lane * 16) : (uint32_t)(((c & 3) * KTf * 192 + (c >> 2) * 64 + lane) * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_bf + (isA ? 0u : (uint32_t)(NST * STAGE * 16)) + (uint32_t)c0 * 1024u;
#define X3Q_ISSUE(T, ST)                                                                         \
    {                                                                                            \
        const int t_ = (T) < KT ? (T) : KT - 1;                                                  \
        const char* b_ = base + (int64_t)t_ * kstep;                                             \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) glds16(b_, voff[j], lds0 + (uint32_t)((ST) * STAGE * 16 + j * 1024)); \
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define X3Q_FRAGS(ST, FA, FW)                                                                    \
    {                                                                                            \
        const i32x4* Ac = As + (ST) * STAGE + (2 * wm) * 64 + lane;                              \
        const i32x4* Wc = Ws + (ST) * STAGE + (2 * wn) * 64 + lane;                              \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                            \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                   \
                FA[t][pc] = Ac[(pc * 4 + t) * 64];                                               \
                FW[t][pc] = Wc[(pc * 4 + t) * 64];                                               \
            }                                                                                    \
    }
#define X3Q_TERM(FA, FW, PW, PA)                        \
    MELLOW_BF(FW[0][PW], FA[0][PA], acc[0][0])          \
    MELLOW_BF(FW[0][PW], FA[1][PA], acc[0][1])          \
    MELLOW_BF(FW[1][PW], FA[0][PA], acc[1][0])          \
    MELLOW_BF(FW[1][PW], FA[1][PA], acc[1][1])
#define X3Q_MFMAS(FA, FW) X3Q_TERM(FA, FW, 2, 0) X3Q_TERM(FA, FW, 0, 2) X3Q_TERM(FA, FW, 1, 1) X3Q_TERM(FA, FW, 1, 0) X3Q_TERM(FA, FW, 0, 1) X3Q_TERM(FA, FW, 0, 0)
#define X3Q_SCHED()                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) {                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        if (i_ < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
    }
    i32x4 fa0[2][3], fw0[2][3], fa1[2][3], fw1[2][3];
    X3Q_ISSUE(0, 0)
    X3Q_ISSUE(1, 1)
    X3Q_ISSUE(2, 2)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                        // tiles 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    X3Q_FRAGS(0, fa0, fw0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                           // stage 0 is free for tile 3
#ifndef MELLOW_X3Q_VAR
#define MELLOW_X3Q_VAR 8
#endif
    // developer variants (tools/ab_build.sh -DMELLOW_X3Q_VAR=bits): 1 no LDS-DMA in the loop (wrong), 2 no vmcnt wait (wrong),
    // 4 no setprio, 8 LDS-DMA issued between the last 12 MFMAs instead of before the first
#define X3Q_WAIT() if (MELLOW_X3Q_VAR & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
#define X3Q_ITER(T, S0, S1, FA, FW, FAN, FWN)                                                    \
    if ((T) < KT) {                                                                              \
        if (!(MELLOW_X3Q_VAR & 9)) X3Q_ISSUE((T) + 3, S0)                                        \
        if (!(MELLOW_X3Q_VAR & 4)) __builtin_amdgcn_s_setprio(2);                                \
        X3Q_FRAGS(S1, FAN, FWN)                                                                  \
        if (MELLOW_X3Q_VAR & 8) {                                                                \
            X3Q_TERM(FA, FW, 2, 0) X3Q_TERM(FA, FW, 0, 2) X3Q_TERM(FA, FW, 1, 1)                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                  \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            const int t_ = (T) + 3 < KT ? (T) + 3 : KT - 1;                                      \
            const char* b_ = base + (int64_t)t_ * kstep;                                         \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                      \
                glds16(b_, voff[j], lds0 + (uint32_t)((S0) * STAGE * 16 + j * 1024));            \
                if (j == 0) { MELLOW_BF(FW[0][1], FA[0][0], acc[0][0]) MELLOW_BF(FW[0][1], FA[1][0], acc[0][1]) } \
                if (j == 1) { MELLOW_BF(FW[1][1], FA[0][0], acc[1][0]) MELLOW_BF(FW[1][1], FA[1][0], acc[1][1]) } \
                if (j == 2) { MELLOW_BF(FW[0][0], FA[0][1], acc[0][0]) MELLOW_BF(FW[0][0], FA[1][1], acc[0][1]) } \
                if (j == 3) { MELLOW_BF(FW[1][0], FA[0][1], acc[1][0]) MELLOW_BF(FW[1][0], FA[1][1], acc[1][1]) } \
                if (j == 4) { MELLOW_BF(FW[0][0], FA[0][0], acc[0][0]) MELLOW_BF(FW[0][0], FA[1][0], acc[0][1]) } \
                if (j == 5) { MELLOW_BF(FW[1][0], FA[0][0], acc[1][0]) MELLOW_BF(FW[1][0], FA[1][0], acc[1][1]) } \
                __builtin_amdgcn_sched_barrier(0);                                               \
            }                                                                                    \
        } else {                                                                                 \
            X3Q_MFMAS(FA, FW)                                                                    \
            X3Q_SCHED()                                                                          \
        }                                                                                        \
        if (!(MELLOW_X3Q_VAR & 4)) __builtin_amdgcn_s_setprio(0);                                \
        X3Q_WAIT()                                                                               \
        __builtin_amdgcn_s_barrier();                                                            \
    }
    // A wave whose 64 columns lie entirely beyond the weight's rows (the second half of the last column tile when N is an odd
    // multiple of 64: N = 576 -> 1 wave pair in 10, N = 960 -> 1 in 16) keeps its loading duty and its barriers but issues no
    // MFMA: the matrix pipe it would have kept busy with zeros goes to the other workgroup of the CU.
    const bool idle = pn * BN + wn * 64 >= g.Nw;            // wave-uniform
#define X3Q_ITER_IDLE(T, S0)                                                                     \
    if ((T) < KT) {                                                                              \
        X3Q_ISSUE((T) + 3, S0)                                                                   \
        X3Q_WAIT()                                                                               \
        __builtin_amdgcn_s_barrier();                                                            \
    }
    if (!idle) {
        for (int kt = 0; kt < KT; kt += 6) {
            X3Q_ITER(kt + 0, 0, 1, fa0, fw0, fa1, fw1)
            X3Q_ITER(kt + 1, 1, 2, fa1, fw1, fa0, fw0)
            X3Q_ITER(kt + 2, 2, 0, fa0, fw0, fa1, fw1)
            X3Q_ITER(kt + 3, 0, 1, fa1, fw1, fa0, fw0)
            X3Q_ITER(kt + 4, 1, 2, fa0, fw0, fa1, fw1)
            X3Q_ITER(kt + 5, 2, 0, fa1, fw1, fa0, fw0)
        }
    } else {
        for (int kt = 0; kt < KT; kt += 3) {
            X3Q_ITER_IDLE(kt + 0, 0)
            X3Q_ITER_IDLE(kt + 1, 1)
            X3Q_ITER_IDLE(kt + 2, 2)
        }
    }
#undef X3Q_ITER_IDLE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the clamped re-loads of the last tiles
#undef X3Q_ISSUE
#undef X3Q_FRAGS
#undef X3Q_TERM
#undef X3Q_MFMAS
#undef X3Q_SCHED
#undef X3Q_ITER
#undef X3Q_WAIT
    if (p.ks > 1) return splitk_store(p, acc, tile, split, tid);
    gemm_epilogue<WN, EPI>(g, acc, pm, pn, wm, wn, lane, BM, BN, g.rs_ssq ? rs_rows : nullptr);   // (written before the main loop's barriers)
}

// ---- "x3w": the K = 96 GEMMs of Swin stage 0 (M = 4096 rows per clip), weight-stationary and persistent ---------------------
// A 128 x 128 tile of these GEMMs is six k16 steps: 5 us of matrix work behind a prologue (first stages from memory) and in front
// of an epilogue of 64 KB of stores, and the chip ran them at 17 us per pair of workgroups whichever kernel did it (x3p, x3q, the
// fused K = 96 kernel).  Here a workgroup keeps ITS 64 weight columns for its whole life -- every wave holds the fragments of its
// 32 columns for all six k16 steps in 72 VGPRs (keeping the tile in LDS instead measured the same: LDS bandwidth is not the
// bound) -- and walks over the row panels (128 rows each) of the pre-split activation (APB) through a SIX-stage LDS-DMA ring, a
// whole panel ahead: the ring never drains, while panel p runs, panel p + G is landing.  Each panel's 128 x 64 outputs leave
// from its epilogue (bias from registers, optional exact-erf GELU, fp32 stores; the stores are asynchronous, the next panel's
// MFMAs start behind them).  Four waves, wave tile 64 x 32, 72 KB of LDS: two workgroups per CU, which drift apart, so one's
// epilogue runs under the other's MFMAs.  The same MFMAs in the same order per output element as the other x3 kernels.
// vmcnt: the LDS-DMA loads of the four newest stages (3 per wave and stage) may stay in flight (loads complete in order among
// loads; pending stores can only make the wait longer).
#ifndef MELLOW_X3W_ABL
#define MELLOW_X3W_ABL 0       // developer ablation (tools/ab_build.sh -DMELLOW_X3W_ABL=bits): 1 no output stores, 2 no MFMAs, 4 output stored column-tile-major (wrong results, timing only)
#endif
template <bool GELU>
__global__ __launch_bounds__(256, 2) void gemm_x3w_kernel(const GemmBDev p) {
    constexpr int KT = 6, NST = 6;                                          // element (panel, kt) of the A stream lives in stage kt
    constexpr int ASTAGE = 3 * 4 * 64;                                      // i32x4 per A stage: [piece][32-row tile][lane]
    extern __shared__ __attribute__((aligned(16))) i32x4 smem_bf[];
    i32x4* As = smem_bf;
    const GemmArgs& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = p.gm, gn = p.gn;                                          // workgroups per column tile; 64-column tiles
    const int L = xcd_remap((int)blockIdx.x, G * gn);
    const int g0 = L / gn, pn = L % gn;                                     // the gn workgroups of a group walk the same panels (one L2)
    const int panels = (g.M + 127) >> 7;
    if (g0 >= panels) return;
    const int n_it = (panels - g0 + G - 1) / G;
    const uint32_t lds_a = (uint32_t)(uintptr_t)smem_bf;
    // the weight fragments of this wave's n-tile (2 pn + wn), once: PB order [n-tile][k16][piece][lane]
    i32x4 wf[KT][3];
    {
        const i32x4* wp = reinterpret_cast<const i32x4*>(g.W8) + ((int64_t)(2 * pn + wn) * KT) * 192 + lane;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) wf[kt][pc] = wp[(kt * 3 + pc) * 64];
    }
    // A ring: wave w carries chunks 3 w .. 3 w + 2 of the 12 of a stage
    const char* ab = reinterpret_cast<const char*>(g.A8);
    const uint32_t a_voff = (uint32_t)(wave * 3 * 1024 + lane * 16);
#define X3W_ISSUE(IT, KTI)                                                                       \
    {                                                                                            \
        const int it_ = (IT) < n_it ? (IT) : n_it - 1;                                           \
        const char* b_ = ab + ((int64_t)(g0 + it_ * G) * KT + (KTI)) * 12288;                    \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                            \
            glds16(b_, a_voff + (uint32_t)(j * 1024), lds_a + (uint32_t)(((KTI) * ASTAGE) * 16 + (wave * 3 + j) * 1024)); \
    }
    // bias of this lane's columns (32 wn + 8 gq + 4 h .. + 3), kept in registers for every panel
    const int h = lane >> 5;
    const int col0 = pn * 64 + wn * 32 + 4 * h;
    float bias[4][4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int col = col0 + 8 * gq;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias && col < g.N) b4 = *reinterpret_cast<const f32x4*>(g.bias + col);
        bias[gq][0] = b4.x; bias[gq][1] = b4.y; bias[gq][2] = b4.z; bias[gq][3] = b4.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the bias and weight-fragment loads are in
    X3W_ISSUE(0, 0)
    X3W_ISSUE(0, 1)
    X3W_ISSUE(0, 2)
    X3W_ISSUE(0, 3)
    X3W_ISSUE(0, 4)
    X3W_ISSUE(0, 5)
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#define X3W_FRAGS(KTI, FA)                                                                       \
    {                                                                                            \
        const i32x4* Ac = As + (KTI) * ASTAGE + (2 * wm) * 64 + lane;                            \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                       \
            FA[0][pc] = Ac[(pc * 4) * 64];                                                       \
            FA[1][pc] = Ac[(pc * 4 + 1) * 64];                                                   \
        }                                                                                        \
    }
#define X3W_TERM(FA, FW, PW, PA) MELLOW_BF(FW[PW], FA[0][PA], acc[0]) MELLOW_BF(FW[PW], FA[1][PA], acc[1])
#define X3W_MFMAS(FA, FW) X3W_TERM(FA, FW, 2, 0) X3W_TERM(FA, FW, 0, 2) X3W_TERM(FA, FW, 1, 1) X3W_TERM(FA, FW, 1, 0) X3W_TERM(FA, FW, 0, 1) X3W_TERM(FA, FW, 0, 0)
    // a STEP is two k16 tiles (24 MFMAs per barrier: with one tile per barrier the kernel ran at 146 / 199 us, the barrier and the
    // fragment-read latency of a 12-MFMA step being as long as its MFMAs); fragment sets are double-buffered per step
    i32x4 fa0[2][2][3], fa1[2][2][3];                                       // [k16 tile of the step][m-tile][piece]
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                        // steps 0 and 1 (stages 0..3) have landed
    __builtin_amdgcn_s_barrier();
    X3W_FRAGS(0, fa0[0])
    X3W_FRAGS(1, fa0[1])
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                           // stages 0, 1 are free for the next panel's first step
    // iteration (it, step): fragments of the next step -> registers, the 24 MFMAs of this one, the LDS-DMA of the same step of the
    // NEXT panel into the two stages this one occupied, wait (all but the six loads just issued), barrier
#define X3W_ITER(ST, FA, FAN)                                                                    \
    {                                                                                            \
        __builtin_amdgcn_s_setprio(2);                                                           \
        X3W_FRAGS((2 * (ST) + 2) % KT, FAN[0])                                                   \
        X3W_FRAGS((2 * (ST) + 3) % KT, FAN[1])                                                   \
        if (!(MELLOW_X3W_ABL & 2)) {                                                             \
            X3W_MFMAS(FA[0], wf[2 * (ST)])                                                       \
            X3W_MFMAS(FA[1], wf[2 * (ST) + 1])                                                   \
        } else {                                                                                 \
            acc[0][0] += __int_as_float(FA[0][0][0][0] ^ FA[1][1][2][3]);                        \
        }                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) {                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            if (i_ < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        X3W_ISSUE(it + 1, 2 * (ST))                                                              \
        X3W_ISSUE(it + 1, 2 * (ST) + 1)                                                          \
        __builtin_amdgcn_s_setprio(0);                                                           \
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                              \
        __builtin_amdgcn_s_barrier();                                                            \
    }
    // epilogue of panel g0 + it * G: lane owns row (lane % 32) of each of its two 32-row tiles
#define X3W_EPILOGUE()                                                                           \
    {                                                                                            \
        const int64_t prow = (int64_t)(g0 + it * G) * 128 + wm * 64 + (lane & 31);               \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                       \
            const int64_t m = prow + mi * 32;                                                    \
            if (m < g.M) {                                                                       \
                _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                               \
                    const int col = col0 + 8 * gq;                                               \
                    if (col < g.N) {                                                             \
                        float v[4];                                                              \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                          \
                            v[j] = acc[mi][4 * gq + j] + bias[gq][j];                            \
                            if (GELU) v[j] = gelu_erf(v[j]);                                     \
                        }                                                                        \
                        if (MELLOW_X3W_ABL & 4) *reinterpret_cast<f32x4*>(g.C + ((int64_t)(pn % (g.N >> 6)) * g.M + m) * 64 + (col & 63)) = f32x4{v[0], v[1], v[2], v[3]}; \
                        else if (!(MELLOW_X3W_ABL & 1) || v[0] == 1.2345f) *reinterpret_cast<f32x4*>(g.C + m * g.ldc + col) = f32x4{v[0], v[1], v[2], v[3]}; \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;                     \
        }                                                                                        \
    }
    // three steps per panel: the two fragment sets swap roles from one panel to the next
#define X3W_PANEL(A0, A1) X3W_ITER(0, A0, A1) X3W_ITER(1, A1, A0) X3W_ITER(2, A0, A1) X3W_EPILOGUE()
    for (int it = 0; it < n_it;) {
        X3W_PANEL(fa0, fa1)
        if (++it >= n_it) break;
        X3W_PANEL(fa1, fa0)
        ++it;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the clamped re-loads behind the last panel
#undef X3W_ISSUE
#undef X3W_FRAGS
#undef X3W_TERM
#undef X3W_MFMAS
#undef X3W_ITER
#undef X3W_EPILOGUE
#undef X3W_PANEL
}

#undef MELLOW_BF

// A fp32 [M][lda] -> APB: i32x4 index ((panel * K/16 + kt) * 12 + piece * 4 + tile) * 64 + lane, row = panel * 128 + tile * 32
// + lane % 32, k = kt * 16 + (lane / 32) * 8 .. + 7; rows >= M are written as zeros (the buffer holds roundup(M, 128) rows)
__global__ __launch_bounds__(256) void split_rows_apb_kernel(const float* __restrict__ A, int64_t lda, int M, int K, i32x4* __restrict__ out) {
    const int KT = K >> 4;
    const int64_t total = (int64_t)((M + 127) / 128) * KT * 256;            // one thread per (panel, kt, tile, lane)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), tile = (int)((i >> 6) & 3);
        const int64_t pk = i >> 8;
        const int kt = (int)(pk % KT);
        const int64_t panel = pk / KT;
        const int64_t m = panel * 128 + tile * 32 + (lane & 31);
        const int k = kt * 16 + (lane >> 5) * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < M) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(A + m * lda + k), y = *reinterpret_cast<const f32x4*>(A + m * lda + k + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
        }
        i32x4 p0, p1, p2;
        split8(v, p0, p1, p2);
        i32x4* o = out + (pk * 12 + tile) * 64 + lane;
        o[0] = p0; o[4 * 64] = p1; o[8 * 64] = p2;
    }
}
void launch_split_rows_apb(const float* A, int64_t lda, int M, int K, void* out, hipStream_t s) {
    const int64_t total = (int64_t)((M + 127) / 128) * (K >> 4) * 256;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_rows_apb_kernel, dim3(blocks), dim3(256), 0, s, A, lda, M, K, reinterpret_cast<i32x4*>(out));
}
template <int EPI>
static void launchq(const GemmArgs& a, hipStream_t s) {
    GemmBDev d;
    d.a = a;
    d.gm = (a.M + 127) / 128;
    d.gn = (a.Nw + 127) / 128;
    d.ks = splitk_for(a, d.gm * d.gn, a.K >> 4);
    d.ng = (d.ks == 1 && d.gn % 8 == 0 && d.gn >= 16) ? 1 : 0;
    const size_t lds = (size_t)3 * (2 * 3 * 4 * 64) * 16;                              // 72 KiB
    set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_x3q_kernel<EPI>), lds);
    hipLaunchKernelGGL((gemm_x3q_kernel<EPI>), dim3(d.gm * d.gn * d.ks), dim3(256), lds, s, d);
    if (d.ks > 1) hipLaunchKernelGGL((splitk_finish_kernel<EPI>), dim3(d.gm * d.gn), dim3(256), 0, s, d);
}
// K = 96, plain LINEAR epilogue (bias, optional GELU, fp32 output): the weight-stationary persistent kernel
static bool x3w_fits(const GemmArgs& a) {
    return !a.no_x3w && a.epi == EPI_LINEAR && a.K == 96 && a.M >= 8192 && a.C && !a.C3 && !a.resid && !a.crow_map && !a.rs_ssq &&
           (a.act == ACT_NONE || a.act == ACT_GELU) && a.N % 4 == 0;
}
static void launchw(const GemmArgs& a, hipStream_t s) {
    GemmBDev d;
    d.a = a;
    d.gn = (a.Nw + 63) / 64;
    const int panels = (a.M + 127) / 128;
    d.gm = 512 / d.gn < panels ? 512 / d.gn : panels;                      // workgroups per column tile (two per CU in all)
    d.ks = 1;
    const size_t lds = (size_t)(6 * (3 * 4 * 64)) * 16;                                 // 72 KiB
    set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_x3w_kernel<false>), lds);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_x3w_kernel<true>), lds);
    if (a.act == ACT_GELU) hipLaunchKernelGGL((gemm_x3w_kernel<true>), dim3(d.gm * d.gn), dim3(256), lds, s, d);
    else hipLaunchKernelGGL((gemm_x3w_kernel<false>), dim3(d.gm * d.gn), dim3(256), lds, s, d);
}
// g.A8 = APB (launch_split_rows_apb or a producer's split output), g.W8 = PB; K % 16 == 0, K >= 48
void launch_gemm_bf16x3_apb(const GemmArgs& a, hipStream_t s) {
    if (x3w_fits(a)) return launchw(a, s);
    switch (a.epi) {
        case EPI_LINEAR: launchq<EPI_LINEAR>(a, s); break;
        case EPI_SWIGLU: launchq<EPI_SWIGLU>(a, s); break;
        case EPI_QKV_ROPE: launchq<EPI_QKV_ROPE>(a, s); break;
        default: break;
    }
}

template <int EPI>
static void launchbf(const GemmArgs& a, hipStream_t s) {
    GemmBDev d;
    d.a = a;
    d.gm = (a.M + 127) / 128;
    d.gn = (a.Nw + 127) / 128;
    d.ks = 1;
    if (a.K % 16 == 0 && a.K >= 192) {      // shorter K: the 3-stage prologue costs more than it hides (K = 96: 77 vs 84 TF)
        const size_t lds = (size_t)3 * (2 * 3 * 4 * 64) * 16;                          // 72 KiB
        set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_x3p_kernel<EPI>), lds);
        d.ks = splitk_for(a, d.gm * d.gn, a.K >> 4);
        hipLaunchKernelGGL((gemm_x3p_kernel<EPI>), dim3(d.gm * d.gn * d.ks), dim3(256), lds, s, d);
        if (d.ks > 1) hipLaunchKernelGGL((splitk_finish_kernel<EPI>), dim3(d.gm * d.gn), dim3(256), 0, s, d);
        return;
    }
    const size_t lds = (size_t)2 * (2 * 4 * 64 + 3 * 4 * 64) * 16;                 // 40 KiB
    hipLaunchKernelGGL((gemm_bf16x3f_kernel<EPI, 1>), dim3(d.gm * d.gn), dim3(256), lds, s, d);
}
// fused variant: g.A (fp32, row-major, lda) + g.W8 (PB); K % 32 == 0; six partial products
void launch_gemm_bf16x3_fused(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_LINEAR: launchbf<EPI_LINEAR>(a, s); break;
        case EPI_SWIGLU: launchbf<EPI_SWIGLU>(a, s); break;
        case EPI_QKV_ROPE: launchbf<EPI_QKV_ROPE>(a, s); break;
        case EPI_POWER: launchbf<EPI_POWER>(a, s); break;       // A_FRAMES only through the pipelined kernel (K = 1024)
        case EPI_LOGMEL: launchbf<EPI_LOGMEL>(a, s); break;
        default: break;
    }
}

}  // namespace mellow
