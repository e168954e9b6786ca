// KV-cached decode step (replaces the reference's full re-forward per generated token, wrapper.py:216-249;
// SURVEY.md §8a A15/A16).  B = 32 rows per row-block, fp32 everywhere, bit-reproducible (no atomics).
//
// What the profiles of the first two designs showed (profiles/): a decode step moves only
// 1.2 GB, so it is bound by (a) ~4 us per *dependent* launch (inter-kernel data crosses the XCD L2s via the
// memory side), (b) the per-CU vector-memory rate (64 B/clk/CU: a workgroup that gathers an activation matrix
// row by row spends >10k cycles just issuing loads) and (c) serialised round trips (a load behind a branch
// or a runtime-count loop is a full wait).  Hence this design:
//
//   * 5 launches per layer, each spread over 72..288 workgroups:
//       qkv  (split-K slabs) -> attn (row-parallel, key-split) -> o_proj (complete output) ->
//       gate/up (complete output) -> down (split-K slabs, consumed by the next layer's qkv/attn)
//   * every GEMM operand is read with fully coalesced 1 KiB-per-wave loads: weights in P-layout /
//     P16-layout (kernels.h), activations written by their PRODUCER in MFMA fragment order ("F-layout"):
//       F32[rb][k/8][lane][4],  lane = (m%32) + 32*((k%8)/4)   (B operand of v_mfma_f32_32x32x2_f32)
//       F16[rb][k/16][half][lane][4], lane = (m%16) + 16*((k%16)/4), half = (m%32)/16  (…16x16x4_f32)
//   * all loads of a wave are issued up front in straight-line code (compile-time slab counts, clamped
//     addresses instead of branches, sched_barrier between the load block and the math);
//   * the RMSNorm weight is folded into the qkv / gate-up weights at load time and the per-row scale
//     r = rsqrt(mean(x^2)+eps) is applied by the consumer (attention: q,k,v are linear in r; down: inside
//     the SwiGLU), so no normalisation launch exists; split-K partial sums ("slabs") are summed by their
//     consumers in a fixed order.
#include "common.h"
#include "kernels.h"
#include <cstdio>
#include <mutex>
#include <set>
#include <utility>

namespace mellow {

// developer instrumentation: when a debug buffer is set, workgroup 0 / thread 0 stamps s_memtime at phase points
// (compiled in only with -DMELLOW_KDEBUG: the stamps cost ~9 % of a decode step)
__device__ uint64_t* g_kdbg = nullptr;
#ifdef MELLOW_KDEBUG
__device__ __forceinline__ void kstamp(int slot, int idx, bool who) {
    if (g_kdbg && who) g_kdbg[slot * 8 + idx] = __builtin_readcyclecounter();
}
// span of a whole launch: earliest start / latest end over ALL its workgroups on the constant 100 MHz clock (s_memrealtime,
// one counter for the device), at g_kdbg[64 + 2 * seq (+1)]; seq = DecArgs::dbg_seq, set per launch by the host
__device__ __forceinline__ void kspan(int seq, int end) {
    if (g_kdbg && seq >= 0 && seq < 480 && threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        if (end) atomicMax(reinterpret_cast<unsigned long long*>(g_kdbg) + 64 + 2 * seq + 1, t);
        else atomicMin(reinterpret_cast<unsigned long long*>(g_kdbg) + 64 + 2 * seq, t);
    }
}
#else
#define kstamp(slot, idx, who) ((void)0)
#define kspan(seq, end) ((void)0)
#endif
void set_kernel_debug_buffer(uint64_t* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kdbg), &p, sizeof(p)); }

// softmax weights: exp(x) = 2^(x*log2 e) on the hardware v_exp_f32 (x <= 0 here; ~1e-6 relative, far inside the fp32
// summation-order noise of a 64..700-key softmax); exp(-inf) = 0 exactly
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// A/B switches for developer builds (tools/ab_build.sh): -DMELLOW_NO_BLK_EXIT, -DMELLOW_NO_NT
#ifdef MELLOW_NO_BLK_EXIT
#define MELLOW_BLK_EXIT(RB)
#else
#define MELLOW_BLK_EXIT(RB) if (BLK) { if (a.blk_live[RB] == 0) return; }   // BLK: compile-time (a run-time null test costs 0.26 us per launch)
#endif
// streamed-once operands (weights, KV pages): non-temporal loads (`global_load_dwordx4 ... nt`).  Each of these lines is read
// by exactly one workgroup per step, so keeping it in L2 buys nothing, and the nt policy shortens issue -> landed by ~18 %
// on this part (MI355X_MICROARCH.md, row nt-weights)
__device__ __forceinline__ float4 ldg_nt(const float4* p) {
#ifdef MELLOW_NO_NT
    return *p;
#else
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#endif
}
// Decode weights in e4m3 (BASELINE config 5, `W8` variants of the GEMM kernels): the SAME slot order as the fp32 layouts
// (one float4 slot = one 4-byte word of four e4m3 values), one scale per packed weight row, applied to the reduced output.
// W8 == 1: the values are widened to fp32 in registers and multiplied on the fp32 matrix pipe, the activations stay fp32
// (MELLOW_FP8_DECODE_ACT=0: the form the weight pin of the parity suite runs); W8 == 2, the mode's default: see ldw8 below.
template <bool W8>
__device__ __forceinline__ float4 ldw(const float* __restrict__ Wp, int64_t slot) {
    if constexpr (W8) {
        const uint32_t u = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(Wp) + slot);
        const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(u, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(u, true);
        return make_float4(lo[0], lo[1], hi[0], hi[1]);
    } else {
        return ldg_nt(reinterpret_cast<const float4*>(Wp) + slot);
    }
}
// W8 == 2 ("A8"): the e4m3 words go to the fp8 matrix pipe as they are (v_mfma_f32_32x32x16_fp8_fp8 / 16x16x32) and the
// ACTIVATIONS are quantised to e4m3 in the consumer's registers, right after the fp32 fragments have landed: one scale per
// (batch row, k-slice of the wave) = amax / 448 of the values the wave holds for that row, applied to the wave's accumulators
// before the cross-wave reduction -- finer than a per-row scale and free of any cross-workgroup statistic.  No repacking: a
// lane's e4m3 word of k-tile t and of k-tile t+1 form the 8-byte A operand, and the activation fragments of the same two
// tiles (same lane -> same k positions) the B operand; the contraction only needs A and B to agree on which k a byte holds.
__device__ __forceinline__ uint32_t ldw8(const float* __restrict__ Wp, int64_t slot) {
    return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(Wp) + slot);
}
__device__ __forceinline__ float f4amax(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ uint32_t q4_e4m3(float4 v, float inv) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * inv, v.y * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * inv, v.w * inv, w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ long pk64(uint32_t lo, uint32_t hi) { return (long)(((uint64_t)hi << 32) | (uint64_t)lo); }
// N k-tiles of a wave on the fp8 pipe, two tiles per instruction (an odd last tile is paired with zeros)
template <int N>
__device__ __forceinline__ f32x16 mfma8_32(f32x16 acc, const uint32_t (&w)[N], const uint32_t (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pk64(w[i], i + 1 < N ? w[i + 1] : 0u), pk64(x[i], i + 1 < N ? x[i + 1] : 0u), acc, 0, 0, 0);
    return acc;
}
template <int N>
__device__ __forceinline__ f32x4 mfma8_16(f32x4 acc, const uint32_t (&w)[N], const uint32_t (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 2)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pk64(w[i], i + 1 < N ? w[i + 1] : 0u), pk64(x[i], i + 1 < N ? x[i + 1] : 0u), acc, 0, 0, 0);
    return acc;
}
// (amax of a row's slice) -> (1 / scale, scale); an all-zero slice quantises to zeros
__device__ __forceinline__ void a8_scales(float amax, float& inv, float& sc) {
    inv = amax > 0.f ? 448.0f / amax : 0.f;
    sc = amax * (1.0f / 448.0f);
}
// Stores of a kernel's OUTPUT (read by the next launch, on other XCDs).  Developer A/B (tools/ab_build.sh -DMELLOW_ST_MODE=n):
// 0 plain (write-back in this XCD's L2, flushed by the release at the end of the kernel), 1 non-temporal, 2 write-through (sc0 sc1)
#ifndef MELLOW_ST_MODE
#define MELLOW_ST_MODE 0
#endif
__device__ __forceinline__ void st_out(float4* p, float4 v) {
#if MELLOW_ST_MODE == 1
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
#elif MELLOW_ST_MODE == 2
    const f32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
#else
    *p = v;
#endif
}
// A field of the by-value argument struct that only the EPILOGUE needs is otherwise fetched by a scalar load right where it is
// used -- after the reduction barrier, followed at once by s_waitcnt lgkmcnt(0): a cold scalar-cache miss (the argument block's
// lines beyond the preloaded dwords) on the critical tail of every launch, sometimes two in a row.  Naming the value in an empty
// asm right after the kernel's vector loads have been ISSUED makes the compiler fetch it there, where the wait hides behind the
// loads that are in flight anyway.  MEASURED (round 4, same box, tools/ab_run.sh): 46.89 ms of decode per 63 steps with the hoist
// against 46.83 without (B = 64: 68.7 against 68.35) -- those scalar loads hit the scalar cache and cost nothing where they are;
// the wait in front of the load block costs a little.  Off; -DMELLOW_HOIST_ON keeps the A/B.
#ifdef MELLOW_HOIST_ON
#define MELLOW_HOIST(x) asm volatile("" ::"s"(x))
#else
#define MELLOW_HOIST(x) ((void)0)
#endif
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4ssq(float4 v) { return (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
__device__ __forceinline__ f32x16 mfma4(f32x16 acc, float4 w, float4 x) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x.w, acc, 0, 0, 0);
    return acc;
}
// F32-layout float4 index of (row m of row-block rb, columns k..k+3, k % 4 == 0) for an activation with K8 k-tiles
__device__ __forceinline__ int64_t f32_idx(int rb, int K8, int m, int k) {
    return ((int64_t)rb * K8 + (k >> 3)) * 64 + m + 32 * ((k >> 2) & 1);
}

// ----------------------------------------------------------------------------------------------------
// f32x3 mode (the engine's default): the decode GEMMs on the bf16 matrix pipe at fp32 accuracy.  Every fp32 operand value is
// the EXACT sum of three bf16 pieces (split3, common.h: the same split as the prefill GEMMs of gemm_bf16x3.hip), formed here
// IN REGISTERS right after the fp32 fragments have landed -- the weights stay 4 bytes per value in HBM, the activations stay
// fp32 between launches -- and the six largest partial products run on v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD for
// K = 16 against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same K), smallest first, fp32 accumulation.  No repacking:
// a lane's float4 of k-tile t and of k-tile t + 1 are the eight k positions of its bf16 operand; A (weights, P-layout) and
// B (activations, F32-layout) agree on which k a slot holds (lane half h of tile t: k = 8t + 4h + j), which is all the
// contraction needs (the e4m3 form above pairs its tiles the same way).
// Row blocks: one workgroup serves EVERY 32-row block of the batch with the weight fragments it split once (the fp32 kernels
// above replicate the grid per row block and re-stream the weights).
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_pair(float4 a, float4 b, i32x4& p0, i32x4& p1, i32x4& p2) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    split8(v, p0, p1, p2);
}
#define MELLOW_BF16(W, X, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, X), ACC, 0, 0, 0)
// six of the nine partial products, smallest first (pieces: 0 = leading, 2 = trailing; dropped: (1,2), (2,1), (2,2) < 2^-23 |w x|)
__device__ __forceinline__ f32x16 mma6(f32x16 acc, const i32x4 (&w)[3], const i32x4 (&x)[3]) {
    MELLOW_BF16(w[2], x[0], acc);
    MELLOW_BF16(w[0], x[2], acc);
    MELLOW_BF16(w[1], x[1], acc);
    MELLOW_BF16(w[1], x[0], acc);
    MELLOW_BF16(w[0], x[1], acc);
    MELLOW_BF16(w[0], x[0], acc);
    return acc;
}

// Pre-split activations ("F3" layouts): where ONE launch produces an activation that MANY workgroups of the next launch
// multiply, the producer also stores it as bf16 triples in the consumer's fragment order, so that the split is paid once per
// value instead of once per consuming workgroup (the lm_head has 1536 of them).  6 bytes per value instead of 4.
//   F3-32 (B operand of v_mfma_f32_32x32x16_bf16; pairs T = K / 16):
//       16-byte slot ((rb * K/16 + T) * 3 + piece) * 64 + lane,  lane = m % 32 + 32 h;  element e of the slot: k = 16 T + 8 h + e
//   F3-16 (B operand of v_mfma_f32_16x16x32_bf16; pairs T = K / 32, row halves mh):
//       slot (((rb * K/32 + T) * 2 + mh) * 3 + piece) * 64 + lane,  lane = m % 16 + 16 q;  k = 32 T + 8 q + e
// Eight CONSECUTIVE k per slot: a producer thread owns four consecutive k of one row, its neighbour (lane ^ 1) the other four
// of the same slot, so the even thread of a pair collects both halves (one DPP step) and stores whole 16-byte slots -- no
// workgroup ever writes half a slot (half-slot stores from two workgroups on different XCDs cost 0.9 us per launch).
// The weight side agrees on the k order: a 32-row weight tile pair (P-layout: lane half h of tile t holds k = 8 t + 4 h + j) is
// brought to k = 16 T + 8 h + e with one v_permlane32_swap per value (pair_natural); the 16-row gate/up tiles have their own
// packed copy in that order (P16N, launch_pack_weight16n).
struct F3Quad { uint2 p[3]; };      // four consecutive k of one row as bf16 triples: p[piece] = 4 bf16
__device__ __forceinline__ F3Quad f3_split4(float4 y) {
    __bf16 h[4], m[4], l[4];
    split3(y.x, h[0], m[0], l[0]); split3(y.y, h[1], m[1], l[1]); split3(y.z, h[2], m[2], l[2]); split3(y.w, h[3], m[3], l[3]);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    F3Quad q;
    q.p[0] = __builtin_bit_cast(uint2, bf16x4{h[0], h[1], h[2], h[3]});
    q.p[1] = __builtin_bit_cast(uint2, bf16x4{m[0], m[1], m[2], m[3]});
    q.p[2] = __builtin_bit_cast(uint2, bf16x4{l[0], l[1], l[2], l[3]});
    return q;
}
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
// the neighbour's quad (lane ^ 1).  Both threads of a pair must be active.
__device__ __forceinline__ F3Quad f3_partner(const F3Quad& q) {
    F3Quad o;
#pragma unroll
    for (int c = 0; c < 3; ++c) { o.p[c].x = dpp_xor1(q.p[c].x); o.p[c].y = dpp_xor1(q.p[c].y); }
    return o;
}
// whole slot (columns k .. k + 7, k % 8 == 0): lo = the quad of k .. k + 3, hi = of k + 4 .. k + 7
__device__ __forceinline__ void f3_store8(void* base, int64_t slot0, const F3Quad& lo, const F3Quad& hi) {
    uint4* p = reinterpret_cast<uint4*>(base) + slot0;
    p[0] = make_uint4(lo.p[0].x, lo.p[0].y, hi.p[0].x, hi.p[0].y);
    p[64] = make_uint4(lo.p[1].x, lo.p[1].y, hi.p[1].x, hi.p[1].y);
    p[128] = make_uint4(lo.p[2].x, lo.p[2].y, hi.p[2].x, hi.p[2].y);
}
// slot of piece 0 for (row m of block rb, columns k .. k + 7, k % 8 == 0)
__device__ __forceinline__ int64_t f3_32_slot(int rb, int KP, int m, int k) {
    return ((int64_t)rb * KP + (k >> 4)) * 3 * 64 + m + 32 * ((k >> 3) & 1);
}
__device__ __forceinline__ int64_t f3_16_slot(int rb, int KP, int m, int k) {
    return (((int64_t)rb * KP + (k >> 5)) * 2 + (m >> 4)) * 3 * 64 + (m & 15) + 16 * ((k >> 3) & 3);
}
// weight fragments of a tile pair (a = tile 2T, b = tile 2T + 1, P-layout) -> the slot order above: lane half 0 gets tile 2T's
// eight k, lane half 1 tile 2T + 1's
__device__ __forceinline__ void pair_natural(float4& a, float4& b) {
    float* pa = &a.x; float* pb = &b.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(pa[j]), __float_as_uint(pb[j]), false, false);
        pa[j] = __uint_as_float(r[0]);
        pb[j] = __uint_as_float(r[1]);
    }
}

// ----------------------------------------------------------------------------------------------------
// K1  qkv projection, split-K.  grid (30 n-tiles, DEC_KC_QKV, RB) = 240 workgroups, 3 waves x 3 k-tiles (+1 epilogue wave).
//     X = baseF + sum_{s<KCD} dslabF[s]  (residual stream, un-normalised; norm weight folded into W)
//     out: pq[kc][row][960] row-major slabs (consumer = attention, row-parallel)
// ----------------------------------------------------------------------------------------------------
#ifndef MELLOW_QKV_WAVES
#define MELLOW_QKV_WAVES 9     // 9 x 1 k-tile: 53.0 vs 53.6 ms of decode per 63 steps with 3 x 3
#endif
constexpr int QW = MELLOW_QKV_WAVES;                 // compute waves: 3 x 3 k-tiles or 9 x 1
constexpr int QKV_THREADS = QW * 64 < 256 ? 256 : QW * 64;
template <int KCD, bool BLK, bool FIRST, int W8>      // W8: 0 fp32 weights, 1 e4m3 weights widened, 2 e4m3 weights and activations (fp8 MFMA)
__global__ __launch_bounds__(QKV_THREADS) void dec_qkv_kernel(const float* __restrict__ Wp, const float* __restrict__ xmidF_p,
                                                              const float* __restrict__ dslabF_p, int K8p, const DecArgs a,
                                                              const float* __restrict__ wscale) {
    kspan(a.dbg_seq, 0);
    __shared__ __attribute__((aligned(16))) float red[QW * 16 * 64];
    __shared__ float ssq_s[QW * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, kc = blockIdx.y, rb = blockIdx.z;
    constexpr int KPW = 72 / (DEC_KC_QKV * QW);   // k-tiles per compute wave (with QW = 3, wave 3 only helps the epilogue)
    // The first kernel of a step (a.first): advance the position word (nothing of the previous step reads it any more)
    // and stage this position's RoPE row at a fixed address, so that no attention kernel of the step has to chase
    // pos -> table row (two dependent round trips).  One half-wave: the loads of *d_pos precede the store in program order.
    if (FIRST && nt == 29 && kc == 0 && rb == 0 && tid < 32) {     // FIRST: compile-time, like BLK
        const int p = *a.d_pos + a.inc_pos;
        const float c = a.rope_cos[(int64_t)p * 32 + tid], sn = a.rope_sin[(int64_t)p * 32 + tid];
        a.rope_cur[tid] = c;
        a.rope_cur[32 + tid] = sn;
        if (tid == 0 && a.inc_pos) *a.d_pos = p;
    }
    MELLOW_BLK_EXIT(rb)      // every row of this block has stopped (workgroup-uniform)
    if (wave < QW) {
        const int k8_0 = (kc * QW + wave) * KPW;
        const int64_t wslot = ((int64_t)nt * K8p + k8_0) * 64 + lane;
        const float4* xb = reinterpret_cast<const float4*>(xmidF_p) + ((int64_t)rb * 72 + k8_0) * 64 + lane;
        const float4* sb = reinterpret_cast<const float4*>(dslabF_p) + ((int64_t)rb * 72 + k8_0) * 64 + lane;
        float4 w[KPW], x[KPW], sl[KPW][KCD > 0 ? KCD : 1];
        uint32_t w8[KPW];
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            if constexpr (W8 == 2) w8[i] = ldw8(Wp, wslot + i * 64);
            else w[i] = ldw<W8 != 0>(Wp, wslot + i * 64);
            x[i] = xb[i * 64];
#pragma unroll
            for (int s = 0; s < KCD; ++s) sl[i][s] = sb[(int64_t)s * a.slabF_stride4 + i * 64];
        }
        MELLOW_HOIST(a.pq); MELLOW_HOIST(a.rows); MELLOW_HOIST(a.ssq1); MELLOW_HOIST(a.xnewR);
        if (W8) MELLOW_HOIST(wscale);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float ssp = 0.f;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            float4 xv = x[i];
#pragma unroll
            for (int s = 0; s < KCD; ++s) xv = f4add(xv, sl[i][s]);
            if constexpr (W8 == 2) x[i] = xv;            // quantised below, once the wave's slice of the row is known
            else acc = mfma4(acc, w[i], xv);
            // side jobs of two n-tiles (the workgroups of one k-chunk see every row of x_new on these 72 columns):
            //   nt == 0: partial sum of squares per row (the attention's RMS statistic)
            //   nt == 1: x_new row-major (the o_proj's residual operand)
            if (nt == 0) ssp += f4ssq(xv);
            if (nt == 1)
                *reinterpret_cast<float4*>(a.xnewR + ((int64_t)rb * 32 + (lane & 31)) * 576 + (k8_0 + i) * 8 + (lane >> 5) * 4) = xv;
        }
        if constexpr (W8 == 2) {
            float am = 0.f, inv, sc;
#pragma unroll
            for (int i = 0; i < KPW; ++i) am = fmaxf(am, f4amax(x[i]));
            a8_scales(half_max(am), inv, sc);
            uint32_t xq[KPW];
#pragma unroll
            for (int i = 0; i < KPW; ++i) xq[i] = q4_e4m3(x[i], inv);
            acc = mfma8_32<KPW>(acc, w8, xq);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= sc;
        }
        if (nt == 0) {
            ssp = half_sum(ssp);                 // the two k-halves of a row sit in lanes m and m + 32
            if (lane < 32) ssq_s[wave * 32 + lane] = ssp;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (nt == 0 && tid < 32)
    {
        float ssum = ssq_s[tid & 31];
#pragma unroll
        for (int wv = 1; wv < QW; ++wv) ssum += ssq_s[wv * 32 + (tid & 31)];
        a.ssq1[((int64_t)rb * 32 + tid) * DEC_KC_QKV + kc] = ssum;
    }
    if (tid >= 256) return;                              // (QW = 9 only) the epilogue is 256 threads wide; no barrier follows
    const int mm = tid & 31, hh = (tid >> 5) & 1, gq = tid >> 6;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * gq + j;
        float sacc = red[r * 64 + mm + 32 * hh];
#pragma unroll
        for (int wv = 1; wv < QW; ++wv) sacc += red[(wv * 16 + r) * 64 + mm + 32 * hh];
        v[j] = sacc;
    }
    const int n = nt * 32 + 8 * gq + 4 * hh;
    if (W8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= wscale[n + j];
    }
    if (n < 960)
        *reinterpret_cast<float4*>(a.pq + ((int64_t)kc * a.rows + rb * 32 + mm) * 960 + n) = make_float4(v[0], v[1], v[2], v[3]);
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// K2  attention (flash decoding).  grid (3 kv heads, rows, key splits: DEC_TS at one row block, 1 from two on), 8 waves.
//     r1 = RMS scale of x_new from the qkv kernel's per-chunk sums;  q,k,v = r1 * sum_kc pq slabs;  RoPE; KV append;
//     per-wave online softmax over an interleaved set of 4-key groups; partial (m, l, o) per split.
//     Split sp owns key groups [sp*gs, (sp+1)*gs) (the last split: everything from sp*gs on, plus the new key).
// ----------------------------------------------------------------------------------------------------
#ifndef MELLOW_DA_ABL
#define MELLOW_DA_ABL 0      // developer ablation (wrong results, timing only): 1 no K/V page loads, 2 no score / softmax / PV loop, 4 no slab prologue loads
#endif
#ifndef MELLOW_DA_WAVES
#define MELLOW_DA_WAVES 8
#define MELLOW_DA_G 7
#endif
#ifndef MELLOW_DA_MINW
#define MELLOW_DA_MINW 4      // waves per SIMD the register allocation must allow: 4 = two 8-wave workgroups per CU (128 VGPRs), so that
#endif                        // the 384 workgroups of a 64-row batch are resident together instead of in two rounds
constexpr int DA_MINW = MELLOW_DA_WAVES == 8 || MELLOW_DA_WAVES == 4 ? MELLOW_DA_MINW : 1;
constexpr int DA_WAVES = MELLOW_DA_WAVES;
constexpr int DA_G = MELLOW_DA_G;   // 4-key groups in flight per wave: one chunk covers 2 * DA_WAVES * DA_G * 4 = 448 keys
#ifndef MELLOW_DA_G1
#define MELLOW_DA_G1 5      // 5 of 7: 48.55 ms of decode per 63 steps; 2 / 3 / 4 / 7 (= everything up front): 49.35 / 49.15 / 49.0 / 49.75 (same box)
#endif
constexpr int DA_G1 = MELLOW_DA_G1 < DA_G ? MELLOW_DA_G1 : DA_G;    // key groups requested before the prologue
// bf16 pages (KV16, the fp8 mode): the SCORES on the matrix pipe.  Ablations of the vector form at B = 128 (same box, decode per 63
// steps): 75.5 ms; without the K/V loads 67.1; without the score / softmax / PV loop 60.3 (per 28 keys of a wave ~500 vector
// instructions: 84 of them the 16-lane DPP sums of 21 dot products, ~80 hazard s_nops, 63
// redundant exps).  Here a wave's chunk is ONE 32-key tile (~200 vector instructions): lane (key kk = lane % 32, half hf = lane / 32)
// loads the 64 contiguous bytes of its key's dims 32 hf .. 32 hf + 31 (four 16-byte loads = the B operands of four
// v_mfma_f32_32x32x16_bf16, k order: step s <-> dims 32 hf + 8 s ..), the A operand holds q as EXACT bf16 triples in rows
// 4 piece + head (rows are free: 32 of them, 9 used), so S = q . k is the fp32 dot product of the fp32 q with the bf16 key up to
// summation order, one value per (key lane, head) -- no cross-lane sums, 3 exps per key.  The weights go through 4 KiB of LDS
// (wave-private, in-order: no barrier) to the lanes of the unchanged P V accumulation (lane = key sub x dim quad).
// Same box, fp8 mode, decode per 63 steps: B = 128 75.9 -> 72.4 ms, B = 32 43.9 -> 41.7; with the loads compiled out the two forms
// cost the same (66.5 / 67.1): the loop is a dependent chain (scores -> max -> exp -> weights -> P V) more than an issue stream, and
// what the matrix form wins is the chain's length and the registers (116 instead of 218 at one row block).
#ifndef MELLOW_DA16_MFMA
#define MELLOW_DA16_MFMA 1
#endif
constexpr int DA_GM = 8;            // key groups per wave and chunk of the matrix form: 32 keys = one MFMA tile
#ifndef MELLOW_DA16_KNT
#define MELLOW_DA16_KNT 0           // K tile loads plain: a lane's four 16-byte pieces share a 64-byte segment that four instructions touch, and a
#endif                              // non-temporal line does not stay for the next one (same box, fp8 B = 128: 78.9 ms with nt, 72.5 without)
// Measured and dropped: the tile as 32 consecutive keys loaded in 1 KiB runs and turned into the operand order through a swizzled
// LDS image (72.8 against 72.5 ms: the LDS round trip costs what the coalescing gains).
// Physical waves (template parameter NW): the eight chunk streams of a workgroup ("virtual waves": own weights region, own
// partial in the merge) can run on 8 waves or, two after the other, on 4 -- the same arithmetic in the same order, so the
// results are bit-identical and four 256-thread workgroups share a CU (the 768 workgroups of B = 128 resident together instead
// of in 1.5 rounds).  Measured +-0 (fp8 B = 128, same box: 72.4 ms on 8 waves, 72.4-73.5 on 4): MELLOW_DA16_NW stays 8.
#ifndef MELLOW_DA16_MINW
#define MELLOW_DA16_MINW MELLOW_DA_MINW
#endif
#ifndef MELLOW_DA16_NW
#define MELLOW_DA16_NW 8            // physical waves of the matrix form from two row blocks on (4 = two virtual waves per wave)
#endif

// KV16 (fp8 mode): the decode step reads and extends a bf16 SHADOW of the K/V pages (engine_lm.cpp: converted from the fp32
// pages the prefill wrote, once per call): half the bytes of the step's largest stream.  The new key / value of the step itself
// are used in fp32 (from LDS) and appended rounded to nearest even.  One 8-byte load per lane = 4 bf16 of one key.
template <bool KV16>
__device__ __forceinline__ float4 ld_kv(const float* page, int64_t elem) {
    if constexpr (KV16) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(page) + elem));
        return make_float4(__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u));
    } else {
        return ldg_nt(reinterpret_cast<const float4*>(page + elem));
    }
}
template <bool KV16>
__device__ __forceinline__ void st_kv(float* page, int64_t elem, float v) {
    if constexpr (KV16) reinterpret_cast<__bf16*>(page)[elem] = static_cast<__bf16>(v);
    else page[elem] = v;
}
// FUSED: the producer was dec_qkv2_kernel (the previous layer's down projection and this layer's q/k/v in one launch): the
// projected values arrive as Q2_NPQ slabs, and x_new = x_mid + sum of the Q2_HC down slabs is formed HERE (the 144 float4 of the
// row, by every workgroup of the row: its sum of squares is the RMS statistic; the (kv head 0, split 0) workgroup also writes the
// row for the o_proj's residual).  !FUSED: the producer was dec_qkv_kernel (first layer of a step: 8 slabs + its own statistic).
// ONE: the launch has a single 32-row block (192 workgroups: one per CU, registers are free) -> the chunk loop in its plain form
//      (221 VGPRs, 0.7 ms of decode per 63 steps faster at B = 32); otherwise the first chunk is peeled by hand so that the kernel
//      fits 128 VGPRs and two workgroups share a CU (B = 64: 384 workgroups resident together, 73.7 -> 71.2 ms)
template <bool BLK, bool FUSED, bool ONE, bool KV16, int NW, int TS>
__global__ __launch_bounds__(NW * 64, ONE ? 2 : (KV16 && MELLOW_DA16_MFMA && DA_MINW > 1 ? MELLOW_DA16_MINW : DA_MINW)) void dec_attn_kernel(float* __restrict__ k_cache, float* __restrict__ v_cache,
                                                                  const int32_t* __restrict__ d_pos_p, const float* __restrict__ pq_p,
                                                                  const float* __restrict__ xmidF_p, int Tmax_p, int gs_p, int rows_p,
                                                                  const DecArgs a) {
    kspan(a.dbg_seq, 0);
    // (the leading scalar arguments repeat fields of `a`: the first 14 dwords of the argument block are preloaded into SGPRs
    //  by the command processor -- build.py, -amdgpu-kernarg-preload-count -- so the K/V and slab addresses do not wait for a
    //  scalar load from the argument buffer at the start of every launch)
    __shared__ float ssq_part[3];
    __shared__ __attribute__((aligned(16))) float qs[3 * 64];            // RoPE'd, pre-scaled q
    __shared__ __attribute__((aligned(16))) float knew[64], vnew[64];
    __shared__ __attribute__((aligned(16))) float xs[320];               // q (3 x 64) | k | v before RoPE, RMS-scaled
    __shared__ __attribute__((aligned(16))) float ored[DA_WAVES * 3 * 64];
    __shared__ float mred[DA_WAVES * 3], lred[DA_WAVES * 3];
    __shared__ float snew_s[3];
    constexpr bool MF = KV16 && MELLOW_DA16_MFMA != 0;                   // scores on the matrix pipe (comment at DA_GM)
    __shared__ __attribute__((aligned(16))) float pl[MF ? DA_WAVES * 32 * 4 : 4];     // softmax weights [wave][key of the tile][head | pad]
    __shared__ __attribute__((aligned(16))) __bf16 qb[MF ? 10 * 64 : 8];               // q as bf16 triples [piece][head][dim] + a zero row
    constexpr int VPW = DA_WAVES / NW;                                   // virtual waves per physical wave (matrix form only)
    static_assert(DA_WAVES % NW == 0 && (MF || NW == DA_WAVES), "the vector form runs one chunk stream per wave");
    // key group of slot u in the chunk that starts at group g0: the interleaved groups of the vector form
    auto mf_group = [&](int g0, int u) { return g0 + u * DA_WAVES; };

    static_assert(TS == DEC_TS || !(ONE || KV16), "one row block and the bf16 pages always run DEC_TS key splits (kernels.h dec_key_splits)");
    const int g = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    int row = b;                  // the example whose KV pages this slot reads and extends (slot == example unless rows migrate)
    if (BLK) {
        const int live = a.blk_live[b >> 5];
        if (a.row_of_slot) row = a.row_of_slot[b];
        if (live == 0 || row < 0) return;       // the block has stopped / the slot is empty (workgroup-uniform)
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tmax = Tmax_p;
    // (KV16: the same element offsets into pages of 2-byte elements: half the float offset)
    float* kpage = k_cache + (((int64_t)row * 3 + g) * Tmax * 64 >> (KV16 ? 1 : 0));
    float* vpage = v_cache + (((int64_t)row * 3 + g) * Tmax * 64 >> (KV16 ? 1 : 0));
    const int sub = lane >> 4, quad = lane & 15;   // lane -> (key sub, dim quad): a wave instruction = 4 keys x 64 dims

    // ===== ONE round trip, every load independent and straight-line: position word, RMS partials, qkv slabs,
    //       the staged RoPE row and the first chunk of K/V (key ranges of the splits are fixed per launch, a.gs,
    //       so their addresses do not wait for the position; keys >= pos are masked after the loads land) =====
    const bool dbg = tid == 0 && g == 0 && b == 0 && sp == 0;
    kstamp(1, 0, dbg);
#ifdef MELLOW_POS_VECTOR_LOAD
    const int pos = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(d_pos_p));
#else
    const int pos = *d_pos_p;        // keys 0..pos-1 are cached; the new key is key `pos`
#endif
    // prologue: the 320 projected values of this (row, kv head) -- q of 3 heads | k | v -- are the sums of the qkv kernel's
    // split-K slabs.  80 threads own one float4 of columns each (8 slab loads of 16 B; before: 512 threads x 16 dword loads),
    // scale by the RMS statistic and park the result in LDS; 192 threads then apply RoPE / append the new key and value.
    //   r <48: query head 3g + r/16, columns 4(r%16)..   r <64: key columns 4(r-48)..   r <80: value columns 4(r-64)..
    const int hsel = (tid >> 5) & 3, i = tid & 31;
    const int r = tid < 80 ? tid : 0;
    const int pcol = r < 48 ? (3 * g + (r >> 4)) * 64 + 4 * (r & 15) : r < 64 ? 576 + g * 64 + 4 * (r - 48) : 768 + g * 64 + 4 * (r - 64);
    const float* prow = pq_p + (int64_t)b * 960 + pcol;
    constexpr int NPQ = FUSED ? Q2_NPQ : DEC_KC_QKV;
    float4 sl4[NPQ], sq4[2];
    if (tid < 80) {
#pragma unroll
        for (int s = 0; s < NPQ; ++s)
            sl4[s] = (MELLOW_DA_ABL & 4) ? make_float4(0.1f * s, 0.2f, 0.3f, 0.4f) : *reinterpret_cast<const float4*>(prow + (int64_t)s * rows_p * 960);
        if (!FUSED) {
            sq4[0] = reinterpret_cast<const float4*>(a.ssq1 + (int64_t)b * DEC_KC_QKV)[0];
            sq4[1] = reinterpret_cast<const float4*>(a.ssq1 + (int64_t)b * DEC_KC_QKV)[1];
        }
    }
    // FUSED: float4 c (columns 4c..4c+3) of the residual row, by thread c < 144: x_mid + the down slabs
    float4 xr4 = make_float4(0.f, 0.f, 0.f, 0.f), xd4[FUSED ? Q2_HC : 1];
    if (FUSED && tid < 144) {
        const int64_t fi = ((int64_t)(b >> 5) * 72 + (tid >> 1)) * 64 + (b & 31) + 32 * (tid & 1);      // f32_idx(rb, 72, m, 4 tid)
        xr4 = reinterpret_cast<const float4*>(xmidF_p)[fi];
#pragma unroll
        for (int s = 0; s < Q2_HC; ++s) xd4[s] = reinterpret_cast<const float4*>(a.dslabF)[(int64_t)s * a.slabF_stride4 + fi];
    }
    const float c = a.rope_cur[i], sn = a.rope_cur[32 + i];
    const int gbeg = sp * gs_p;                                   // groups of 4 keys
    const int gend_fixed = sp == TS - 1 ? 0x3fffffff : gbeg + gs_p;
    // The K/V stream of a workgroup (107 KB at 420 keys) is bound by what the memory side delivers per CU (~12 B/clk): a wave
    // that issues all 14 page loads up front sits in the issue queue for ~4 us, and the prologue's barriers wait for the
    // slowest wave.  So only the first DA_G1 key groups are requested before the prologue (enough bytes in flight to keep the
    // stream busy while it runs: slab sums, RoPE, two barriers); the rest is requested after it, and the score loop consumes
    // the groups in arrival order.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int kk = lane & 31, hf = lane >> 5;                     // matrix form: lane -> (key of the 32-key tile, half of its dims)
    i32x4 kq[4];
    u32x2 vq[DA_GM];
    auto load_ktile = [&](int g0) {
        const int gi = g0 + (kk >> 2) * DA_WAVES;
        const int tc = min(gi * 4 + (kk & 3), Tmax - 1);
        const i32x4* kp = reinterpret_cast<const i32x4*>(reinterpret_cast<const uint16_t*>(kpage) + (int64_t)tc * 64 + hf * 32);
#pragma unroll
        for (int st = 0; st < 4; ++st)
            kq[st] = (MELLOW_DA_ABL & 1) ? i32x4{tc, 0x3c003c00, st, 0x3c003c00} : (MELLOW_DA16_KNT ? __builtin_nontemporal_load(kp + st) : kp[st]);
    };
    auto load_vgroup = [&](int g0, int u) {
        const int gi = mf_group(g0, u);
        const int tc = min(gi * 4 + sub, Tmax - 1);
        vq[u] = (MELLOW_DA_ABL & 1) ? u32x2{0x3c003c00u, (unsigned)tc}
                                    : __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(vpage) + (int64_t)tc * 64 + quad * 4));
    };
    float4 k4[MF ? 1 : DA_G], v4[MF ? 1 : DA_G];
    auto load_group = [&](int u) {
        const int gi = gbeg + wave + u * DA_WAVES;
        const int tc = min(gi * 4 + sub, Tmax - 1);               // inside the page; validity is decided later
        if (MELLOW_DA_ABL & 1) {
            k4[u] = make_float4(0.01f * tc, 0.02f, 0.03f, 0.04f);
            v4[u] = make_float4(0.01f, 0.02f * tc, 0.03f, 0.04f);
        } else {
            k4[u] = ld_kv<KV16>(kpage, (int64_t)tc * 64 + quad * 4);
            v4[u] = ld_kv<KV16>(vpage, (int64_t)tc * 64 + quad * 4);
        }
    };
    const int g_first_mf = gbeg + wave;                           // virtual wave wave + NW j starts at group g_first_mf + NW j
    if constexpr (MF) {
        load_ktile(g_first_mf);
#pragma unroll
        for (int u = 0; u < DA_GM / 2; ++u) load_vgroup(g_first_mf, u);
    } else {
#pragma unroll
        for (int u = 0; u < DA_G1; ++u) load_group(u);
    }
    MELLOW_HOIST(a.attF16); MELLOW_HOIST(a.att_ml); MELLOW_HOIST(a.RB); MELLOW_HOIST(a.eps);
    if (FUSED) MELLOW_HOIST(a.xnewR);
    __builtin_amdgcn_sched_barrier(0);
    kstamp(1, 1, dbg);

    const int ngroups = (pos + 3) >> 2;                           // groups of 4 cached keys
    const int gend = min(ngroups, gend_fixed);
    static_assert(DEC_KC_QKV == 8, "fixed summation order below");
    if (FUSED) {
        if (tid < 192) {                     // waves 0..2: the residual row and its sum of squares (threads 144..191 add zeros)
            float4 x = xr4;
#pragma unroll
            for (int s = 0; s < Q2_HC; ++s) x = f4add(x, xd4[s]);                // slab 0, then 1, ...
            const float ss = wave_sum(tid < 144 ? f4ssq(x) : 0.f);
            if (lane == 0) ssq_part[wave] = ss;
            if (g == 0 && sp == 0 && tid < 144) reinterpret_cast<float4*>(a.xnewR + (int64_t)b * 576)[tid] = x;
        }
        if (tid < 80) {
            float4 x = sl4[0];
#pragma unroll
            for (int s = 1; s < NPQ; ++s) x = f4add(x, sl4[s]);
            *reinterpret_cast<float4*>(xs + 4 * tid) = x;          // scaled by the RMS statistic after the barrier
        }
    } else if (tid < 80) {
        float4 x = sl4[0];
#pragma unroll
        for (int s = 1; s < DEC_KC_QKV; ++s) x = f4add(x, sl4[s]);           // slab 0, then 1, ... (the order of every build)
        const float ssum = ((sq4[0].x + sq4[0].y) + (sq4[0].z + sq4[0].w)) + ((sq4[1].x + sq4[1].y) + (sq4[1].z + sq4[1].w));
        const float rscale = 1.0f / sqrtf(ssum / 576.0f + a.eps);
        *reinterpret_cast<float4*>(xs + 4 * tid) = make_float4(x.x * rscale, x.y * rscale, x.z * rscale, x.w * rscale);
    }
    __syncthreads();
    kstamp(1, 2, dbg);
    const float rs2 = FUSED ? 1.0f / sqrtf(((ssq_part[0] + ssq_part[1]) + ssq_part[2]) / 576.0f + a.eps) : 1.0f;
    if (tid < 128) {
        const int base = hsel < 3 ? hsel * 64 : 192;
        const float x1 = xs[base + i] * rs2, x2 = xs[base + i + 32] * rs2;
        const float o1 = __fadd_rn(__fmul_rn(x1, c), __fmul_rn(-x2, sn));
        const float o2 = __fadd_rn(__fmul_rn(x2, c), __fmul_rn(x1, sn));
        if (hsel < 3) {
            qs[hsel * 64 + i] = o1 * 0.125f;        // head_dim^-0.5 = 1/8 exactly
            qs[hsel * 64 + i + 32] = o2 * 0.125f;
            if constexpr (MF) {                     // the same values as exact bf16 triples: the A operand of the score MFMAs
                __bf16 p0, p1, p2;
                split3(o1 * 0.125f, p0, p1, p2);
                qb[(0 * 3 + hsel) * 64 + i] = p0; qb[(1 * 3 + hsel) * 64 + i] = p1; qb[(2 * 3 + hsel) * 64 + i] = p2;
                split3(o2 * 0.125f, p0, p1, p2);
                qb[(0 * 3 + hsel) * 64 + i + 32] = p0; qb[(1 * 3 + hsel) * 64 + i + 32] = p1; qb[(2 * 3 + hsel) * 64 + i + 32] = p2;
            }
        } else {
            knew[i] = o1; knew[i + 32] = o2;
            if (sp == 0) { st_kv<KV16>(kpage, (int64_t)pos * 64 + i, o1); st_kv<KV16>(kpage, (int64_t)pos * 64 + i + 32, o2); }
        }
    } else if (tid < 192) {
        const float x1 = xs[256 + tid - 128] * rs2;
        vnew[tid - 128] = x1;
        if (sp == 0) st_kv<KV16>(vpage, (int64_t)pos * 64 + (tid - 128), x1);
    } else if (MF && tid < 256) {
        qb[9 * 64 + tid - 192] = static_cast<__bf16>(0.f);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MF) {
#pragma unroll
        for (int u = DA_GM / 2; u < DA_GM; ++u) load_vgroup(g_first_mf, u);
    } else {
#pragma unroll
        for (int u = DA_G1; u < DA_G; ++u) load_group(u);
    }
    __builtin_amdgcn_sched_barrier(0);
    kstamp(1, 3, dbg);

    if (wave < 3) {                      // score of the new key (q . k_new, both in LDS), consumed after the last barrier
        const float sn_ = wave_sum(qs[wave * 64 + lane] * knew[lane]);
        if (lane == 0) snew_s[wave] = sn_;
    }
    float4 q4[3];
    // matrix form: row 4 piece + head of the A operand in the k order of the key lanes (step s: dims 32 hf + 8 s ..); rows without
    // a (piece, head) read the zero row of qb.  Read from LDS per chunk: 16 registers less across the loop.
    const int aq_off = ((kk >> 2) < 3 && (kk & 3) < 3 ? (kk >> 2) * 3 + (kk & 3) : 9) * 64 + hf * 32;
    if constexpr (!MF) {
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) q4[hh] = *reinterpret_cast<const float4*>(qs + hh * 64 + quad * 4);
    }
    float m_run[3] = {-INFINITY, -INFINITY, -INFINITY};
    float l_run[3] = {0.f, 0.f, 0.f};      // per-lane partial (this lane's keys only)
    float4 acc[3];
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);

    // one chunk = the DA_G key groups a wave holds in registers (masked by weight, exp(-inf) = 0: slots beyond the context hold
    // finite values, engine_lm.cpp clear_page_tails; the 16 dim-quads of a key are one DPP row).  The first chunk (every context
    // up to 448 keys per split) works on the registers loaded above; further chunks (longer contexts) reload and repeat.
    // Two code shapes around the same body (a macro, so that both are the literal source text):
    //   ONE   the plain loop; the compiler keeps a second set of K/V registers alive across the back edge (220 VGPRs)
    //   !ONE  the first chunk peeled by hand: 118 VGPRs, two workgroups per CU (DA_MINW)
#define MELLOW_DA_CHUNK(g0)                                                                                              \
    {                                                                                                                    \
        float sc[DA_G][3];                                                                                               \
        float cmax[3] = {-INFINITY, -INFINITY, -INFINITY};                                                               \
_Pragma("unroll")                                                                                                        \
        for (int u = 0; u < DA_G; ++u) {                                                                                 \
            const int gi = g0 + u * DA_WAVES;                                                                            \
            const int t = gi * 4 + sub;                                                                                  \
            const bool ok = gi < gend && t < pos;                                                                        \
_Pragma("unroll")                                                                                                        \
            for (int hh = 0; hh < 3; ++hh) {                                                                             \
                float sv = q4[hh].x * k4[u].x + q4[hh].y * k4[u].y + q4[hh].z * k4[u].z + q4[hh].w * k4[u].w;            \
                sv = row16_sum(sv);                                                                                      \
                sv = ok ? sv : -INFINITY;                                                                                \
                sc[u][hh] = sv;                                                                                          \
                cmax[hh] = fmaxf(cmax[hh], sv);                                                                          \
            }                                                                                                            \
        }                                                                                                                \
_Pragma("unroll")                                                                                                        \
        for (int hh = 0; hh < 3; ++hh) {                                                                                 \
            float cm = cmax[hh];                                                                                         \
            cm = fmaxf(cm, swz_xor16(cm));                                                                               \
            cm = half_max(cm);                                                                                           \
            const float m_new = fmaxf(m_run[hh], cm);                                                                    \
            const float alpha = fast_exp(m_run[hh] - m_new);                                                             \
            m_run[hh] = m_new;                                                                                           \
            float lsum = 0.f;                                                                                            \
            float4 o = make_float4(acc[hh].x * alpha, acc[hh].y * alpha, acc[hh].z * alpha, acc[hh].w * alpha);          \
_Pragma("unroll")                                                                                                        \
            for (int u = 0; u < DA_G; ++u) {                                                                             \
                const float p = fast_exp(sc[u][hh] - m_new);                                                             \
                lsum += p;                                                                                               \
                o.x += p * v4[u].x; o.y += p * v4[u].y; o.z += p * v4[u].z; o.w += p * v4[u].w;                          \
            }                                                                                                            \
            acc[hh] = o;                                                                                                 \
            l_run[hh] = l_run[hh] * alpha + lsum;                                                                        \
        }                                                                                                                \
    }
    const int g_first = gbeg + wave, g_stop = (MELLOW_DA_ABL & 2) ? gbeg : gend;
#define MELLOW_DA_RELOAD(g0)                                                                                             \
    {                                                                                                                    \
        _Pragma("unroll") for (int u = 0; u < DA_G; ++u) {                                                               \
            const int gi = (g0) + u * DA_WAVES;                                                                          \
            const int tc = min(gi * 4 + sub, Tmax - 1);                                                                  \
            k4[u] = ld_kv<KV16>(kpage, (int64_t)tc * 64 + quad * 4);                                                    \
            v4[u] = ld_kv<KV16>(vpage, (int64_t)tc * 64 + quad * 4);                                                    \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
    // matrix form of a chunk (KV16): scores by four MFMAs, one softmax weight per (key lane, head), P V as above
    auto chunk_mf = [&](int vw, int g0) {
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const i32x4 aq = *reinterpret_cast<const i32x4*>(qb + aq_off + 8 * st);
            MELLOW_BF16(aq, kq[st], sacc);
        }
        // output lane (key kk, half hf'): rows 8 (r / 4) + 4 hf' + r % 4 -> piece 0 in r = head (hf' = 0), piece 1 in r = head (hf' = 1),
        // piece 2 in r = 4 + head (hf' = 0); rows 12 .. 14 (r = 4 + head, hf' = 1) are zero rows of A
        const int gi = mf_group(g0, kk >> 2), t = gi * 4 + (kk & 3);
        const bool ok = gi < gend && t < pos;
        float pw[3], alpha[3];
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            float sv = half_sum(sacc[hh] + sacc[4 + hh]);              // both halves of the wave hold the key's score
            sv = ok ? sv : -INFINITY;
            float cm = row16_max(sv);
            cm = fmaxf(cm, swz_xor16(cm));                            // max over the 32 keys of the tile (at least one is valid)
            const float m_new = fmaxf(m_run[hh], cm);
            alpha[hh] = fast_exp(m_run[hh] - m_new);
            m_run[hh] = m_new;
            pw[hh] = fast_exp(sv - m_new);
            l_run[hh] = l_run[hh] * alpha[hh] + pw[hh];               // per key lane; summed over the tile's lanes at the end
        }
        float4* plw = reinterpret_cast<float4*>(pl) + vw * 32;
        if (hf == 0) plw[kk] = make_float4(pw[0], pw[1], pw[2], 0.f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // wave-private hand-over: LDS operations of a wave complete in order
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) acc[hh] = make_float4(acc[hh].x * alpha[hh], acc[hh].y * alpha[hh], acc[hh].z * alpha[hh], acc[hh].w * alpha[hh]);
#pragma unroll
        for (int u = 0; u < DA_GM; ++u) {
            const float4 pp = plw[4 * u + sub];                        // the three heads' weights of key (u, sub): one address per 16 lanes
            const float4 v = make_float4(__uint_as_float(vq[u][0] << 16), __uint_as_float(vq[u][0] & 0xffff0000u),
                                         __uint_as_float(vq[u][1] << 16), __uint_as_float(vq[u][1] & 0xffff0000u));
            acc[0].x += pp.x * v.x; acc[0].y += pp.x * v.y; acc[0].z += pp.x * v.z; acc[0].w += pp.x * v.w;
            acc[1].x += pp.y * v.x; acc[1].y += pp.y * v.y; acc[1].z += pp.y * v.z; acc[1].w += pp.y * v.w;
            acc[2].x += pp.z * v.x; acc[2].y += pp.z * v.y; acc[2].z += pp.z * v.z; acc[2].w += pp.z * v.w;
        }
        asm volatile("" ::: "memory");                                 // the next chunk's weights are written after these reads
    };
    // reduce the 4 key-subs of the wave; publish (m, l, o) of the (virtual) wave vw
    auto publish = [&](int vw) {
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            float l = l_run[hh];                 // identical across the 16 quads of a sub; sum over the 4 subs
            if constexpr (MF) {                  // matrix form: one partial per key lane, the two halves of the wave hold the same 32
                l = row16_sum(l);
                l += swz_xor16(l);
            } else {
                l += swz_xor16(l);
                l = half_sum(l);
            }
            float4 o = acc[hh];
            o.x += swz_xor16(o.x); o.y += swz_xor16(o.y);
            o.z += swz_xor16(o.z); o.w += swz_xor16(o.w);
            o.x = half_sum(o.x); o.y = half_sum(o.y);
            o.z = half_sum(o.z); o.w = half_sum(o.w);
            if (sub == 0) *reinterpret_cast<float4*>(ored + (vw * 3 + hh) * 64 + quad * 4) = o;
            if (lane == 0) { mred[vw * 3 + hh] = m_run[hh]; lred[vw * 3 + hh] = l; }
        }
    };
    if constexpr (MF) {
        // virtual wave wave + NW j of this physical wave, one after the other on the same registers: its chunks
        // gbeg + vw + c * (DA_WAVES * DA_GM) below g_stop, then its partial; the first chunk of j = 0 was requested around the prologue
#pragma clang loop unroll(disable)
        for (int j = 0; j < VPW; ++j) {
            const int vw = wave + NW * j;
            if (j != 0) {
#pragma unroll
                for (int hh = 0; hh < 3; ++hh) { m_run[hh] = -INFINITY; l_run[hh] = 0.f; acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f); }
            }
#pragma clang loop unroll(disable)
            for (int g0 = gbeg + vw; g0 < g_stop; g0 += DA_WAVES * DA_GM) {
                if (g0 != g_first_mf) {          // every chunk but the first: load, then the same body
                    load_ktile(g0);
#pragma unroll
                    for (int u = 0; u < DA_GM; ++u) load_vgroup(g0, u);
                    __builtin_amdgcn_sched_barrier(0);
                }
                chunk_mf(vw, g0);
            }
            publish(vw);
        }
    } else if constexpr (ONE) {
        for (int g0 = g_first; g0 < g_stop; g0 += DA_WAVES * DA_G) {
            if (g0 != g_first) MELLOW_DA_RELOAD(g0)        // later chunks (only for contexts beyond 448 keys per split)
            MELLOW_DA_CHUNK(g0)
        }
    } else {
        if (g_first < g_stop) MELLOW_DA_CHUNK(g_first)
#pragma clang loop unroll(disable)
        for (int g0 = g_first + DA_WAVES * DA_G; g0 < g_stop; g0 += DA_WAVES * DA_G) {
            MELLOW_DA_RELOAD(g0)
            MELLOW_DA_CHUNK(g0)
        }
    }
#undef MELLOW_DA_CHUNK
#undef MELLOW_DA_RELOAD
    if constexpr (!MF) {
        // (the vector form keeps its own text: routing it through the lambda above changes the register allocation of the one-row-block
        //  kernel -- +128 v_mov, 7.9 -> 8.7 us per launch at B = 32, measured)
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            float l = l_run[hh];                 // identical across the 16 quads of a sub; sum over the 4 subs
            l += swz_xor16(l);
            l = half_sum(l);
            acc[hh].x += swz_xor16(acc[hh].x); acc[hh].y += swz_xor16(acc[hh].y);
            acc[hh].z += swz_xor16(acc[hh].z); acc[hh].w += swz_xor16(acc[hh].w);
            acc[hh].x = half_sum(acc[hh].x); acc[hh].y = half_sum(acc[hh].y);
            acc[hh].z = half_sum(acc[hh].z); acc[hh].w = half_sum(acc[hh].w);
            if (sub == 0) *reinterpret_cast<float4*>(ored + (wave * 3 + hh) * 64 + quad * 4) = acc[hh];
            if (lane == 0) { mred[wave * 3 + hh] = m_run[hh]; lred[wave * 3 + hh] = l; }
        }
    }
    kstamp(1, 4, dbg && l_run[0] == l_run[0]);
    __syncthreads();
    kstamp(1, 5, dbg);
    static_assert(DA_WAVES == 8 || DA_WAVES == 16 || DA_WAVES == 4, "the merge below reduces over 8-lane groups (8 waves) or serially");
    if (DA_WAVES == 8)
#pragma unroll
    for (int it = tid; it < 384; it += NW * 64) {      // (whole 8-lane groups: 384 and NW * 64 are multiples of 8)
        // item -> (pair = (head hh, dim quad dq), wave w): the 8 waves' partials of a pair sit in 8 adjacent lanes and are
        // combined with three DPP steps (xor 1, xor 2, mirror of the 8-lane half-row) instead of a serial loop in 48 threads
        const int w = it & 7, pair = it >> 3, hh = pair >> 4, dq = pair & 15;
        const float mw = mred[w * 3 + hh];
        const float snew = snew_s[hh];
        float M = mw;
        M = fmaxf(M, dpp_mov<0xB1>(M));
        M = fmaxf(M, dpp_mov<0x4E>(M));
        M = fmaxf(M, dpp_mov<0x141>(M));
        if (sp == TS - 1) M = fmaxf(M, snew);          // the last split also owns the new key
        const float f = M > -INFINITY ? fast_exp(mw - M) : 0.f;      // waves without keys: m = -inf -> factor 0; an empty split: all 0
        const float4 ow = *reinterpret_cast<const float4*>(ored + (w * 3 + hh) * 64 + dq * 4);
        float L = lred[w * 3 + hh] * f;
        float4 O = make_float4(ow.x * f, ow.y * f, ow.z * f, ow.w * f);
#define MELLOW_R8(V) V += dpp_mov<0xB1>(V); V += dpp_mov<0x4E>(V); V += dpp_mov<0x141>(V)
        MELLOW_R8(L); MELLOW_R8(O.x); MELLOW_R8(O.y); MELLOW_R8(O.z); MELLOW_R8(O.w);
#undef MELLOW_R8
        if (w == 0) {
            if (sp == TS - 1 && M > -INFINITY) {
                const float pn = fast_exp(snew - M);
                const float4 vn = *reinterpret_cast<const float4*>(vnew + dq * 4);
                L += pn;
                O.x += pn * vn.x; O.y += pn * vn.y; O.z += pn * vn.z; O.w += pn * vn.w;
            }
            // F16-layout (B operand of the o_proj's 16x16x4 MFMA): column k = head*64 + 4*dq
            const int head = 3 * g + hh, k = head * 64 + dq * 4;
            const int rb = b >> 5, m = b & 31;
            const int64_t o4 = ((((int64_t)sp * a.RB + rb) * 36 + (k >> 4)) * 2 + (m >> 4)) * 64 + (m & 15) + 16 * ((k >> 2) & 3);
            st_out(reinterpret_cast<float4*>(a.attF16) + o4, O);
            if (dq == 0)      // (m, l) of this split: [head][row][split] pairs, so that the o_proj reads both splits of a row with one 16-byte load
                *reinterpret_cast<float2*>(a.att_ml + (((int64_t)head * a.rows + b) * TS + sp) * 2) = make_float2(M, L);
        }
    }
    if (DA_WAVES != 8 && tid < 48) {
        // thread -> (head hh, dim quad dq): merged float4 of the waves (+ the new key on the last split), serial form
        const int hh = tid >> 4, dq = tid & 15;
        const float snew = snew_s[hh];
        float M = sp == TS - 1 ? snew : -INFINITY;   // the last split also owns the new key
#pragma unroll
        for (int w = 0; w < DA_WAVES; ++w) M = fmaxf(M, mred[w * 3 + hh]);
        float L = 0.f;
        float4 O = make_float4(0.f, 0.f, 0.f, 0.f);
        if (M > -INFINITY) {                 // an empty split (short context) publishes m = -inf, l = 0, o = 0
            if (sp == TS - 1) {
                const float pn = fast_exp(snew - M);
                const float4 vn = *reinterpret_cast<const float4*>(vnew + dq * 4);
                L = pn;
                O = make_float4(pn * vn.x, pn * vn.y, pn * vn.z, pn * vn.w);
            }
#pragma unroll
            for (int w = 0; w < DA_WAVES; ++w) {
                const float f = fast_exp(mred[w * 3 + hh] - M);     // waves without keys: m = -inf -> factor 0
                const float4 ow = *reinterpret_cast<const float4*>(ored + (w * 3 + hh) * 64 + dq * 4);
                L += lred[w * 3 + hh] * f;
                O.x += ow.x * f; O.y += ow.y * f; O.z += ow.z * f; O.w += ow.w * f;
            }
        }
        const int head = 3 * g + hh, k = head * 64 + dq * 4;
        const int rb = b >> 5, m = b & 31;
        const int64_t o4 = ((((int64_t)sp * a.RB + rb) * 36 + (k >> 4)) * 2 + (m >> 4)) * 64 + (m & 15) + 16 * ((k >> 2) & 3);
        reinterpret_cast<float4*>(a.attF16)[o4] = O;
        if (dq == 0)
            *reinterpret_cast<float2*>(a.att_ml + (((int64_t)head * a.rows + b) * TS + sp) * 2) = make_float2(M, L);
    }
    kstamp(1, 6, dbg);
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// K3  o_proj with complete output.  grid (36 n16-tiles, 2 row halves x RB), 16 waves; v_mfma_f32_16x16x4_f32,
//     W in P16-layout.  A workgroup owns 16 rows x 16 columns and the whole K = 576 (one MFMA tile), so its
//     output is complete: x_mid = x_new + merge(attention splits) Wo^T.  Splitting the 32 batch rows into two
//     halves doubles the workgroups (72) and halves the activation bytes each CU has to pull.
//     writes x_mid row-major + F32-layout (next layer's qkv) + F16-layout (gate/up) + per-tile sum of squares
// ----------------------------------------------------------------------------------------------------
#ifndef MELLOW_OPROJ_WAVES
#define MELLOW_OPROJ_WAVES 12
#endif
// waves per workgroup: 12 x 3 k16-tiles = the 36 tiles exactly (16 x 3 left 12 empty slots whose loads were still issued:
// +1.1 ms of decode per 63 steps, measured on one box with tools/ab_build.sh)
constexpr int OP_WAVES = MELLOW_OPROJ_WAVES;
// rows per workgroup: 16 (two halves of a 32-row block, 72 workgroups) or 8 (four quarters, 144 workgroups: the MFMA tile still
// has 16 batch columns, eight of them zero, but a workgroup pulls half the attention partials -- its body is bound by the
// bytes one CU has to ingest)
// Chosen per launch: 8 rows for a single row block (B <= 32: 52.8 -> 52.2 ms of decode per 63 steps), 16 rows otherwise (at
// B = 64 the 288 eight-row workgroups no longer fit the 256 CUs: 76.3 -> 78.5 ms).  MELLOW_OPROJ_ROWS forces one form.
template <bool BLK, int W8, int OP_ROWS, int TS>
__global__ __launch_bounds__(OP_WAVES * 64) void dec_oproj_kernel(const float* __restrict__ Wp16, const float* __restrict__ attF16_p,
                                                                  const float* __restrict__ att_ml_p, const float* __restrict__ xnewR_p,
                                                                  int RB_p, int rows_p, const DecArgs a,
                                                                  const float* __restrict__ wscale) {
    kspan(a.dbg_seq, 0);
    __shared__ __attribute__((aligned(16))) float red[OP_WAVES * 4 * 64];   // [wave][acc reg][lane]
    constexpr int K16 = 36, TPW = (K16 + OP_WAVES - 1) / OP_WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PARTS = 32 / OP_ROWS;                     // workgroups per 32-row block
    const int nt = blockIdx.x, rb = blockIdx.y / PARTS, part = blockIdx.y % PARTS;
    const int mh = OP_ROWS == 16 ? part : part >> 1;        // which 16-row half the MFMA tile covers
    MELLOW_BLK_EXIT(rb)
    const int ml = lane & 15;
    const bool lrow = OP_ROWS == 16 || (ml >> 3) == (part & 1);          // this lane's batch row belongs to the workgroup
    const int64_t wslot = (int64_t)nt * K16 * 64 + lane;
    // epilogue operand issued up front: thread (m, nq) of the 16 x 16 tile owns 4 consecutive columns
    const int em = (tid >> 2) & 15, enq = tid & 3;
    const bool erow_ok = OP_ROWS == 16 || (em >> 3) == (part & 1);
    const int64_t erow = (int64_t)rb * 32 + mh * 16 + em;
    float4 xres = make_float4(0.f, 0.f, 0.f, 0.f);
    if (erow_ok) xres = *reinterpret_cast<const float4*>(xnewR_p + erow * 576 + nt * 16 + enq * 4);

    float4 w[TPW], os[TPW][TS];
    uint32_t w8[TPW];
    float ms[TPW][TS], ls[TPW][TS];
    const int64_t row = (int64_t)rb * 32 + mh * 16 + ml;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + OP_WAVES * i, tc = t < K16 ? t : K16 - 1;     // clamped: out-of-range tiles get zero weights
        if constexpr (W8 == 2) {
            w8[i] = ldw8(Wp16, wslot + (int64_t)tc * 64);
            if (t >= K16) w8[i] = 0u;
        } else {
            w[i] = ldw<W8 != 0>(Wp16, wslot + (int64_t)tc * 64);
            if (t >= K16) w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int h = tc >> 2;                                        // tile = 16 k of head h
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            ms[i][s] = 0.f; ls[i][s] = 1.f; os[i][s] = make_float4(0.f, 0.f, 0.f, 0.f);     // rows of another workgroup: x = 0
            if (lrow) os[i][s] = reinterpret_cast<const float4*>(attF16_p)[((((int64_t)s * RB_p + rb) * 36 + tc) * 2 + mh) * 64 + lane];
        }
        if (lrow) {
            const float* mlp = att_ml_p + ((int64_t)h * rows_p + row) * TS * 2;
            if constexpr (TS == 2) {                     // (m, l) of both splits of the row: one 16-byte load
                const float4 v = *reinterpret_cast<const float4*>(mlp);
                ms[i][0] = v.x; ls[i][0] = v.y; ms[i][1] = v.z; ls[i][1] = v.w;
            } else if constexpr (TS == 1) {
                const float2 v = *reinterpret_cast<const float2*>(mlp);
                ms[i][0] = v.x; ls[i][0] = v.y;
            } else {
#pragma unroll
                for (int s = 0; s < TS; ++s) { ms[i][s] = mlp[2 * s]; ls[i][s] = mlp[2 * s + 1]; }
            }
        }
    }
    MELLOW_HOIST(a.xmidF); MELLOW_HOIST(a.xmidF16); MELLOW_HOIST(a.ssq);
    if (W8) MELLOW_HOIST(wscale);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        // merge the key splits: x = sum_s f_s o_s / sum_s f_s l_s, f_s = exp(m_s - max m)
        float M = ms[i][0];
#pragma unroll
        for (int s = 1; s < TS; ++s) M = fmaxf(M, ms[i][s]);
        float L = 0.f;
        float4 O = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const float f = fast_exp(ms[i][s] - M);
            L += ls[i][s] * f;
            O.x += os[i][s].x * f; O.y += os[i][s].y * f; O.z += os[i][s].z * f; O.w += os[i][s].w * f;
        }
        const float inv = 1.0f / L;
        if constexpr (W8 == 2) {
            os[i][0] = make_float4(O.x * inv, O.y * inv, O.z * inv, O.w * inv);      // the merged activation, quantised below
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, O.x * inv, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, O.y * inv, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, O.z * inv, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, O.w * inv, acc, 0, 0, 0);
        }
    }
    if constexpr (W8 == 2) {
        // a batch row's values of this wave's k-tiles sit in the four lanes ml + 16 q
        float am = 0.f, inv, sc;
#pragma unroll
        for (int i = 0; i < TPW; ++i) am = fmaxf(am, f4amax(os[i][0]));
        am = fmaxf(am, swz_xor16(am));
        a8_scales(half_max(am), inv, sc);
        uint32_t xq[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) xq[i] = q4_e4m3(os[i][0], inv);
        acc = mfma8_16<TPW>(acc, w8, xq);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= sc;
    }
    // D[i = n_local = 4*(lane>>4) + r][j = m_local = lane&15]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (tid < 64 && erow_ok) {
        // columns n = 4*enq + r live in registers r = 0..3 of lane em + 16*enq
        const int src_lane = em + 16 * enq;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wv = 0; wv < OP_WAVES; ++wv)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += red[(wv * 4 + r) * 64 + src_lane];
        if (W8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= wscale[nt * 16 + enq * 4 + r];
        }
        const float4 y = make_float4(xres.x + v[0], xres.y + v[1], xres.z + v[2], xres.w + v[3]);
        const int k = nt * 16 + enq * 4;
        st_out(reinterpret_cast<float4*>(a.xmidF) + f32_idx(rb, 72, mh * 16 + em, k), y);
        if (a.xmid3_32) {                  // f32x3 layer kernels: x_mid pre-split for the q/k/v part (F3-32) and for gate/up (F3-16)
            const F3Quad yq = f3_split4(y), yo = f3_partner(yq);      // threads (enq, enq ^ 1) hold the halves of one slot
            if (!(enq & 1)) {
                f3_store8(a.xmid3_32, f3_32_slot(rb, 36, mh * 16 + em, k), yq, yo);
                f3_store8(a.xmid3_16, f3_16_slot(rb, 18, mh * 16 + em, k), yq, yo);
            }
        } else
        st_out(reinterpret_cast<float4*>(a.xmidF16) + (((int64_t)rb * 36 + nt) * 2 + mh) * 64 + em + 16 * enq, y);
        float ss = f4ssq(y);
        ss += dpp_mov<0xB1>(ss);             // sum over the quad (the 4 column groups of one row)
        ss += dpp_mov<0x4E>(ss);
        if (enq == 0) a.ssq[erow * 40 + nt] = ss;
    }
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// (measured and not kept: two 32-column tiles per workgroup sharing the x registers, 768 workgroups: 38.9 vs 36.7 us)
// K4  full-K projection (lm_head).  grid (n-tiles, 1, RB), 4 waves x 18 k-tiles, all 36 loads of a
//     wave in flight; X in F32-layout.  Writes row-major logits (optional) + fused arg-max candidates per tile.
// ----------------------------------------------------------------------------------------------------
enum { OUT_LOGITS = 1 };
#ifndef MELLOW_LM_WAVES
#define MELLOW_LM_WAVES 8     // 8 x 9 k-tiles: 53.45 vs 54.35 ms of decode per 63 steps with 4 x 18 (6 / 9 / 12 waves: 54.1 / 54.1 / 53.7)
#endif
constexpr int LM_WAVES = MELLOW_LM_WAVES;
template <int OUT, bool BLK, int W8>
__global__ __launch_bounds__(LM_WAVES * 64) void dec_fullk_kernel(const float* __restrict__ Wp, const float* __restrict__ XF, int K8p,
                                                        int N, const DecArgs a, const float* __restrict__ wscale) {
    kspan(a.dbg_seq, 0);
    __shared__ __attribute__((aligned(16))) float red[LM_WAVES * 16 * 64];
    constexpr int KPW = 72 / LM_WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, rb = blockIdx.z;
    MELLOW_BLK_EXIT(rb)
    const int k8_0 = wave * KPW;
    const int64_t wslot = ((int64_t)nt * K8p + k8_0) * 64 + lane;
    const float4* xp = reinterpret_cast<const float4*>(XF) + ((int64_t)rb * 72 + k8_0) * 64 + lane;
    float4 w[KPW], x[KPW];
    uint32_t w8[KPW];
    const bool dbg = tid == 0 && nt == 0 && rb == 0;
    const int dslot = 5;
    kstamp(dslot, 0, dbg);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        if constexpr (W8 == 2) w8[i] = ldw8(Wp, wslot + i * 64);
        else w[i] = ldw<W8 != 0>(Wp, wslot + i * 64);
        x[i] = xp[i * 64];
    }
    MELLOW_HOIST(a.logits); MELLOW_HOIST(a.cand_val); MELLOW_HOIST(a.cand_idx); MELLOW_HOIST(N);
    if (W8) MELLOW_HOIST(wscale);
    __builtin_amdgcn_sched_barrier(0);
    kstamp(dslot, 1, dbg);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (W8 == 2) {
        float am = 0.f, inv, sc;
#pragma unroll
        for (int i = 0; i < KPW; ++i) am = fmaxf(am, f4amax(x[i]));
        a8_scales(half_max(am), inv, sc);
        uint32_t xq[KPW];
#pragma unroll
        for (int i = 0; i < KPW; ++i) xq[i] = q4_e4m3(x[i], inv);
        acc = mfma8_32<KPW>(acc, w8, xq);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= sc;
    } else {
#pragma unroll
        for (int i = 0; i < KPW; ++i) acc = mfma4(acc, w[i], x[i]);
    }
    kstamp(dslot, 2, dbg && acc[0] == acc[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    kstamp(dslot, 3, dbg);
    const bool epi = tid < 256;             // the epilogue is 256 threads wide (more waves only with MELLOW_LM_WAVES > 4)
    const int mm = tid & 31, hh = (tid >> 5) & 1, gq = (tid >> 6) & 3;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * gq + j;
        float sacc = red[r * 64 + mm + 32 * hh];
#pragma unroll
        for (int wv = 1; wv < LM_WAVES; ++wv) sacc += red[(wv * 16 + r) * 64 + mm + 32 * hh];      // fixed order
        v[j] = sacc;
    }
    {
        const int n = nt * 32 + 8 * gq + 4 * hh;
        if (W8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= wscale[n + j];
        }
        const int64_t row = (int64_t)rb * 32 + mm;
        if (epi && a.logits && n < N) *reinterpret_cast<float4*>(a.logits + row * N + n) = make_float4(v[0], v[1], v[2], v[3]);
        // best (value, lowest index) of this 32-column tile per row (torch.argmax tie rule)
        __syncthreads();
        float bv = v[0];
        int bi = n;
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (arg_better(v[j], n + j, bv, bi)) { bv = v[j]; bi = n + j; }
        if (epi) {
            red[tid] = bv;
            reinterpret_cast<int*>(red)[256 + tid] = bi;
        }
        __syncthreads();
        if (tid < 32) {
            float best = red[tid];
            int idx = reinterpret_cast<int*>(red)[256 + tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const float ov = red[tid + 32 * q];
                const int oi = reinterpret_cast<int*>(red)[256 + tid + 32 * q];
                if (arg_better(ov, oi, best, idx)) { best = ov; idx = oi; }
            }
            const int64_t o = ((int64_t)rb * 32 + tid) * (N >> 5) + nt;          // N / 32 = gridDim.x (a dispatch-packet read in the tail otherwise)
            a.cand_val[o] = best;
            a.cand_idx[o] = idx;
        }
    }
    kspan(a.dbg_seq, 1);
}


// ----------------------------------------------------------------------------------------------------
// K4y  lm_head, f32x3 form, as a streaming GEMM: one WAVE owns a 32-row n-tile over the WHOLE K = 576 (no split-K, no LDS
//      reduction), a workgroup = H3_NW waves = H3_NW consecutive n-tiles, and the pre-split activations (F3-32, written by the
//      final norm) of up to G row blocks sit in LDS, shared by the waves.  The weights are
//      streamed ONCE per step whatever the batch (the fp32 kernel above re-reads them per row block and every 32-row tile
//      re-reads x: at B = 128 that is 4 x 113 MB of weights and 0.45 GB of activations through the L2s), split to bf16 triples
//      in registers once per chunk and used for all G row blocks.  Rows beyond G blocks: further passes.
//      The arg-max candidates of a tile are formed from the accumulators directly (a wave holds all 32 columns of its rows).
// ----------------------------------------------------------------------------------------------------
#ifndef MELLOW_H3_NW
#define MELLOW_H3_NW 8
#endif
#ifndef MELLOW_H3R_D1
#define MELLOW_H3R_D1 1           // weight chunks in flight per wave at 1 / 2 / >= 3 row blocks (2 and 3 measured slower)
#endif
#ifndef MELLOW_H3R_D2
#define MELLOW_H3R_D2 1
#endif
#ifndef MELLOW_H3R_D4
#define MELLOW_H3R_D4 1
#endif
constexpr int H3_NW = MELLOW_H3_NW, H3_CP = 3, H3_NC = 36 / H3_CP;
//      Activations RESIDENT in LDS: the G row blocks' fragments of 36 / G k-pairs fill the 108 KiB stage once per phase (G
//      phases), and inside a phase the waves run without any barrier -- each streams its own n-tile's weights D chunks ahead
//      (D = 1: deeper register prefetch measured slower) and reads the fragments it needs from LDS.  (A first form refilled a small
//      stage every chunk -- one barrier + one L2 round trip per chunk: tools/experiments/r05_dec_head3_kernel.hip.txt.)
template <int G, int D, bool BLK>
__global__ __launch_bounds__(H3_NW * 64) void dec_head3r_kernel(const float* __restrict__ Wp, const i32x4* __restrict__ X3, int K8p, int N,
                                                                int RB_p, const DecArgs a) {
    kspan(a.dbg_seq, 0);
    extern __shared__ __attribute__((aligned(16))) i32x4 xs[];        // [G][PPH][3][64]
    constexpr int PPH = 36 / G, NCH = PPH / H3_CP;                     // pairs / chunks per phase
    constexpr int STAGE = G * PPH * 3 * 64;                            // = 36 * 192 slots = 108 KiB
    static_assert(PPH * G == 36 && NCH * H3_CP == PPH && H3_NC % D == 0 && NCH % D == 0, "phases hold whole chunks");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x * H3_NW + wave;
    const float4* wbase = reinterpret_cast<const float4*>(Wp) + (int64_t)nt * K8p * 64 + lane;
    for (int rb0 = 0; rb0 < RB_p; rb0 += G) {
        const int gn = min(G, RB_p - rb0);
        f32x16 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        float4 wn[D][2 * H3_CP];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < 2 * H3_CP; ++i) wn[d][i] = ldg_nt(wbase + (d * 2 * H3_CP + i) * 64);
        for (int ph = 0; ph < G; ++ph) {
            if (ph || rb0) __syncthreads();               // every wave has read the previous phase's fragments
            // fill: slot i = (g, pair pp of the phase, piece, lane) <- X3[((rb0 + g) * 36 + ph * PPH + pp) * 3 + piece][lane]
            {
                constexpr int FILL = (STAGE + H3_NW * 64 - 1) / (H3_NW * 64);      // all loads of a thread in flight together
                i32x4 t[FILL];
#pragma unroll
                for (int j = 0; j < FILL; ++j) {
                    const int i = tid + j * H3_NW * 64;
                    if (i < STAGE) {
                        const int g = i / (PPH * 3 * 64), rest = i % (PPH * 3 * 64);
                        t[j] = X3[(((int64_t)(rb0 + (g < gn ? g : 0)) * 36 + ph * PPH) * 3) * 64 + rest];
                    }
                }
#pragma unroll
                for (int j = 0; j < FILL; ++j) {
                    const int i = tid + j * H3_NW * 64;
                    if (i < STAGE) xs[i] = t[j];
                }
            }
            __syncthreads();
            for (int c0 = 0; c0 < NCH; c0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int cc = c0 + d, c = ph * NCH + cc;      // chunk of the phase / of the whole K
                    float4 w[2 * H3_CP];
#pragma unroll
                    for (int i = 0; i < 2 * H3_CP; ++i) w[i] = wn[d][i];
                    if (c + D < H3_NC) {
#pragma unroll
                        for (int i = 0; i < 2 * H3_CP; ++i) wn[d][i] = ldg_nt(wbase + ((c + D) * 2 * H3_CP + i) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    i32x4 wp[H3_CP][3];
#pragma unroll
                    for (int p = 0; p < H3_CP; ++p) {
                        pair_natural(w[2 * p], w[2 * p + 1]);
                        split_pair(w[2 * p], w[2 * p + 1], wp[p][0], wp[p][1], wp[p][2]);
                    }
                    const i32x4* st = xs + (cc * H3_CP) * 3 * 64 + lane;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (g < gn) {
#pragma unroll
                            for (int p = 0; p < H3_CP; ++p) {
                                const i32x4 xq[3] = {st[((g * PPH + p) * 3 + 0) * 64], st[((g * PPH + p) * 3 + 1) * 64], st[((g * PPH + p) * 3 + 2) * 64]};
                                acc[g] = mma6(acc[g], wp[p], xq);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const int m = lane & 31, h = lane >> 5;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g < gn && !(BLK && a.blk_live[rb0 + g] == 0)) {
                const int64_t row = (int64_t)(rb0 + g) * 32 + m;
                float bv = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nt * 32 + 8 * q + 4 * h;
                    if (a.logits) *reinterpret_cast<float4*>(a.logits + row * N + n) = make_float4(acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (arg_better(acc[g][4 * q + j], n + j, bv, bi)) { bv = acc[g][4 * q + j]; bi = n + j; }
                }
                auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
                auto ri = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
                const float ov = __uint_as_float(h ? rv[0] : rv[1]);
                const int oi = (int)(h ? ri[0] : ri[1]);
                if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                if (h == 0) {
                    const int64_t o = row * (N >> 5) + nt;
                    a.cand_val[o] = bv;
                    a.cand_idx[o] = bi;
                }
            }
        }
    }
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// f32x3 forms of the two big GEMM launches of a decode layer, for batches of MORE than one 32-row block (north_star's batch 64,
// BASELINE configs[3] / [4]): the same workgroup -> weight-tile mapping, slab outputs and consumers as dec_qkv2_kernel /
// dec_gateup16_kernel, but
//   * ONE workgroup serves every row block: its weight fragments are loaded once and split to bf16 triples once (the fp32
//     kernels replicate the grid per row block: weights re-read, fp32 MFMA time per block),
//   * the activations arrive pre-split from their producer (F3 layouts above: o_proj writes x_mid, gate/up writes h), so a
//     row block costs 3 x 16-byte loads and six bf16 MFMAs per k-pair and no VALU,
//   * RBM row blocks are accumulated in registers per pass, one LDS reduction across the waves per pass.
// ----------------------------------------------------------------------------------------------------
#define MELLOW_BF16S(W, X, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, X), ACC, 0, 0, 0)
__device__ __forceinline__ f32x4 mma6s(f32x4 acc, const i32x4 (&w)[3], const i32x4 (&x)[3]) {
    MELLOW_BF16S(w[2], x[0], acc);
    MELLOW_BF16S(w[0], x[2], acc);
    MELLOW_BF16S(w[1], x[1], acc);
    MELLOW_BF16S(w[1], x[0], acc);
    MELLOW_BF16S(w[0], x[1], acc);
    MELLOW_BF16S(w[0], x[0], acc);
    return acc;
}
// One wave's share of a 32-row weight tile: NP k-pairs of W (fp32, P-layout, tiles k8_0 ..) against the same pairs of every
// row block's F3-32 activations.  acc[g] += W . x[rb0 + g] for g < gn.
template <int NP, int NPA>
__device__ __forceinline__ void x3_load(i32x4 (&xq)[NPA][3], const i32x4* __restrict__ x3, int64_t off) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) xq[p][c] = x3[off + (p * 3 + c) * 64];
}
// xq[0] holds the fragments of block rb0 on entry (requested by the caller together with the weights)
template <int NP, int RBM, int NPA>
__device__ __forceinline__ void x3_tile_job(const i32x4 (&wp)[NPA][3], i32x4 (&xq)[2][NPA][3], const i32x4* __restrict__ x3 /* (rb 0, pair T0, piece 0, lane) */,
                                            int64_t rb_stride, int rb0, int gn, f32x16 (&acc)[RBM]) {
#pragma unroll
    for (int g = 0; g < RBM; ++g) {
        if (g < gn) {
            if (g + 1 < RBM && g + 1 < gn) x3_load<NP, NPA>(xq[(g + 1) & 1], x3, (int64_t)(rb0 + g + 1) * rb_stride);
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[g] = mma6(acc[g], wp[p], xq[g & 1][p]);
        }
    }
}
#ifndef MELLOW_Q3_WAVES
#define MELLOW_Q3_WAVES 6
#endif
constexpr int Q3W = MELLOW_Q3_WAVES;
// K5+K1 (f32x3, any number of row blocks): see dec_qkv2_kernel for the algebra and the workgroup types.  xmid3 = x_mid in F3-32
// (36 pairs per row block), h3 = h in F3-32 (96 pairs).
// One workgroup's job for either type (NP k-pairs per wave): weights + the first block's fragments requested together, weights
// split once, RBM row blocks accumulated per pass, one LDS reduction per pass, slabs out.
template <int NP, int RBM, bool BLK>
__device__ __forceinline__ void q3_body(const float4* __restrict__ wp4, const i32x4* __restrict__ xsrc, int64_t rb_stride, int nt, int slab,
                                        bool side, int RB_p, const DecArgs& a, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    i32x4 wp[NP][3];
    i32x4 xq[2][NP][3];
    constexpr bool ALL = RBM <= 2;            // a pass's fragments fit the registers: request all of them at once (no dependent second round trip)
    {
        float4 w[2 * NP];
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i) w[i] = ldg_nt(wp4 + i * 64);
        x3_load<NP, NP>(xq[0], xsrc, 0);
        if (ALL && RBM == 2 && RB_p > 1) x3_load<NP, NP>(xq[1], xsrc, rb_stride);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            pair_natural(w[2 * p], w[2 * p + 1]);
            split_pair(w[2 * p], w[2 * p + 1], wp[p][0], wp[p][1], wp[p][2]);
        }
    }
    for (int rb0 = 0; rb0 < RB_p; rb0 += RBM) {
        const int gn = min(RBM, RB_p - rb0);
        f32x16 acc[RBM];
#pragma unroll
        for (int g = 0; g < RBM; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        if (rb0) {
            x3_load<NP, NP>(xq[0], xsrc, (int64_t)rb0 * rb_stride);
            if (ALL && RBM == 2 && gn > 1) x3_load<NP, NP>(xq[1], xsrc, (int64_t)(rb0 + 1) * rb_stride);
        }
        if constexpr (ALL && RBM == 2) {
            if (gn > 1) {               // two row blocks = two independent accumulator chains, interleaved term by term
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#define MELLOW_Q3_TERM(PW, PX) MELLOW_BF16(wp[p][PW], xq[0][p][PX], acc[0]); MELLOW_BF16(wp[p][PW], xq[1][p][PX], acc[1]);
                    MELLOW_Q3_TERM(2, 0) MELLOW_Q3_TERM(0, 2) MELLOW_Q3_TERM(1, 1) MELLOW_Q3_TERM(1, 0) MELLOW_Q3_TERM(0, 1) MELLOW_Q3_TERM(0, 0)
#undef MELLOW_Q3_TERM
                }
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[0] = mma6(acc[0], wp[p], xq[0][p]);
            }
        } else if constexpr (ALL) {
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[0] = mma6(acc[0], wp[p], xq[0][p]);
        } else {
            x3_tile_job<NP, RBM, NP>(wp, xq, xsrc, rb_stride, rb0, gn, acc);
        }
        if (rb0) __syncthreads();                 // the previous pass's epilogue has read `red`
#pragma unroll
        for (int g = 0; g < RBM; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * RBM + g) * 16 + r) * 64 + lane] = acc[g][r];
        __syncthreads();
        if (tid < 256) {
            const int mm = tid & 31, hh = (tid >> 5) & 1, gq = tid >> 6;
            const int n = nt * 32 + 8 * gq + 4 * hh;
            for (int g = 0; g < gn; ++g) {
                const int rb = rb0 + g;
                if (BLK && a.blk_live[rb] == 0) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 4 * gq + j;
                    float sacc = red[((0 * RBM + g) * 16 + q) * 64 + mm + 32 * hh];
#pragma unroll
                    for (int wv = 1; wv < Q3W; ++wv) sacc += red[((wv * RBM + g) * 16 + q) * 64 + mm + 32 * hh];      // fixed order
                    v[j] = sacc;
                }
                const float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (side) st_out(reinterpret_cast<float4*>(a.dslabF) + (int64_t)slab * a.slabF_stride4 + f32_idx(rb, 72, mm, n), o);
                else st_out(reinterpret_cast<float4*>(a.pq + ((int64_t)slab * a.rows + rb * 32 + mm) * 960 + n), o);
            }
        }
    }
}
template <bool BLK, int RBM>
__global__ __launch_bounds__(Q3W * 64) void dec_qkv2x3_kernel(const float* __restrict__ Wx, const float* __restrict__ Wh,
                                                              const float* __restrict__ Wd, const i32x4* __restrict__ xmid3,
                                                              const i32x4* __restrict__ h3, int K8x, int K8h, int RB_p, const DecArgs a) {
    kspan(a.dbg_seq, 0);
    extern __shared__ __attribute__((aligned(16))) float red_dyn[];        // [Q3W][RBM][16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    constexpr int XP = 18 / Q3W, HP = (96 / Q2_HC) / Q3W;                    // k-pairs per wave
    static_assert(XP * Q3W == 18 && HP * Q3W * Q2_HC == 96 && Q3W * 64 >= 256, "waves must divide the k-chunks into whole pairs; the epilogue is 256 threads wide");
    if (b < 60) {              // x part (workgroup-uniform branch; each side is its own straight-line body)
        const int nt = b % 30, slab = b / 30;
        const int T0 = slab * 18 + wave * XP;
        q3_body<XP, RBM, BLK>(reinterpret_cast<const float4*>(Wx) + ((int64_t)nt * K8x + 2 * T0) * 64 + lane, xmid3 + (int64_t)T0 * 3 * 64 + lane,
                              (int64_t)36 * 3 * 64, nt, slab, false, RB_p, a, red_dyn);
    } else {                   // h part
        const int idx = b - 60, hc = idx / 48, nt = idx % 48;
        const bool side = nt >= 30;
        const int T0 = hc * (96 / Q2_HC) + wave * HP;
        const float4* wp4 = side ? reinterpret_cast<const float4*>(Wd) + ((int64_t)(nt - 30) * 192 + 2 * T0) * 64 + lane
                                 : reinterpret_cast<const float4*>(Wh) + ((int64_t)nt * K8h + 2 * T0) * 64 + lane;
        q3_body<HP, RBM, BLK>(wp4, h3 + (int64_t)T0 * 3 * 64 + lane, (int64_t)96 * 3 * 64, side ? nt - 30 : nt, side ? hc : 2 + hc, side, RB_p, a, red_dyn);
    }
    kspan(a.dbg_seq, 1);
}

#ifndef MELLOW_GU3_WAVES
#define MELLOW_GU3_WAVES 6
#endif
#ifndef MELLOW_G3_ABL
#define MELLOW_G3_ABL 0       // developer ablation (wrong results, timing only): 1 no pre-split store of h, 2 no MFMAs, 4 no activation loads, 8 no weight split, 16 no fp32 store of h
#endif
constexpr int GU3W = MELLOW_GU3_WAVES;
// K4b (f32x3, any number of row blocks): see dec_gateup16_kernel.  x3 = x_mid in F3-16 (18 pairs x 2 row halves per block).
// Writes h both as fp32 (guF: the last layer's fp32 down projection reads it) and pre-split (h3, F3-32 with 96 pairs).
template <bool BLK, int RBM>
__global__ __launch_bounds__(GU3W * 64) void dec_gateup3_kernel(const float* __restrict__ Wp16, const i32x4* __restrict__ x3,
                                                                const float* __restrict__ ssq_in, int RB_p, const DecArgs a) {
    kspan(a.dbg_seq, 0);
    extern __shared__ __attribute__((aligned(16))) float red_dyn[];        // [GU3W][RBM][8][64]
    float* red = red_dyn;
    constexpr int NP = 18 / GU3W;
    static_assert(NP * GU3W == 18, "waves must divide the 18 k-pairs");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x;
    const int T0 = wave * NP;
    i32x4 wp[NP][3];
    i32x4 xq[2][NP][2][3];
    const i32x4* xsrc = x3 + (int64_t)T0 * 2 * 3 * 64 + lane;
    constexpr int64_t RBS = (int64_t)18 * 2 * 3 * 64;          // slots per row block
    {
        // P16N: [n16-tile][pair T][half][lane][4 floats]: lane (n, q) holds k = 32 T + 8 q + 4 half .. + 3 (1 KiB per wave load)
        const float4* wp4 = reinterpret_cast<const float4*>(Wp16) + ((int64_t)nt * 18 + T0) * 128 + lane;
        float4 w[2 * NP];
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i) w[i] = ldg_nt(wp4 + i * 64);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int c = 0; c < 6; ++c) xq[0][p][c / 3][c % 3] = (MELLOW_G3_ABL & 4) ? i32x4{c, p, 1, 2} : xsrc[(p * 6 + c) * 64];          // block 0, requested with the weights
        if (RBM == 2 && RB_p > 1) {              // a two-block pass: both blocks' fragments at once
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int c = 0; c < 6; ++c) xq[1][p][c / 3][c % 3] = (MELLOW_G3_ABL & 4) ? i32x4{c, p, 3, 2} : xsrc[RBS + (p * 6 + c) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (MELLOW_G3_ABL & 8) { wp[p][0] = __builtin_bit_cast(i32x4, w[2 * p]); wp[p][1] = __builtin_bit_cast(i32x4, w[2 * p + 1]); wp[p][2] = wp[p][0]; }
            else split_pair(w[2 * p], w[2 * p + 1], wp[p][0], wp[p][1], wp[p][2]);
        }
    }
    for (int rb0 = 0; rb0 < RB_p; rb0 += RBM) {
        const int gn = min(RBM, RB_p - rb0);
        // the epilogue's RMS statistic (36 partial sums per row, written by the o_proj): requested with the operands, not after
        // the reduction barrier.  Epilogue thread e = tid (+ a second round when gn * 64 > the block): row block e / 64, row (e % 64) / 2
        float4 s4[2][9];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * GU3W * 64;
            if (e < gn * 64 && (k == 0 || RBM * 64 > GU3W * 64)) {
                const float4* sq = reinterpret_cast<const float4*>(ssq_in + ((int64_t)(rb0 + (e >> 6)) * 32 + ((e & 63) >> 1)) * 40);
#pragma unroll
                for (int j = 0; j < 9; ++j) s4[k][j] = sq[j];
            }
        }
        f32x4 acc0[RBM], acc1[RBM];
        if (rb0) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int c = 0; c < 6; ++c) xq[0][p][c / 3][c % 3] = xsrc[(int64_t)rb0 * RBS + (p * 6 + c) * 64];
            if (RBM == 2 && gn > 1) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int c = 0; c < 6; ++c) xq[1][p][c / 3][c % 3] = xsrc[(int64_t)(rb0 + 1) * RBS + (p * 6 + c) * 64];
            }
        }
#pragma unroll
        for (int g = 0; g < RBM; ++g) {
            acc0[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g < gn) {
                if (RBM > 2 && g + 1 < RBM && g + 1 < gn) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int c = 0; c < 6; ++c) xq[(g + 1) & 1][p][c / 3][c % 3] = xsrc[(int64_t)(rb0 + g + 1) * RBS + (p * 6 + c) * 64];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (MELLOW_G3_ABL & 2) { acc0[g][0] += __int_as_float(wp[p][0][0] ^ xq[g & 1][p][0][0][0]); acc1[g][1] += __int_as_float(wp[p][2][1] ^ xq[g & 1][p][1][2][1]); continue; }
                    // the two row halves are independent accumulator chains: term by term, alternating (a dependent MFMA waits ~2x its issue time)
#define MELLOW_G3_TERM(PW, PX) MELLOW_BF16S(wp[p][PW], xq[g & 1][p][0][PX], acc0[g]); MELLOW_BF16S(wp[p][PW], xq[g & 1][p][1][PX], acc1[g]);
                    MELLOW_G3_TERM(2, 0) MELLOW_G3_TERM(0, 2) MELLOW_G3_TERM(1, 1) MELLOW_G3_TERM(1, 0) MELLOW_G3_TERM(0, 1) MELLOW_G3_TERM(0, 0)
#undef MELLOW_G3_TERM
                }
            }
        }
        if (rb0) __syncthreads();
#pragma unroll
        for (int g = 0; g < RBM; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[((wave * RBM + g) * 8 + r) * 64 + lane] = acc0[g][r];
                red[((wave * RBM + g) * 8 + 4 + r) * 64 + lane] = acc1[g][r];
            }
        __syncthreads();
        // thread (g, m, q): as the epilogue of dec_gateup16_kernel, 64 threads per row block
        static_assert(RBM * 64 <= 2 * GU3W * 64, "two epilogue rounds cover a pass");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * GU3W * 64;
            if (e >= gn * 64 || (k == 1 && RBM * 64 <= GU3W * 64)) continue;
            const int g = e >> 6, t = e & 63, rb = rb0 + g;
            if (BLK && a.blk_live[rb] == 0) continue;
            const int m = t >> 1, q = t & 1;
            const int gl = (m & 15) + 16 * q, ul = gl + 32, rbase = (m >> 4) * 4;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) ss += (s4[k][j].x + s4[k][j].y) + (s4[k][j].z + s4[k][j].w);
            const float r2 = 1.0f / sqrtf(ss / 576.0f + a.eps);
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gv = red[((0 * RBM + g) * 8 + rbase + r) * 64 + gl], uv = red[((0 * RBM + g) * 8 + rbase + r) * 64 + ul];
#pragma unroll
                for (int wv = 1; wv < GU3W; ++wv) {                       // fixed order
                    gv += red[((wv * RBM + g) * 8 + rbase + r) * 64 + gl];
                    uv += red[((wv * RBM + g) * 8 + rbase + r) * 64 + ul];
                }
                h[r] = __fmul_rn(siluf_(gv * r2), uv * r2);
            }
            const float4 hv = make_float4(h[0], h[1], h[2], h[3]);
            if (!(MELLOW_G3_ABL & 16)) st_out(reinterpret_cast<float4*>(a.guF) + ((int64_t)rb * 192 + nt) * 64 + m + 32 * q, hv);
            const F3Quad hq = f3_split4(hv), ho = f3_partner(hq);     // threads (q = 0, q = 1) of a row hold the halves of one slot
            if (!(MELLOW_G3_ABL & 1) && q == 0) f3_store8(a.h3, f3_32_slot(rb, 96, m, 8 * nt), hq, ho);
        }
    }
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// K4b gate/up projection + SwiGLU on 16-row weight tiles.  grid (192 n16-tiles, RB), 4 waves x 9 k16-tiles,
//     v_mfma_f32_16x16x4_f32; W = folded gate/up in P16-layout, tile t = gate[8t..8t+7] | up[8t..8t+7];
//     X = x_mid in F16-layout.  192 workgroups x (36 KB of W + 72 KB of X).
//     epilogue: r2[m] = rsqrt(mean(x_mid[m]^2) + eps) from the o_proj's per-tile sums;
//     h = silu(r2 g) * (r2 u)  ->  hF[rb][hidden/8][lane][4]  (F32-layout B operand of the down projection)
// ----------------------------------------------------------------------------------------------------
#ifndef MELLOW_GU_WAVES
#define MELLOW_GU_WAVES 4
#endif
constexpr int GU_WAVES = MELLOW_GU_WAVES;
template <bool BLK, int W8>
__global__ __launch_bounds__(GU_WAVES * 64) void dec_gateup16_kernel(const float* __restrict__ Wp16, const float* __restrict__ xF16,
                                                                     const float* __restrict__ ssq_in, const DecArgs a,
                                                                     const float* __restrict__ wscale) {
    kspan(a.dbg_seq, 0);
    __shared__ __attribute__((aligned(16))) float red[GU_WAVES * 8 * 64];
    constexpr int K16 = 36, TPW = K16 / GU_WAVES;
    static_assert(TPW * GU_WAVES == K16, "waves must divide the 36 k16-tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, rb = blockIdx.y;
    MELLOW_BLK_EXIT(rb)
    const int t0 = wave * TPW;
    const int64_t wslot = ((int64_t)nt * K16 + t0) * 64 + lane;
    const float4* xp = reinterpret_cast<const float4*>(xF16) + (((int64_t)rb * 36 + t0) * 2) * 64 + lane;
    // epilogue thread (m = tid>>1, q = tid&1), tid < 64: its row's 36 sum-of-squares partials, issued up front
    const float4* sq = reinterpret_cast<const float4*>(ssq_in + ((int64_t)rb * 32 + ((tid >> 1) & 31)) * 40);
    float4 w[TPW], x0[TPW], x1[TPW], s4[9];
    uint32_t w8[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        if constexpr (W8 == 2) w8[i] = ldw8(Wp16, wslot + i * 64);
        else w[i] = ldw<W8 != 0>(Wp16, wslot + i * 64);
        x0[i] = xp[(i * 2) * 64];
        x1[i] = xp[(i * 2 + 1) * 64];
    }
    if (tid < 64) {
#pragma unroll
        for (int j = 0; j < 9; ++j) s4[j] = sq[j];
    }
    MELLOW_HOIST(a.eps); MELLOW_HOIST(a.guF);
    if (W8) MELLOW_HOIST(wscale);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (W8 == 2) {
        // rows 0..15 (x0) and 16..31 (x1): a row's values of this wave's k-tiles sit in the four lanes (m % 16) + 16 q
        float am0 = 0.f, am1 = 0.f, inv0, sc0, inv1, sc1;
#pragma unroll
        for (int i = 0; i < TPW; ++i) { am0 = fmaxf(am0, f4amax(x0[i])); am1 = fmaxf(am1, f4amax(x1[i])); }
        am0 = fmaxf(am0, swz_xor16(am0)); am1 = fmaxf(am1, swz_xor16(am1));
        a8_scales(half_max(am0), inv0, sc0);
        a8_scales(half_max(am1), inv1, sc1);
        uint32_t q0[TPW], q1[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) { q0[i] = q4_e4m3(x0[i], inv0); q1[i] = q4_e4m3(x1[i], inv1); }
        acc0 = mfma8_16<TPW>(acc0, w8, q0);
        acc1 = mfma8_16<TPW>(acc1, w8, q1);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] *= sc0; acc1[r] *= sc1; }
    } else {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, x0[i].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, x1[i].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, x0[i].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, x1[i].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, x0[i].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, x1[i].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, x0[i].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, x1[i].w, acc1, 0, 0, 0);
    }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(wave * 8 + r) * 64 + lane] = acc0[r];
        red[(wave * 8 + 4 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
    if (tid < 64) {
        // D[i = tile row 4*(lane>>4) + r][j = batch row lane&15], accumulator (m>>4).  Tile rows 0..7 = gate of hidden
        // units 8nt..8nt+7, rows 8..15 = up of the same units: thread (m, q) combines gate rows 4q..4q+3 (lane group q)
        // with up rows 8+4q.. (lane group q+2).
        const int m = tid >> 1, q = tid & 1;
        const int gl = (m & 15) + 16 * q, ul = gl + 32, rbase = (m >> 4) * 4;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) ss += (s4[j].x + s4[j].y) + (s4[j].z + s4[j].w);
        const float r2 = 1.0f / sqrtf(ss / 576.0f + a.eps);
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gv = red[(rbase + r) * 64 + gl], uv = red[(rbase + r) * 64 + ul];
#pragma unroll
            for (int wv = 1; wv < GU_WAVES; ++wv) {                       // fixed order
                gv += red[(wv * 8 + rbase + r) * 64 + gl];
                uv += red[(wv * 8 + rbase + r) * 64 + ul];
            }
            if (W8) { gv *= wscale[nt * 16 + 4 * q + r]; uv *= wscale[nt * 16 + 8 + 4 * q + r]; }     // packed tile rows
            h[r] = __fmul_rn(siluf_(gv * r2), uv * r2);
        }
        // hidden unit k = 8*nt + 4*q + r: down k-tile nt, F32-layout lane' = m + 32*q
        st_out(reinterpret_cast<float4*>(a.guF) + ((int64_t)rb * 192 + nt) * 64 + m + 32 * q, make_float4(h[0], h[1], h[2], h[3]));
    }
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// K5  down projection, split-K.  grid (18 n-tiles, DEC_KC_DOWN, RB), 6 waves x 4 k-tiles.
//     X = h (SwiGLU output of K4b, F32-layout);  out: down slabs in F32-layout (next layer's qkv / final norm)
// ----------------------------------------------------------------------------------------------------
#ifndef MELLOW_DOWN_WAVES
#define MELLOW_DOWN_WAVES 4
#endif
// waves per workgroup: 4 x 6 k-tiles (one wave per SIMD, a 4-way LDS reduction): 52.2 ms of decode per 63 steps against
// 53.0-53.2 with 6 x 4, 8 x 3 or 12 x 2 (same box, tools/ab_build.sh)
constexpr int DN_WAVES = MELLOW_DOWN_WAVES;
static_assert(DN_WAVES >= 4 && 192 % (8 * DN_WAVES) == 0, "the epilogue needs 256 threads; waves must divide the k-tiles");
template <bool BLK, int W8>
__global__ __launch_bounds__(DN_WAVES * 64) void dec_down_kernel(const float* __restrict__ Wp, const float* __restrict__ guF_p, int K8p,
                                                                 const DecArgs a, const float* __restrict__ wscale) {
    kspan(a.dbg_seq, 0);
    __shared__ __attribute__((aligned(16))) float red[DN_WAVES * 16 * 64];
    constexpr int KPW = 192 / (DEC_KC_DOWN * DN_WAVES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, kc = blockIdx.y, rb = blockIdx.z;
    MELLOW_BLK_EXIT(rb)
    const int k8_0 = (kc * DN_WAVES + wave) * KPW;
    const int64_t wslot = ((int64_t)nt * K8p + k8_0) * 64 + lane;
    const float4* hp = reinterpret_cast<const float4*>(guF_p) + ((int64_t)rb * 192 + k8_0) * 64 + lane;
    const bool dbg = tid == 0 && nt == 0 && kc == 0 && rb == 0;
    kstamp(4, 0, dbg);
    float4 w[KPW], h4[KPW];
    uint32_t w8[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        if constexpr (W8 == 2) w8[i] = ldw8(Wp, wslot + i * 64);
        else w[i] = ldw<W8 != 0>(Wp, wslot + i * 64);
        h4[i] = hp[i * 64];
    }
    MELLOW_HOIST(a.dslabF); MELLOW_HOIST(a.slabF_stride4);
    if (W8) MELLOW_HOIST(wscale);
    __builtin_amdgcn_sched_barrier(0);
    kstamp(4, 1, dbg);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if constexpr (W8 == 2) {
        float am = 0.f, inv, sc;
#pragma unroll
        for (int i = 0; i < KPW; ++i) am = fmaxf(am, f4amax(h4[i]));
        a8_scales(half_max(am), inv, sc);
        uint32_t xq[KPW];
#pragma unroll
        for (int i = 0; i < KPW; ++i) xq[i] = q4_e4m3(h4[i], inv);
        acc = mfma8_32<KPW>(acc, w8, xq);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] *= sc;
    } else {
#pragma unroll
        for (int i = 0; i < KPW; ++i) acc = mfma4(acc, w[i], h4[i]);
    }
    kstamp(4, 2, dbg && acc[0] == acc[0]);
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[q];
    __syncthreads();
    kstamp(4, 3, dbg);
    if (tid < 256) {
        const int mm = tid & 31, hh = (tid >> 5) & 1, gq = tid >> 6;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * gq + j;
            float sacc = 0.f;
#pragma unroll
            for (int wv = 0; wv < DN_WAVES; ++wv) sacc += red[(wv * 16 + q) * 64 + mm + 32 * hh];
            v[j] = sacc;
        }
        const int n = nt * 32 + 8 * gq + 4 * hh;
        if (W8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= wscale[n + j];
        }
        reinterpret_cast<float4*>(a.dslabF)[(int64_t)kc * a.slabF_stride4 + f32_idx(rb, 72, mm, n)] = make_float4(v[0], v[1], v[2], v[3]);
    }
    kstamp(4, 4, dbg);
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// K5+K1  the down projection of layer l and the q/k/v projection of layer l+1 in ONE launch (no kernel boundary between them).
//     qkv_{l+1} = W' x_new with x_new = x_mid + Wd h is linear in (x_mid, h):  W' x_mid + (W' Wd) h.  The product Q = W' Wd
//     (960 x 1536) is formed once at load time in fp64 and rounded to fp32 (engine_weights.cpp), and stored behind W' in one P-layout
//     matrix Wq2 [30 n-tiles][72 + 192 k-tiles].  The residual stream itself still goes through Wd (the "side" tiles), so the
//     composed weights only ever feed the next RMS-scaled projection, never the residual.
//     Workgroup types (grid.x):   [0, 60)              x part: n-tile b % 30, k-chunk b / 30 (2 chunks of 36 k-tiles of x_mid)
//                                 [60, 60 + 48 Q2_HC)  h part: k-chunk hc of h (192 / Q2_HC k-tiles);
//                                                      n-tile < 30: Q rows -> pq slab 2 + hc;  n-tile >= 30: Wd rows -> down slab hc
//     Consumer: dec_attn_kernel<FUSED> sums the Q2_NPQ pq slabs and forms x_new = x_mid + sum of the Q2_HC down slabs.
// ----------------------------------------------------------------------------------------------------
// Q2W = waves per workgroup (chosen per launch: 4 for one row block, Q2W otherwise)
// Operands: Wx = W' (P-layout, K8x k-tiles per n-tile), Wh = W' Wd (K8h k-tiles per n-tile), Wd = the down weight (192 k-tiles).
//   fp32 (W8 == 0): Wx and Wh are the two column ranges of ONE matrix [30][72 + 192] (K8x = K8h = Q2_K8)
//   e4m3 (W8 != 0): three separately quantised matrices (the q/k/v copy of the unfused layer, the composed one, the down copy),
//                   one scale per packed row each: sc_x, sc_h, sc_d; W8 == 2 also quantises x_mid / h per wave slice (see ldw8)
template <bool BLK, int Q2W, int W8>
__global__ __launch_bounds__(Q2W * 64) void dec_qkv2_kernel(const float* __restrict__ Wx, const float* __restrict__ Wh,
                                                            const float* __restrict__ Wd, const float* __restrict__ xmidF_p,
                                                            const float* __restrict__ guF_p, int K8x, int K8h, const DecArgs a,
                                                            const float* __restrict__ sc_x, const float* __restrict__ sc_h,
                                                            const float* __restrict__ sc_d) {
    kspan(a.dbg_seq, 0);
    __shared__ __attribute__((aligned(16))) float red[Q2W * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, rb = blockIdx.y;
    MELLOW_BLK_EXIT(rb)
    const bool dbg = tid == 0 && b == 60 && rb == 0;          // an h-part workgroup (the longest kind)
    kstamp(7, 0, dbg);
    constexpr int XT = 36 / Q2W, HT = (192 / Q2_HC) / Q2W;      // k-tiles per wave
    static_assert(XT * Q2W == 36 && HT * Q2W * Q2_HC == 192, "waves must divide the k-chunks");
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int nt, slab;
    bool side = false;
    const float* wsc = sc_x;
    if (b < 60) {              // x part (workgroup-uniform branch; each side keeps its loads straight-line)
        nt = b % 30; slab = b / 30;
        const int k8_0 = slab * 36 + wave * XT;
        const int64_t wslot = ((int64_t)nt * K8x + k8_0) * 64 + lane;
        const float4* xb = reinterpret_cast<const float4*>(xmidF_p) + ((int64_t)rb * 72 + k8_0) * 64 + lane;
        float4 w[XT], x[XT];
        uint32_t w8[XT];
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            if constexpr (W8 == 2) w8[i] = ldw8(Wx, wslot + i * 64);
            else w[i] = ldw<W8 != 0>(Wx, wslot + i * 64);
            x[i] = xb[i * 64];
        }
        MELLOW_HOIST(a.pq); MELLOW_HOIST(a.rows);
        if (W8) MELLOW_HOIST(sc_x);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (W8 == 2) {
            float am = 0.f, inv, sc;
#pragma unroll
            for (int i = 0; i < XT; ++i) am = fmaxf(am, f4amax(x[i]));
            a8_scales(half_max(am), inv, sc);
            uint32_t xq[XT];
#pragma unroll
            for (int i = 0; i < XT; ++i) xq[i] = q4_e4m3(x[i], inv);
            acc = mfma8_32<XT>(acc, w8, xq);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= sc;
        } else {
#pragma unroll
            for (int i = 0; i < XT; ++i) acc = mfma4(acc, w[i], x[i]);
        }
    } else {                   // h part
        const int idx = b - 60, hc = idx / 48;
        nt = idx % 48;
        side = nt >= 30;
        slab = side ? hc : 2 + hc;
        wsc = side ? sc_d : sc_h;
        const int k8_0 = hc * (192 / Q2_HC) + wave * HT;
        const float* wbase = side ? Wd : Wh;
        const int64_t wslot = side ? ((int64_t)(nt - 30) * 192 + k8_0) * 64 + lane : ((int64_t)nt * K8h + k8_0) * 64 + lane;
        const float4* hp = reinterpret_cast<const float4*>(guF_p) + ((int64_t)rb * 192 + k8_0) * 64 + lane;
        float4 w[HT], h4[HT];
        uint32_t w8[HT];
#pragma unroll
        for (int i = 0; i < HT; ++i) {
            if constexpr (W8 == 2) w8[i] = ldw8(wbase, wslot + i * 64);
            else w[i] = ldw<W8 != 0>(wbase, wslot + i * 64);
            h4[i] = hp[i * 64];
        }
        MELLOW_HOIST(a.pq); MELLOW_HOIST(a.rows); MELLOW_HOIST(a.dslabF); MELLOW_HOIST(a.slabF_stride4);
        if (W8) MELLOW_HOIST(wsc);
        __builtin_amdgcn_sched_barrier(0);
        kstamp(7, 1, dbg);
        if constexpr (W8 == 2) {
            float am = 0.f, inv, sc;
#pragma unroll
            for (int i = 0; i < HT; ++i) am = fmaxf(am, f4amax(h4[i]));
            a8_scales(half_max(am), inv, sc);
            uint32_t xq[HT];
#pragma unroll
            for (int i = 0; i < HT; ++i) xq[i] = q4_e4m3(h4[i], inv);
            acc = mfma8_32<HT>(acc, w8, xq);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= sc;
        } else {
#pragma unroll
            for (int i = 0; i < HT; ++i) acc = mfma4(acc, w[i], h4[i]);
        }
        kstamp(7, 2, dbg && acc[0] == acc[0]);
        if (side) nt -= 30;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    kstamp(7, 3, dbg);
    if (tid < 256) {
        const int mm = tid & 31, hh = (tid >> 5) & 1, gq = tid >> 6;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * gq + j;
            float sacc = red[q * 64 + mm + 32 * hh];
#pragma unroll
            for (int wv = 1; wv < Q2W; ++wv) sacc += red[(wv * 16 + q) * 64 + mm + 32 * hh];      // fixed order
            v[j] = sacc;
        }
        const int n = nt * 32 + 8 * gq + 4 * hh;
        if (W8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= wsc[n + j];
        }
        const float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (side) st_out(reinterpret_cast<float4*>(a.dslabF) + (int64_t)slab * a.slabF_stride4 + f32_idx(rb, 72, mm, n), o);
        else st_out(reinterpret_cast<float4*>(a.pq + ((int64_t)slab * a.rows + rb * 32 + mm) * 960 + n), o);
    }
    kstamp(7, 4, dbg);
    kspan(a.dbg_seq, 1);
}

// ----------------------------------------------------------------------------------------------------
// row-parallel helpers (one workgroup per batch row)
// ----------------------------------------------------------------------------------------------------
// final RMSNorm: xn = w * ((x_mid + sum down slabs) * rsqrt(mean^2 + eps)) -> F32-layout operand of the lm_head
template <int KCD, bool BLK>
__global__ __launch_bounds__(192) void dec_final_norm_kernel(const float* __restrict__ norm_w, const float* __restrict__ xmidF_p,
                                                             const float* __restrict__ dslabF_p, int64_t stride4_p, const DecArgs a) {
    kspan(a.dbg_seq, 0);
    __shared__ float part[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (BLK && b == 0 && tid < 32) a.blk_snap[tid] = a.blk_live[tid];      // for this step's arg-max (kernels.h)
    MELLOW_BLK_EXIT(b >> 5)
    const int xi = tid < 144 ? tid : 0;
    const int64_t fi = f32_idx(b >> 5, 72, b & 31, xi * 4);
    float4 v = reinterpret_cast<const float4*>(xmidF_p)[fi];
    float4 xs[KCD > 0 ? KCD : 1];
#pragma unroll
    for (int s = 0; s < KCD; ++s) xs[s] = reinterpret_cast<const float4*>(dslabF_p)[(int64_t)s * stride4_p + fi];
    const float4 wv = reinterpret_cast<const float4*>(norm_w)[xi];
#pragma unroll
    for (int s = 0; s < KCD; ++s) v = f4add(v, xs[s]);
    float ss = tid < 144 ? f4ssq(v) : 0.f;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float r = 1.0f / sqrtf((part[0] + part[1] + part[2]) / 576.0f + a.eps);
    if (tid < 144) {
        float4 y;
        y.x = __fmul_rn(wv.x, __fmul_rn(v.x, r)); y.y = __fmul_rn(wv.y, __fmul_rn(v.y, r));
        y.z = __fmul_rn(wv.z, __fmul_rn(v.z, r)); y.w = __fmul_rn(wv.w, __fmul_rn(v.w, r));
        if (!a.xn3) reinterpret_cast<float4*>(a.xnF)[fi] = y;
    }
    if (a.xn3) {                       // f32x3 lm_head: pre-split, F3-32; threads 2i, 2i + 1 hold the halves of one slot
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 144) {
            y.x = __fmul_rn(wv.x, __fmul_rn(v.x, r)); y.y = __fmul_rn(wv.y, __fmul_rn(v.y, r));
            y.z = __fmul_rn(wv.z, __fmul_rn(v.z, r)); y.w = __fmul_rn(wv.w, __fmul_rn(v.w, r));
        }
        const F3Quad mine = f3_split4(y), other = f3_partner(mine);
        if (tid < 144 && !(tid & 1)) f3_store8(a.xn3, f3_32_slot(b >> 5, 36, b & 31, tid * 4), mine, other);
    }
    kspan(a.dbg_seq, 1);
}

// arg-max over the lm_head's per-tile candidates, fused with the loop bookkeeping of reference wrapper.py:232-249:
// record the token at column (*d_pos - T0 + 1), track stop ids, and gather its embedding row (embed_tokens,
// wrapper.py:237) as the next step's residual stream (row-major + F32-layout).
__global__ __launch_bounds__(256) void dec_argmax_kernel(const float* __restrict__ cand_val_p, const int32_t* __restrict__ cand_idx_p, int n,
                                                         const DecArgs a, int32_t* __restrict__ tokens,
                                                         const float* __restrict__ embed, int write_x, const LoopArgs lp) {
    kspan(a.dbg_seq, 0);
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int tok_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    // a slot whose whole block has stopped (or that is empty after a repack) computes nothing any more, but still takes part
    // in the step's arrival count
    const int row = lp.row_of_slot ? lp.row_of_slot[b] : b;         // the example in this slot
    const bool dead = (lp.blk_snap && lp.blk_snap[b >> 5] == 0) || row < 0;      // workgroup-uniform; written by an EARLIER launch
    float best = -INFINITY;
    int idx = 0x7fffffff;
    if (!dead) {
        for (int i = tid; i < n; i += 256) {
            const float v = cand_val_p[(int64_t)b * n + i];
            const int id = cand_idx_p[(int64_t)b * n + i];
            if (arg_better(v, id, best, idx)) { best = v; idx = id; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(idx, off, 64);
            if (arg_better(ov, oi, best, idx)) { best = ov; idx = oi; }
        }
        if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    }
    __syncthreads();
    if (tid == 0) {
        if (!dead) {
            for (int w = 1; w < 4; ++w)
                if (arg_better(bv[w], bi[w], best, idx)) { best = bv[w]; idx = bi[w]; }
            idx = min(max(idx, 0), n * 32 - 1);       // always a valid row of the embedding table (n tiles x 32 ids = vocab)
            tokens[b] = idx;
            tok_s = idx;
        }
        if (lp.out_tokens) {
            if (!dead) {
                const int max_len = lp.params[0], stop_id = lp.params[1];
                const int step = *a.d_pos - lp.T0 + 1;
                if (step >= 0 && step < max_len) lp.out_tokens[(int64_t)row * max_len + step] = idx;
                if (idx == stop_id && lp.seen_stop[row] == 0) {
                    lp.seen_stop[row] = 1;
                    atomicAdd(lp.n_seen, 1);
                    if (lp.blk_left && atomicSub(lp.blk_left + (b >> 5), 1) == 1) lp.blk_live[b >> 5] = 0;   // from the NEXT step on
                }
            }
            // the last row of this launch publishes the step: ticket = arg-max launches so far, low word = rows stopped.
            // Ordering: this row's n_seen / blk_left atomics must have been performed before its arrival is counted -- a drained
            // vmcnt is enough for device-scope atomics (no cache write-back: a __threadfence() here cost ~3 us per step).  The
            // host reads only the progress word itself before the stream is synchronised, so its store needs no release.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(lp.arrive, 1) == (int)gridDim.x - 1) {
                *lp.arrive = 0;
                const int t = *lp.ticket + 1;
                *lp.ticket = t;
                const int ns = atomicAdd(lp.n_seen, 0);
                __hip_atomic_store(lp.host_progress, ((unsigned long long)(unsigned)t << 32) | (unsigned)ns, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (dead) return;
    if (write_x) {
        __syncthreads();
        if (tid < 144) {
            const float4 e = reinterpret_cast<const float4*>(embed + (int64_t)tok_s * 576)[tid];
            reinterpret_cast<float4*>(a.xmidF)[f32_idx(b >> 5, 72, b & 31, tid * 4)] = e;
        }
    }
    kspan(a.dbg_seq, 1);
}

// Row migration (reference stop rule, more than one 32-row block).  One workgroup, after the step's arg-max: the rows that
// have not produced the stop id yet are counted; if they fit into fewer 32-row blocks than are live, they are packed (stable)
// into the lowest slots: their next-step residual rows move, row_of_slot is rewritten, the emptied blocks' live words are
// cleared.  A row's arithmetic does not depend on its slot (MFMA rows are independent, every reduction is per row), so the
// tokens are unchanged; rows that had already stopped are dropped (their texts are cut at the stop id anyway).
__global__ __launch_bounds__(1024) void dec_compact_kernel(const DecArgs a, int B, const LoopArgs lp) {
    __shared__ int src_of[1024];           // new slot -> old slot
    __shared__ int row_new[1024];
    __shared__ int wsum[16];
    __shared__ int n_act_s, n_live_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = (B + 31) >> 5;
    int row = -1, active = 0;
    if (tid < B) {
        row = lp.row_of_slot[tid];
        active = row >= 0 && lp.blk_live[tid >> 5] != 0 && lp.seen_stop[row] == 0;
    }
    // exclusive prefix sum of `active` over the slots: ballot inside a wave, then over the 16 waves
    const unsigned long long m = __ballot(active);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    if (tid == 0) {
        int live = 0;
        for (int k = 0; k < nblk; ++k) live += lp.blk_live[k] != 0;
        n_live_s = live;
    }
    src_of[tid] = -1;
    row_new[tid] = -1;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < 16; ++w) n += wsum[w];
        n_act_s = n;
    }
    __syncthreads();
    const int n_act = n_act_s;
    if ((n_act + 31) / 32 >= n_live_s) return;            // repacking would not empty a block (workgroup-uniform)
    if (active) { src_of[base + before] = tid; row_new[base + before] = row; }
    __syncthreads();
    // move the residual rows, 7 destination slots (7 x 144 float4 = 1008 threads) at a time in increasing order: a destination
    // is never above its source and sources only grow, so a chunk can only overwrite sources of chunks already done
    float4* x = reinterpret_cast<float4*>(a.xmidF);
    for (int d0 = 0; d0 < n_act; d0 += 7) {
        const int d = d0 + tid / 144, c = tid % 144;
        const bool mv = tid < 1008 && d < n_act && src_of[d] != d;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mv) v = x[f32_idx(src_of[d] >> 5, 72, src_of[d] & 31, c * 4)];
        __syncthreads();
        if (mv) x[f32_idx(d >> 5, 72, d & 31, c * 4)] = v;
        __syncthreads();
    }
    if (tid < B) lp.row_of_slot[tid] = row_new[tid];
    if (tid < 32) {
        const int left = tid < nblk ? min(max(n_act - 32 * tid, 0), 32) : 0;
        lp.blk_left[tid] = left;
        lp.blk_live[tid] = left > 0;
    }
    if (tid == 0) *lp.n_compactions += 1;
}
// fp32 K/V pages -> their bf16 shadow (KV16; round to nearest even), 8 values per thread
__global__ __launch_bounds__(256) void kv_to_bf16_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
        const bf16x8v o = {static_cast<__bf16>(a.x), static_cast<__bf16>(a.y), static_cast<__bf16>(a.z), static_cast<__bf16>(a.w),
                           static_cast<__bf16>(b.x), static_cast<__bf16>(b.y), static_cast<__bf16>(b.z), static_cast<__bf16>(b.w)};
        dst[i] = __builtin_bit_cast(uint4, o);
    }
}
void launch_kv_to_bf16(const float* src, void* dst, int64_t n, hipStream_t s) {
    const int64_t n8 = n >> 3;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(kv_to_bf16_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(src), reinterpret_cast<uint4*>(dst), n8);
}
void launch_dec_compact(const DecArgs& a, int B, const LoopArgs& loop, hipStream_t s) {
    hipLaunchKernelGGL(dec_compact_kernel, dim3(1), dim3(1024), 0, s, a, B, loop);
}

// rows -> residual stream (row-major + F32-layout): src row b = in[row_of(b)]
__global__ __launch_bounds__(192) void dec_load_rows_kernel(const DecArgs a, const float* __restrict__ in, int64_t ld,
                                                            const int32_t* __restrict__ row_ids, int T_last, int n_src) {
    const int b = blockIdx.x, tid = threadIdx.x;
    // caller-supplied ids are clamped to the table (an out-of-range id must not become a wild read; the Python
    // binding rejects it with an IndexError before it gets here)
    const int64_t src = row_ids ? (int64_t)min(max(row_ids[b], 0), n_src - 1) : (int64_t)b * T_last + (T_last - 1);
    if (tid < 144) {
        const float4 e = reinterpret_cast<const float4*>(in + src * ld)[tid];
        reinterpret_cast<float4*>(a.xmidF)[f32_idx(b >> 5, 72, b & 31, tid * 4)] = e;
    }
}

// ---- launchers -------------------------------------------------------------------------------------------
// Dynamic LDS beyond 64 KiB has to be allowed per kernel function AND per device (an engine per GPU, pools of host threads):
// remembered per (function, device) under a lock; the attribute call itself is cheap but not free on a per-step path.
void set_max_dynamic_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({fn, dev}).second) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) fprintf(stderr, "mellow: hipFuncSetAttribute(max dynamic LDS %zu) failed: %s\n", bytes, hipGetErrorString(e));
    }
}
// BLK = per-row-block early exit compiled in (a.blk_live != null); W8 = e4m3 weights (wscale != null): chosen on the host
#define MELLOW_LAUNCH_BLK(KERNEL, GRID, BLOCK, ...)                                                               \
    do {                                                                                                         \
        if (a.blk_live) hipLaunchKernelGGL((KERNEL<true>), GRID, BLOCK, 0, s, __VA_ARGS__);                       \
        else hipLaunchKernelGGL((KERNEL<false>), GRID, BLOCK, 0, s, __VA_ARGS__);                                 \
    } while (0)
// weight mode of a launch: 0 fp32, 1 e4m3 weights widened to fp32, 2 e4m3 weights and activations on the fp8 matrix pipe (a.a8)
static inline int w8_mode(const DecArgs& a, const float* wscale) { return wscale ? (a.a8 ? 2 : 1) : 0; }
#define MELLOW_LAUNCH_BLK_W8(KERNEL, GRID, BLOCK, ...)                                                            \
    do {                                                                                                         \
        const int mode_ = w8_mode(a, wscale);                                                                    \
        if (a.blk_live && mode_ == 2) hipLaunchKernelGGL((KERNEL<true, 2>), GRID, BLOCK, 0, s, __VA_ARGS__, wscale);    \
        else if (a.blk_live && mode_ == 1) hipLaunchKernelGGL((KERNEL<true, 1>), GRID, BLOCK, 0, s, __VA_ARGS__, wscale); \
        else if (a.blk_live) hipLaunchKernelGGL((KERNEL<true, 0>), GRID, BLOCK, 0, s, __VA_ARGS__, wscale);              \
        else if (mode_ == 2) hipLaunchKernelGGL((KERNEL<false, 2>), GRID, BLOCK, 0, s, __VA_ARGS__, wscale);             \
        else if (mode_ == 1) hipLaunchKernelGGL((KERNEL<false, 1>), GRID, BLOCK, 0, s, __VA_ARGS__, wscale);             \
        else hipLaunchKernelGGL((KERNEL<false, 0>), GRID, BLOCK, 0, s, __VA_ARGS__, wscale);                             \
    } while (0)
void launch_dec_qkv(const DecArgs& a, const float* Wp, int K8p, int kcd, hipStream_t s, const float* wscale) {
    const dim3 grid(30, DEC_KC_QKV, a.RB);
    // the first qkv launch of a step (a.first) always starts from a materialised x (kcd == 0); later ones sum the down slabs
#define MELLOW_QKV(KCD, FIRST)                                                                              \
    do {                                                                                                    \
        const int mode_ = w8_mode(a, wscale);                                                                \
        if (a.blk_live && mode_ == 2) hipLaunchKernelGGL((dec_qkv_kernel<KCD, true, FIRST, 2>), grid, dim3(QKV_THREADS), 0, s, Wp, (const float*)a.xmidF, (const float*)a.dslabF, K8p, a, wscale);   \
        else if (a.blk_live && mode_ == 1) hipLaunchKernelGGL((dec_qkv_kernel<KCD, true, FIRST, 1>), grid, dim3(QKV_THREADS), 0, s, Wp, (const float*)a.xmidF, (const float*)a.dslabF, K8p, a, wscale);   \
        else if (a.blk_live) hipLaunchKernelGGL((dec_qkv_kernel<KCD, true, FIRST, 0>), grid, dim3(QKV_THREADS), 0, s, Wp, (const float*)a.xmidF, (const float*)a.dslabF, K8p, a, wscale);        \
        else if (mode_ == 2) hipLaunchKernelGGL((dec_qkv_kernel<KCD, false, FIRST, 2>), grid, dim3(QKV_THREADS), 0, s, Wp, (const float*)a.xmidF, (const float*)a.dslabF, K8p, a, wscale);            \
        else if (mode_ == 1) hipLaunchKernelGGL((dec_qkv_kernel<KCD, false, FIRST, 1>), grid, dim3(QKV_THREADS), 0, s, Wp, (const float*)a.xmidF, (const float*)a.dslabF, K8p, a, wscale);            \
        else hipLaunchKernelGGL((dec_qkv_kernel<KCD, false, FIRST, 0>), grid, dim3(QKV_THREADS), 0, s, Wp, (const float*)a.xmidF, (const float*)a.dslabF, K8p, a, wscale);                       \
    } while (0)
    if (kcd == 0 && a.first) MELLOW_QKV(0, true);
    else if (kcd == 0) MELLOW_QKV(0, false);
    else if (a.first) MELLOW_QKV(DEC_KC_DOWN, true);
    else MELLOW_QKV(DEC_KC_DOWN, false);
#undef MELLOW_QKV
}
void launch_dec_attn(const DecArgs& a, float* k_cache, float* v_cache, bool fused, hipStream_t s) {
    const dim3 grid(3, a.rows, a.ts), block(DA_WAVES * 64);
    const bool one = a.RB == 1 && !a.blk_live;
    if (a.ts != DEC_TS && (one || a.kv16 || a.ts != DEC_TS_MULTI)) {      // kernels.h dec_key_splits
        fprintf(stderr, "mellow: decode attention launched with %d key splits at RB = %d\n", a.ts, a.RB);
        abort();
    }
    // bf16 pages, matrix form: MELLOW_DA16_NW physical waves from two row blocks on -- kernel comment at DA_GM
#define MELLOW_DA(BLKV, FUSEDV, ONEV)                                                                                    \
    do {                                                                                                                 \
        constexpr int NW16 = MELLOW_DA16_MFMA && DA_WAVES == 8 && !(ONEV) ? MELLOW_DA16_NW : DA_WAVES;                     \
        constexpr int TSM = (ONEV) ? DEC_TS : DEC_TS_MULTI;        /* (ONE never meets another split count) */            \
        if (a.kv16) hipLaunchKernelGGL((dec_attn_kernel<BLKV, FUSEDV, ONEV, true, NW16, DEC_TS>), grid, dim3(NW16 * 64), 0, s, k_cache, v_cache, (const int32_t*)a.d_pos, \
                                       (const float*)a.pq, (const float*)a.xmidF, a.Tmax, a.gs, a.rows, a);               \
        else if (a.ts == DEC_TS) hipLaunchKernelGGL((dec_attn_kernel<BLKV, FUSEDV, ONEV, false, DA_WAVES, DEC_TS>), grid, block, 0, s, k_cache, v_cache, (const int32_t*)a.d_pos, \
                                                    (const float*)a.pq, (const float*)a.xmidF, a.Tmax, a.gs, a.rows, a);  \
        else hipLaunchKernelGGL((dec_attn_kernel<BLKV, FUSEDV, ONEV, false, DA_WAVES, TSM>), grid, block, 0, s, k_cache, v_cache, (const int32_t*)a.d_pos, \
                                (const float*)a.pq, (const float*)a.xmidF, a.Tmax, a.gs, a.rows, a);                      \
    } while (0)
    // (the per-block early exit exists only with more than one row block, so <BLK, ONE> never meet)
    if (one) {
        if (fused) MELLOW_DA(false, true, true); else MELLOW_DA(false, false, true);
    } else if (a.blk_live && fused) MELLOW_DA(true, true, false);
    else if (a.blk_live) MELLOW_DA(true, false, false);
    else if (fused) MELLOW_DA(false, true, false);
    else MELLOW_DA(false, false, false);
#undef MELLOW_DA
}
// fp32 form: Wq2 = [30][Q2_K8] (W' | W' Wd);  e4m3 form (sc_x != null): three separately quantised matrices (kernel comment)
static void launch_dec_qkv2_any(const DecArgs& a, const float* Wx, int K8x, const float* Wh, int K8h, const float* Wd,
                                const float* sc_x, const float* sc_h, const float* sc_d, hipStream_t s) {
    // waves per workgroup: 4 x (9 | 12) k-tiles when there is one row block (a 4-way instead of a 12-way LDS reduction: 49.55 vs
    // 49.98 ms of decode per 63 steps at B = 32), 12 x (3 | 4) otherwise (B = 64: 76.3 vs 77.2 ms); -DMELLOW_Q2_WAVES=n forces one
    const dim3 grid(Q2_BLOCKS, a.RB);
#ifdef MELLOW_Q2_WAVES_FORCED
    bool few = false;
#else
    bool few = a.RB == 1;
#endif
    const int mode = w8_mode(a, sc_x);
    // (activations on the fp8 pipe: a wave's k-slice is the unit of the activation scale, so the wave count must not depend on
    //  the batch size -- a row's tokens would otherwise change with the number of row blocks around it)
    if (mode == 2) few = false;
#define MELLOW_Q2(BLKV, WV, MODE) \
    hipLaunchKernelGGL((dec_qkv2_kernel<BLKV, WV, MODE>), grid, dim3(WV * 64), 0, s, Wx, Wh, Wd, (const float*)a.xmidF, (const float*)a.guF, \
                       K8x, K8h, a, sc_x, sc_h, sc_d)
#define MELLOW_Q2_MODES(BLKV, WV)                  \
    do {                                           \
        if (mode == 2) MELLOW_Q2(BLKV, WV, 2);     \
        else if (mode == 1) MELLOW_Q2(BLKV, WV, 1);\
        else MELLOW_Q2(BLKV, WV, 0);               \
    } while (0)
    if (few) {
        if (a.blk_live) MELLOW_Q2_MODES(true, 4); else MELLOW_Q2_MODES(false, 4);
    } else {
        if (a.blk_live) MELLOW_Q2_MODES(true, Q2_WAVES); else MELLOW_Q2_MODES(false, Q2_WAVES);
    }
#undef MELLOW_Q2_MODES
#undef MELLOW_Q2
}
// The launchers raise a kernel's dynamic-LDS limit on first use; a decode step is launched inside a stream capture, so the engine
// calls this once per device beforehand (ensure_lm): every instantiation that needs more than 64 KiB.
void dec_prepare_lds_attributes() {
    const size_t head = (size_t)36 * 3 * 64 * 16, q4 = (size_t)Q3W * 4 * 16 * 64 * 4;
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<1, MELLOW_H3R_D1, true>), head);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<1, MELLOW_H3R_D1, false>), head);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<2, MELLOW_H3R_D2, true>), head);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<2, MELLOW_H3R_D2, false>), head);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<4, MELLOW_H3R_D4, true>), head);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<4, MELLOW_H3R_D4, false>), head);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_qkv2x3_kernel<true, 4>), q4);
    set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_qkv2x3_kernel<false, 4>), q4);
}
void launch_dec_qkv2x3(const DecArgs& a, const float* Wq2, const float* Wd, hipStream_t s) {
    const float* Wh = Wq2 + (size_t)72 * 64 * 4;
    const i32x4 *x3 = reinterpret_cast<const i32x4*>(a.xmid3_32), *h3 = reinterpret_cast<const i32x4*>(a.h3);
#define MELLOW_Q3(BLKV, RBMV)                                                                                        \
    do {                                                                                                             \
        const size_t lds = (size_t)Q3W * RBMV * 16 * 64 * 4;                                                         \
        set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_qkv2x3_kernel<BLKV, RBMV>), lds);                             \
        hipLaunchKernelGGL((dec_qkv2x3_kernel<BLKV, RBMV>), dim3(Q2_BLOCKS), dim3(Q3W * 64), lds, s, Wq2, Wh, Wd, x3, h3, Q2_K8, Q2_K8, a.RB, a); \
    } while (0)
    if (a.RB <= 1) { if (a.blk_live) MELLOW_Q3(true, 1); else MELLOW_Q3(false, 1); }
    else if (a.RB == 2) { if (a.blk_live) MELLOW_Q3(true, 2); else MELLOW_Q3(false, 2); }
    else { if (a.blk_live) MELLOW_Q3(true, 4); else MELLOW_Q3(false, 4); }
#undef MELLOW_Q3
}
void launch_dec_gateup3(const DecArgs& a, const float* Wp16, hipStream_t s) {
    const i32x4* x3 = reinterpret_cast<const i32x4*>(a.xmid3_16);
#define MELLOW_G3(BLKV, RBMV)                                                                                        \
    do {                                                                                                             \
        const size_t lds = (size_t)GU3W * RBMV * 8 * 64 * 4;                                                         \
        set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_gateup3_kernel<BLKV, RBMV>), lds);                            \
        hipLaunchKernelGGL((dec_gateup3_kernel<BLKV, RBMV>), dim3(192), dim3(GU3W * 64), lds, s, Wp16, x3, (const float*)a.ssq, a.RB, a); \
    } while (0)
    if (a.RB <= 1) { if (a.blk_live) MELLOW_G3(true, 1); else MELLOW_G3(false, 1); }
    else if (a.RB == 2) { if (a.blk_live) MELLOW_G3(true, 2); else MELLOW_G3(false, 2); }
    else { if (a.blk_live) MELLOW_G3(true, 4); else MELLOW_G3(false, 4); }
#undef MELLOW_G3
}
void launch_dec_qkv2(const DecArgs& a, const float* Wq2, const float* Wd, hipStream_t s) {
    launch_dec_qkv2_any(a, Wq2, Q2_K8, Wq2 + (size_t)72 * 64 * 4, Q2_K8, Wd, nullptr, nullptr, nullptr, s);
}
void launch_dec_qkv2_w8(const DecArgs& a, const float* Wx8, const float* sc_x, const float* Wh8, const float* sc_h,
                        const float* Wd8, const float* sc_d, hipStream_t s) {
    launch_dec_qkv2_any(a, Wx8, 72, Wh8, 192, Wd8, sc_x, sc_h, sc_d, s);
}
// C[M][N] (fp32) = A[M][K] . B[K][N] with fp64 products and accumulation, rounded once: the load-time composition of two
// weight matrices (W_qkv' . W_down) for dec_qkv2_kernel.  16 x 16 outputs per workgroup, operands staged through LDS.
__global__ __launch_bounds__(256) void compose_f64_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          int M, int N, int K) {
    __shared__ double as[16][17], bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        as[ty][tx] = (m < M && k0 + tx < K) ? (double)A[(int64_t)m * K + k0 + tx] : 0.0;
        bs[ty][tx] = (k0 + ty < K && n < N) ? (double)B[(int64_t)(k0 + ty) * N + n] : 0.0;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += as[ty][kk] * bs[kk][tx];
        __syncthreads();
    }
    if (m < M && n < N) C[(int64_t)m * N + n] = (float)acc;
}
void launch_compose_f64(const float* A, const float* B, float* C, int M, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(compose_f64_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, s, A, B, C, M, N, K);
}
int dec_attn_chunk_groups(bool kv16) { return DA_WAVES * (kv16 && MELLOW_DA16_MFMA ? DA_GM : DA_G); }
void launch_dec_oproj(const DecArgs& a, const float* Wp16, hipStream_t s, const float* wscale) {
#ifdef MELLOW_OPROJ_ROWS
    const int rows = MELLOW_OPROJ_ROWS;
#else
    const int rows = a.RB == 1 ? 8 : 16;
#endif
#define MELLOW_OPROJ_L(BLKV, W8V, ROWSV, TSV, PARTSV)                                                                      \
    hipLaunchKernelGGL((dec_oproj_kernel<BLKV, W8V, ROWSV, TSV>), dim3(36, PARTSV * a.RB), dim3(OP_WAVES * 64), 0, s, Wp16,    \
                       (const float*)a.attF16, (const float*)a.att_ml, (const float*)a.xnewR, a.RB, a.rows, a, wscale)
#define MELLOW_OPROJ(BLKV, W8V)                                                                                             \
    do {                                                                                                                    \
        if (rows == 8 && a.ts == 1) MELLOW_OPROJ_L(BLKV, W8V, 8, 1, 4);                                                     \
        else if (rows == 8) MELLOW_OPROJ_L(BLKV, W8V, 8, DEC_TS, 4);                                                        \
        else if (a.ts == 1) MELLOW_OPROJ_L(BLKV, W8V, 16, 1, 2);                                                            \
        else MELLOW_OPROJ_L(BLKV, W8V, 16, DEC_TS, 2);                                                                      \
    } while (0)
    const int mode = w8_mode(a, wscale);
    if (a.blk_live && mode == 2) MELLOW_OPROJ(true, 2);
    else if (a.blk_live && mode == 1) MELLOW_OPROJ(true, 1);
    else if (a.blk_live) MELLOW_OPROJ(true, 0);
    else if (mode == 2) MELLOW_OPROJ(false, 2);
    else if (mode == 1) MELLOW_OPROJ(false, 1);
    else MELLOW_OPROJ(false, 0);
#undef MELLOW_OPROJ
#undef MELLOW_OPROJ_L
}
void launch_dec_gateup(const DecArgs& a, const float* Wp16, hipStream_t s, const float* wscale) {
    MELLOW_LAUNCH_BLK_W8(dec_gateup16_kernel, dim3(192, a.RB), dim3(GU_WAVES * 64), Wp16, a.xmidF16, a.ssq, a);
}
void launch_dec_down(const DecArgs& a, const float* Wp, int K8p, hipStream_t s, const float* wscale) {
    MELLOW_LAUNCH_BLK_W8(dec_down_kernel, dim3(18, DEC_KC_DOWN, a.RB), dim3(DN_WAVES * 64), Wp, (const float*)a.guF, K8p, a);
}
void launch_dec_final_norm(const DecArgs& a, const float* norm_w, int kcd, hipStream_t s) {
    if (kcd == 0) {
        if (a.blk_live) hipLaunchKernelGGL((dec_final_norm_kernel<0, true>), dim3(a.rows), dim3(192), 0, s, norm_w, (const float*)a.xmidF, (const float*)a.dslabF, a.slabF_stride4, a);
        else hipLaunchKernelGGL((dec_final_norm_kernel<0, false>), dim3(a.rows), dim3(192), 0, s, norm_w, (const float*)a.xmidF, (const float*)a.dslabF, a.slabF_stride4, a);
    } else {
        if (a.blk_live) hipLaunchKernelGGL((dec_final_norm_kernel<DEC_KC_DOWN, true>), dim3(a.rows), dim3(192), 0, s, norm_w, (const float*)a.xmidF, (const float*)a.dslabF, a.slabF_stride4, a);
        else hipLaunchKernelGGL((dec_final_norm_kernel<DEC_KC_DOWN, false>), dim3(a.rows), dim3(192), 0, s, norm_w, (const float*)a.xmidF, (const float*)a.dslabF, a.slabF_stride4, a);
    }
}
// true when the streaming f32x3 lm_head (dec_head3r_kernel) tiles this vocabulary: only then may the final norm hand its output
// over pre-split (DecArgs::xn3, which aliases xnF) -- engine_lm.cpp: ensure_lm
bool dec_head3r_fits(int vocab) { return vocab % 32 == 0 && (vocab / 32) % H3_NW == 0; }
void launch_dec_lm_head(const DecArgs& a, const float* Wp, int K8p, int vocab, hipStream_t s, const float* wscale) {
    if ((a.x3 & DEC_X3_HEAD) && !wscale && a.xn3 && (vocab / 32) % H3_NW == 0) {
        // f32x3 mode, activations pre-split by the final norm: the streaming form (weights read once for every row block)
        const dim3 grid(vocab / 32 / H3_NW), block(H3_NW * 64);
        const i32x4* X3 = reinterpret_cast<const i32x4*>(a.xn3);
#define MELLOW_H3R(GV, DV)                                                                                                \
        do {                                                                                                              \
            const size_t lds = (size_t)36 * 3 * 64 * 16;                                                                  \
            if (a.blk_live) { set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<GV, DV, true>), lds);  \
                              hipLaunchKernelGGL((dec_head3r_kernel<GV, DV, true>), grid, block, lds, s, Wp, X3, K8p, vocab, a.RB, a); } \
            else { set_max_dynamic_lds(reinterpret_cast<const void*>(&dec_head3r_kernel<GV, DV, false>), lds);             \
                   hipLaunchKernelGGL((dec_head3r_kernel<GV, DV, false>), grid, block, lds, s, Wp, X3, K8p, vocab, a.RB, a); } \
        } while (0)
        if (a.RB == 1) MELLOW_H3R(1, MELLOW_H3R_D1); else if (a.RB == 2) MELLOW_H3R(2, MELLOW_H3R_D2); else MELLOW_H3R(4, MELLOW_H3R_D4);
#undef MELLOW_H3R
        return;
    }
    // (fp32 rows in xnF -- taps on caller rows, or a vocabulary the streaming form does not tile: the exact fp32 kernel below)
    const dim3 grid(vocab / 32, 1, a.RB), block(LM_WAVES * 64);
    const int mode = w8_mode(a, wscale);
    if (a.blk_live && mode == 2) hipLaunchKernelGGL((dec_fullk_kernel<OUT_LOGITS, true, 2>), grid, block, 0, s, Wp, (const float*)a.xnF, K8p, vocab, a, wscale);
    else if (a.blk_live && mode == 1) hipLaunchKernelGGL((dec_fullk_kernel<OUT_LOGITS, true, 1>), grid, block, 0, s, Wp, (const float*)a.xnF, K8p, vocab, a, wscale);
    else if (a.blk_live) hipLaunchKernelGGL((dec_fullk_kernel<OUT_LOGITS, true, 0>), grid, block, 0, s, Wp, (const float*)a.xnF, K8p, vocab, a, wscale);
    else if (mode == 2) hipLaunchKernelGGL((dec_fullk_kernel<OUT_LOGITS, false, 2>), grid, block, 0, s, Wp, (const float*)a.xnF, K8p, vocab, a, wscale);
    else if (mode == 1) hipLaunchKernelGGL((dec_fullk_kernel<OUT_LOGITS, false, 1>), grid, block, 0, s, Wp, (const float*)a.xnF, K8p, vocab, a, wscale);
    else hipLaunchKernelGGL((dec_fullk_kernel<OUT_LOGITS, false, 0>), grid, block, 0, s, Wp, (const float*)a.xnF, K8p, vocab, a, wscale);
}
// fp32 packed decode weight (P-layout: 32 rows per tile, or P16: 16 rows per tile; `slots` float4 slots per tile) -> one
// 4-byte word of four e4m3 values per slot + one scale per packed row (amax / 448): one workgroup per tile
__global__ __launch_bounds__(256) void pack_dec_fp8_kernel(const float4* __restrict__ Wp, int slots, int rows, uint32_t* __restrict__ out,
                                                           float* __restrict__ scale) {
    __shared__ unsigned amax_s[32];
    __shared__ float inv_s[32];
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (tid < 32) amax_s[tid] = 0u;
    __syncthreads();
    const float4* src = Wp + (int64_t)tile * slots;
    for (int i = tid; i < slots; i += 256) {
        const float4 v = src[i];
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        atomicMax(&amax_s[(i & 63) & (rows - 1)], __float_as_uint(m));          // non-negative floats order like their bit patterns
    }
    __syncthreads();
    if (tid < rows) {
        const float am = __uint_as_float(amax_s[tid]);
        const float sc = am > 0.f ? am / 448.0f : 1.0f;
        scale[(int64_t)tile * rows + tid] = sc;
        inv_s[tid] = 1.0f / sc;
    }
    __syncthreads();
    for (int i = tid; i < slots; i += 256) {
        const float4 v = src[i];
        const float inv = inv_s[(i & 63) & (rows - 1)];
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * inv, v.y * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * inv, v.w * inv, w, true);
        out[(int64_t)tile * slots + i] = (uint32_t)w;
    }
}
void launch_pack_dec_fp8(const float* Wp, int tiles, int slots_per_tile, int rows_per_tile, void* out, float* scale, hipStream_t s) {
    hipLaunchKernelGGL(pack_dec_fp8_kernel, dim3(tiles), dim3(256), 0, s, reinterpret_cast<const float4*>(Wp), slots_per_tile,
                       rows_per_tile, reinterpret_cast<uint32_t*>(out), scale);
}
void launch_dec_argmax(const DecArgs& a, int B, int n_tiles, int32_t* tokens, const float* embed, int write_x,
                       const LoopArgs& loop, hipStream_t s) {
    hipLaunchKernelGGL(dec_argmax_kernel, dim3(B), dim3(256), 0, s, (const float*)a.cand_val, (const int32_t*)a.cand_idx, n_tiles, a, tokens, embed,
                       write_x, loop);
}
void launch_dec_load_rows(const DecArgs& a, int B, const float* in, int64_t ld, const int32_t* row_ids, int T_last,
                          int n_src, hipStream_t s) {
    hipLaunchKernelGGL(dec_load_rows_kernel, dim3(B), dim3(192), 0, s, a, in, ld, row_ids, T_last, n_src);
}

// ---- full-row arg-max (mellow_argmax tap): torch.argmax tie rule, float4 loads --------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int V, int64_t ld,
                                                      int32_t* __restrict__ tokens) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float4* row = reinterpret_cast<const float4*>(logits + (int64_t)b * ld);
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < (V >> 2); i += 1024) {
        const float4 v = row[i];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (arg_better(vv[j], 4 * i + j, best, idx)) { best = vv[j]; idx = 4 * i + j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (arg_better(ov, oi, best, idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (arg_better(bv[w], bi[w], best, idx)) { best = bv[w]; idx = bi[w]; }
        tokens[b] = min(max(idx, 0), V - 1);
    }
}
void launch_argmax(const float* logits, int B, int V, int64_t ld, int32_t* tokens, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, s, logits, V, ld, tokens);
}

// P16 packing: out[nt][k16][lane][4] = W[nt*16 + (lane&15)][k16*16 + 4*(lane>>4) + j]
__global__ void pack_weight16_kernel(const float* __restrict__ w, int N, int K, float* __restrict__ out) {
    const int64_t total = (int64_t)(N / 16) * (K / 16) * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t tile = i >> 6;
        const int k16 = (int)(tile % (K / 16)), nt = (int)(tile / (K / 16));
        const int n = nt * 16 + (lane & 15), k0 = k16 * 16 + 4 * (lane >> 4);
        reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(w + (int64_t)n * K + k0);
    }
}
// P16N packing (f32x3 gate/up): out[nt][T][half][lane][4] = W[nt*16 + (lane&15)][T*32 + 8*(lane>>4) + 4*half + j]
__global__ void pack_weight16n_kernel(const float* __restrict__ w, int N, int K, float* __restrict__ out) {
    const int64_t total = (int64_t)(N / 16) * (K / 32) * 128;            // float4 units
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), half = (int)((i >> 6) & 1);
        const int64_t tile = i >> 7;
        const int T = (int)(tile % (K / 32)), nt = (int)(tile / (K / 32));
        const int n = nt * 16 + (lane & 15), k0 = T * 32 + 8 * (lane >> 4) + 4 * half;
        reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(w + (int64_t)n * K + k0);
    }
}
void launch_pack_weight16n(const float* w, int N, int K, float* out, hipStream_t s) {
    const int64_t total = (int64_t)(N / 16) * (K / 32) * 128;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight16n_kernel, dim3(blocks), dim3(256), 0, s, w, N, K, out);
}
void launch_pack_weight16(const float* w, int N, int K, float* out, hipStream_t s) {
    const int64_t total = (int64_t)(N / 16) * (K / 16) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight16_kernel, dim3(blocks), dim3(256), 0, s, w, N, K, out);
}

}  // namespace mellow
