// Causal GQA flash attention for LM prefill in exact fp32 on v_mfma_f32_32x32x2_f32 (replaces the
// eager/sdpa attention inside transformers' LlamaAttention for the 389-token prefix).
//
// One workgroup = (example b, kv head g, 32-query tile): 3 compute waves = the 3 query heads that share kv head g
// (GQA 9q/3kv), so each K/V tile is staged in LDS once for three heads, plus a 4th loader wave that stages the next tile.
//
// Register-resident softmax without any cross-lane shuffles of P: the score tile is computed
// TRANSPOSED, S^T = K Q^T (K tile = MFMA A operand, Q = B operand), so a lane owns ONE query column and
// 16 keys of the tile; the row max / row sum are 16 in-lane ops + one half-wave exchange.  The P^T
// accumulator registers are then directly the B operand of O^T += V^T P^T: k-step r pairs the keys
// {(r&3)+8(r>>2), +4} that the two half-waves hold in register r, and the V^T A-operand is read from
// the row-major V tile in LDS with consecutive lanes on consecutive dims (conflict-free).
//   MFMAs per 32x32 tile: 32 (QK^T over d=64) + 32 (PV, two 32-dim halves) = 64 x 64 cycles.
#include "common.h"
#include "kernels.h"

namespace mellow {

constexpr int PA_KT_STRIDE = 33;   // transposed K tile row stride (floats): conflict-free b32 reads/writes
// softmax weights on the hardware exp2 (x <= 0; ~1e-6 relative, inside fp32 summation-order noise); exp(-inf) = 0
__device__ __forceinline__ float pa_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__global__ __launch_bounds__(256) void prefill_attention_kernel(const float* __restrict__ q,
                                                                const float* __restrict__ k_cache,
                                                                const float* __restrict__ v_cache,
                                                                float* __restrict__ o, i32x4* __restrict__ o_apb, uint8_t* __restrict__ o_sc, int T, int Tmax) {
    // wave specialisation: waves 0..2 = the three query heads of kv head g (MFMA + softmax), wave 3 = loader: it owns
    // the global -> LDS staging (K transposed, V row-major) of the NEXT key tile into the other LDS stage while the
    // compute waves work, so they carry no staging registers (148 VGPRs -> three workgroups per CU) and never wait
    // for a load.  (Pairing a long and a short query tile per workgroup was measured too: slower, the hardware's
    // dynamic dispatch of 1248 unequal workgroups balances better than 672 equal ones.)
    __shared__ __attribute__((aligned(16))) float Kt[2][64 * PA_KT_STRIDE];   // [stage][d][key]
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * 64];             // [stage][key][d]
    // heavy tiles first: a query tile qt walks qt+1 key tiles (causal), so the long workgroups must not start last
    const int qt = (int)gridDim.x - 1 - (int)blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* kpage = k_cache + ((int64_t)b * 3 + g) * Tmax * 64;
    const float* vpage = v_cache + ((int64_t)b * 3 + g) * Tmax * 64;

    if (wave == 3) {
        // ---------------- loader wave: 512 float4 per operand per tile = 8 + 8 per lane ----------------
        f32x4 pk[8], pv[8];
        auto fetch = [&](int kt) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = lane + 64 * j, key = i >> 4, quad = i & 15;
                int t = kt * 32 + key;
                t = t < T ? t : T - 1;
                pk[j] = *reinterpret_cast<const f32x4*>(kpage + (int64_t)t * 64 + quad * 4);
                pv[j] = *reinterpret_cast<const f32x4*>(vpage + (int64_t)t * 64 + quad * 4);
            }
        };
        auto stage = [&](int st) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = lane + 64 * j, key = i >> 4, quad = i & 15;
                Kt[st][(quad * 4 + 0) * PA_KT_STRIDE + key] = pk[j].x;
                Kt[st][(quad * 4 + 1) * PA_KT_STRIDE + key] = pk[j].y;
                Kt[st][(quad * 4 + 2) * PA_KT_STRIDE + key] = pk[j].z;
                Kt[st][(quad * 4 + 3) * PA_KT_STRIDE + key] = pk[j].w;
                *reinterpret_cast<f32x4*>(&Vs[st][key * 64 + quad * 4]) = pv[j];
            }
        };
        fetch(0);
        stage(0);
        fetch(qt >= 1 ? 1 : 0);
        __syncthreads();                                   // tile 0 visible
        for (int kt = 0; kt <= qt; ++kt) {
            stage((kt + 1) & 1);                           // tile kt+1 (or a harmless re-read past the end) -> other stage
            fetch(kt + 2 <= qt ? kt + 2 : qt);
            __syncthreads();                               // compute waves are done with stage kt & 1; stage (kt+1) & 1 is visible
        }
        return;
    }

    // ---------------- compute waves ----------------
    const int hq = 3 * g + wave;
    const int h = lane >> 5, ql = lane & 31;
    const int q0 = qt * 32;
    const int qi = q0 + ql;                              // this lane's query position
    const int qc = qi < T ? qi : T - 1;
    // Q as MFMA B operand: step s holds Q[query][2s + h], pre-scaled by 1/8 (exact)
    float qreg[32];
    {
        const float* qrow = q + ((int64_t)b * T + qc) * 576 + hq * 64 + h;
#pragma unroll
        for (int s = 0; s < 32; ++s) qreg[s] = qrow[2 * s] * 0.125f;
    }
    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    __syncthreads();                                       // tile 0 staged by the loader

    for (int kt = 0; kt <= qt; ++kt) {
        const int k0 = kt * 32;
        const float* Kc = Kt[kt & 1];
        const float* Vc = Vs[kt & 1];
        // S^T[key][query] = sum_d K[key][d] Q[query][d]
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a = Kc[(2 * s + h) * PA_KT_STRIDE + ql];
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], S, 0, 0, 0);
        }
        // lane: query ql, keys k0 + (r&3) + 8(r>>2) + 4h
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key > qi) S[r] = -INFINITY;          // causal mask (only bites on the diagonal tile)
            tmax = fmaxf(tmax, S[r]);
        }
        tmax = half_max(tmax);                       // the other 16 keys of this query live in lane ^ 32
        const float m_new = fmaxf(m_run, tmax);      // finite: key k0 <= q0 <= qi is never masked
        const float alpha = pa_exp(m_run - m_new);   // exp(-inf) = 0 on the first tile
        float rsum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = pa_exp(S[r] - m_new);
            rsum += p[r];
        }
        rsum = half_sum(rsum);
        l_run = l_run * alpha + rsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
        // O^T[d][query] += sum_key V[key][d] P^T[key][query]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float a0 = Vc[key * 64 + ql];
            const float a1 = Vc[key * 64 + 32 + ql];
            O0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, p[r], O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, p[r], O1, 0, 0, 0);
        }
        __syncthreads();                                   // done with stage kt & 1; the next tile is visible
    }
    if (qi < T) {
        const float inv = 1.0f / l_run;
        if (o_sc) {       // fp8 mode: the o_proj is gemm_mx8_kernel -- the row as MXFP8 in AMX order (K = 576: blocks 2 hq, 2 hq + 1)
            const int64_t m = (int64_t)b * T + qi;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = O0[r] * inv;
            amx_store_block(o_apb, o_sc, m, hq * 2, 9, 3, v, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = O1[r] * inv;
            amx_store_block(o_apb, o_sc, m, hq * 2 + 1, 9, 3, v, h);
        } else
        if (o_apb) {      // the o_proj is an x3q GEMM: write the row pre-split in APB order (K = 576: 72 column octets)
            const int64_t m = (int64_t)b * T + qi;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float X[4], Y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { X[j] = O0[8 * gp + j] * inv; Y[j] = O0[8 * gp + 4 + j] * inv; }
                apb_store_quads(o_apb, m, hq * 8 + 2 * gp, 36, X, Y, h);
#pragma unroll
                for (int j = 0; j < 4; ++j) { X[j] = O1[8 * gp + j] * inv; Y[j] = O1[8 * gp + 4 + j] * inv; }
                apb_store_quads(o_apb, m, hq * 8 + 4 + 2 * gp, 36, X, Y, h);
            }
        } else {
            float* orow = o + ((int64_t)b * T + qi) * 576 + hq * 64;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d = 8 * gq + 4 * h;
                *reinterpret_cast<float4*>(orow + d) =
                    make_float4(O0[4 * gq] * inv, O0[4 * gq + 1] * inv, O0[4 * gq + 2] * inv, O0[4 * gq + 3] * inv);
                *reinterpret_cast<float4*>(orow + 32 + d) =
                    make_float4(O1[4 * gq] * inv, O1[4 * gq + 1] * inv, O1[4 * gq + 2] * inv, O1[4 * gq + 3] * inv);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// The same attention with both matrix products on the bf16 pipe, fp32-accurate (the f32x3 mode): K, Q, P and V are
// each split EXACTLY into three bf16 pieces and the six largest partial products run on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (24 + 24 MFMAs of 32 cycles per 32 x 32 tile instead of 32 + 32 of 64).  Same workgroup shape and the same
// transposed, register-resident softmax as above; what changes is the operand plumbing:
//   * the loader wave splits K and V while it stages them and writes the pieces in MFMA FRAGMENT order (one 16-byte slot = the
//     8 bf16 one lane feeds to one MFMA), so a compute wave's operand fetch is one conflict-free ds_read_b128 per piece:
//       K image [d-step s 0..3][piece][lane = key + 32 * half]        8 bf16 = d 16 s + 8 half + (0..7)
//       V image [key-step t 0..1][d-half][piece][lane = d + 32 * h]   8 bf16 = keys 16 t + (j&3) + 8 (j>>2) + 4 h, j = 0..7
//     The key order inside a V slot is the order in which the score accumulators hold a query's 16 keys of a step, so the P
//     registers (split in place) are the B operand as they are -- no cross-lane movement of P, no transposition of V by the
//     compute waves.  The loader lane that owns (d-quad, t, h) loads exactly those 8 keys, so V is transposed in registers.
//   * V slots are stored at lane ^ ((lane >> 3) & 3): a lane group of the 16-byte store then covers 8 distinct bank quads
//     (the plain order would put its 8 lanes on 2), and the read groups of ds_read_b128 stay conflict-free.
// ----------------------------------------------------------------------------------------------------------------------------
#ifndef MELLOW_PAX_LOADERS
#define MELLOW_PAX_LOADERS 1      // 1: one wave stages K and V (1.88 ms per pass); 2: a wave each (2.34 ms, same box)
#endif
constexpr int PAX_LOADERS = MELLOW_PAX_LOADERS, PAX_THREADS = (3 + PAX_LOADERS) * 64;
#ifndef MELLOW_PAX_MINW
#define MELLOW_PAX_MINW 2      // waves per SIMD the register allocation must allow (2 workgroups of 4 waves per CU)
#endif
constexpr int PAX_K_SLOTS = 4 * 3 * 64, PAX_V_SLOTS = 2 * 2 * 3 * 64;          // per stage at NP = 3 pieces (NP = 1: a third)
#ifndef MELLOW_PAX_ABL
#define MELLOW_PAX_ABL 0      // developer ablations (wrong results, timing only; tools/microbench/prefill_attn_bench.hip): 1 loader stores
#endif                        // unsplit bits, 2 no softmax arithmetic, 4 no split of P, 8 no score MFMAs, 16 no PV MFMAs, 32 loader fetches tile 0 only
__device__ __forceinline__ int pax_sw(int l) { return l ^ ((l >> 3) & 3); }
#define PAX_MFMA(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)
// the six partial products of (a0 + a1 + a2)(b0 + b1 + b2) that are not below 2^-23 of the result, smallest first
#define PAX_6(A0, A1, A2, B0, B1, B2, ACC) \
    do { PAX_MFMA(A2, B0, ACC); PAX_MFMA(A1, B1, ACC); PAX_MFMA(A0, B2, ACC); PAX_MFMA(A1, B0, ACC); PAX_MFMA(A0, B1, ACC); PAX_MFMA(A0, B0, ACC); } while (0)

// NP = 3: the exact 3-way split (f32x3 mode, fp32-accurate).  NP = 1 (fp8 mode, round 6): K, Q, P and V rounded once to bf16 -- the
// plain bf16 flash attention on the same plumbing (8 MFMAs per 32 x 32 tile instead of 48, no split arithmetic): BASELINE config 5
// quantises every GEMM operand to e4m3 (3 mantissa bits) around it, so an 8-bit-mantissa attention is not what bounds its accuracy
// (agreement figures: DESIGN.md 6b); softmax statistics, accumulation and the K/V pages stay fp32.
template <int NP>
__device__ __forceinline__ void pax_split(const float (&v)[8], i32x4 (&p)[NP]) {
    if constexpr (NP == 3) split8(v, p[0], p[1], p[2]);
    else {
        bf16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = static_cast<__bf16>(v[j]);
        p[0] = __builtin_bit_cast(i32x4, h);
    }
}
// P16 (NP = 1 only; fp8 mode with bf16 K/V pages written by the q/k/v epilogue): the loader wave copies -- a 16-byte load of a key's
// 8 dims IS a K slot, and the V slots are eight 8-byte loads transposed with v_perm_b32; no conversion, half the page bytes.
template <int NP, bool P16>
__global__ __launch_bounds__(PAX_THREADS, MELLOW_PAX_MINW) void prefill_attention_x3_kernel(const float* __restrict__ q, const float* __restrict__ k_cache,
                                                                   const float* __restrict__ v_cache, float* __restrict__ o,
                                                                   i32x4* __restrict__ o_apb, uint8_t* __restrict__ o_sc, int T, int Tmax) {
    __shared__ i32x4 Kp[2][4 * NP * 64];
    __shared__ i32x4 Vp[2][2 * 2 * NP * 64];
    const int qt = (int)gridDim.x - 1 - (int)blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* kpage = k_cache + ((int64_t)b * 3 + g) * Tmax * 64;
    const float* vpage = v_cache + ((int64_t)b * 3 + g) * Tmax * 64;

    if (wave >= 3) {
        // ---------------- loader waves: wave 3 stages K, wave 4 stages V (PAX_LOADERS == 2; one wave does both otherwise):
        //                  8 float4 per lane per tile and operand, split, 12 sixteen-byte LDS stores ----------------
        const bool doK = PAX_LOADERS == 1 || wave == 3, doV = PAX_LOADERS == 1 || wave == 4;
        const int klo = lane & 7, oct = lane >> 3;               // K: keys klo + 8 i (i = 0..3), the 8 dims 8 oct .. 8 oct + 7
        const int q4 = lane & 15, t2 = (lane >> 4) & 1, hv = lane >> 5;     // V: dims 4 q4 .. + 3, key step t2, key half hv
        if constexpr (P16) {
            static_assert(NP == 1, "bf16 pages feed the bf16-once form only");
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            const uint16_t* kp16 = reinterpret_cast<const uint16_t*>(k_cache) + ((int64_t)b * 3 + g) * Tmax * 64;
            const uint16_t* vp16 = reinterpret_cast<const uint16_t*>(v_cache) + ((int64_t)b * 3 + g) * Tmax * 64;
            i32x4 rk[4];
            u32x2_ rv[8];
            auto fetch16 = [&](int kt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int t = kt * 32 + klo + 8 * i;
                    t = t < T ? t : T - 1;
                    rk[i] = *reinterpret_cast<const i32x4*>(kp16 + (int64_t)t * 64 + oct * 8);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int t = kt * 32 + 16 * t2 + (j & 3) + 8 * (j >> 2) + 4 * hv;
                    t = t < T ? t : T - 1;
                    rv[j] = *reinterpret_cast<const u32x2_*>(vp16 + (int64_t)t * 64 + q4 * 4);
                }
            };
            auto stage16 = [&](int st) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Kp[st][(oct >> 1) * 64 + klo + 8 * i + 32 * (oct & 1)] = rk[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // dim 4 q4 + e of keys j = 0..7: half (e & 1) of dword (e >> 1) of every load
                    i32x4 pc;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const unsigned lo = rv[2 * jj][e >> 1], hi = rv[2 * jj + 1][e >> 1];
                        pc[jj] = (int)((e & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u));
                    }
                    const int d = 4 * q4 + e;
                    Vp[st][(t2 * 2 + (d >> 5)) * 64 + pax_sw((d & 31) + 32 * hv)] = pc;
                }
            };
            fetch16(0);
            stage16(0);
            fetch16(qt >= 1 ? 1 : 0);
            __syncthreads();
            for (int kt = 0; kt <= qt; ++kt) {
                stage16((kt + 1) & 1);
                fetch16(kt + 2 <= qt ? kt + 2 : qt);
                __syncthreads();
            }
            return;
        }
        f32x4 pk[8], pv[8];
        auto fetch = [&](int kt) {
            if (doK)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int t = kt * 32 + klo + 8 * i;
                t = t < T ? t : T - 1;
                pk[2 * i] = *reinterpret_cast<const f32x4*>(kpage + (int64_t)t * 64 + oct * 8);
                pk[2 * i + 1] = *reinterpret_cast<const f32x4*>(kpage + (int64_t)t * 64 + oct * 8 + 4);
            }
            if (doV)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int t = kt * 32 + 16 * t2 + (j & 3) + 8 * (j >> 2) + 4 * hv;
                t = t < T ? t : T - 1;
                pv[j] = *reinterpret_cast<const f32x4*>(vpage + (int64_t)t * 64 + q4 * 4);
            }
        };
        auto stage = [&](int st) {
            if (doK)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v[8] = {pk[2 * i][0], pk[2 * i][1], pk[2 * i][2], pk[2 * i][3],
                                    pk[2 * i + 1][0], pk[2 * i + 1][1], pk[2 * i + 1][2], pk[2 * i + 1][3]};
                i32x4 pc[NP];
                if (MELLOW_PAX_ABL & 1) { for (int z = 0; z < NP; ++z) pc[z] = i32x4{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])}; } else pax_split<NP>(v, pc);
                i32x4* dst = &Kp[st][((oct >> 1) * NP) * 64 + klo + 8 * i + 32 * (oct & 1)];
#pragma unroll
                for (int z = 0; z < NP; ++z) dst[z * 64] = pc[z];
            }
            if (doV)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v[8] = {pv[0][e], pv[1][e], pv[2][e], pv[3][e], pv[4][e], pv[5][e], pv[6][e], pv[7][e]};
                i32x4 pc[NP];
                if (MELLOW_PAX_ABL & 1) { for (int z = 0; z < NP; ++z) pc[z] = i32x4{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])}; } else pax_split<NP>(v, pc);
                const int d = 4 * q4 + e;
                i32x4* dst = &Vp[st][((t2 * 2 + (d >> 5)) * NP) * 64 + pax_sw((d & 31) + 32 * hv)];
#pragma unroll
                for (int z = 0; z < NP; ++z) dst[z * 64] = pc[z];
            }
        };
        fetch(0);
        stage(0);
        fetch(qt >= 1 ? 1 : 0);
        __syncthreads();                                   // tile 0 visible
        for (int kt = 0; kt <= qt; ++kt) {
            stage((kt + 1) & 1);                           // tile kt+1 (or a harmless re-read past the end) -> other stage
            if (!(MELLOW_PAX_ABL & 32)) fetch(kt + 2 <= qt ? kt + 2 : qt);
            __syncthreads();                               // compute waves are done with stage kt & 1; stage (kt+1) & 1 is visible
        }
        return;
    }

    // ---------------- compute waves ----------------
    const int hq = 3 * g + wave;
    const int h = lane >> 5, ql = lane & 31;
    const int q0 = qt * 32;
    const int qi = q0 + ql;
    const int qc = qi < T ? qi : T - 1;
    // Q as B operand of the bf16 MFMA: step s, lane (query ql, half h) holds d = 16 s + 8 h + (0..7), pre-scaled by 1/8 (exact)
    i32x4 qp[4][NP];
    if constexpr (P16) {      // q arrives as bf16 rows (rounded once by the q/k/v epilogue); x 1/8 is exact in bf16
        const uint16_t* qrow = reinterpret_cast<const uint16_t*>(q) + ((int64_t)b * T + qc) * 576 + hq * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const i32x4 r = *reinterpret_cast<const i32x4*>(qrow + 16 * s);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float((unsigned)r[j] << 16) * 0.125f; v[2 * j + 1] = __uint_as_float((unsigned)r[j] & 0xffff0000u) * 0.125f; }
            pax_split<NP>(v, qp[s]);
        }
    } else {
        const float* qrow = q + ((int64_t)b * T + qc) * 576 + hq * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + 16 * s), c = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
            const float v[8] = {a.x * 0.125f, a.y * 0.125f, a.z * 0.125f, a.w * 0.125f, c.x * 0.125f, c.y * 0.125f, c.z * 0.125f, c.w * 0.125f};
            pax_split<NP>(v, qp[s]);
        }
    }
    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int vl = pax_sw(lane);
    __syncthreads();                                       // tile 0 staged by the loader

    for (int kt = 0; kt <= qt; ++kt) {
        const int k0 = kt * 32;
        const i32x4* Kc = Kp[kt & 1] + lane;
        const i32x4* Vc = Vp[kt & 1] + vl;
        // two independent accumulation chains (d-steps 0,1 | 2,3), added at the end: a dependent 32x32x16 MFMA cannot issue
        // before its predecessor's accumulator is back, so one chain of 24 runs at the latency, not at the issue rate
        f32x16 S, S2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; S2[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (NP == 1) {
                const i32x4 a0 = Kc[s * 64], c0 = Kc[(s + 2) * 64];
                if (MELLOW_PAX_ABL & 8) { S[0] += __int_as_float(a0[0] ^ c0[0]); continue; }
                PAX_MFMA(a0, qp[s][0], S);      PAX_MFMA(c0, qp[s + 2][0], S2);
            } else {
            const i32x4 a0 = Kc[(s * 3 + 0) * 64], a1 = Kc[(s * 3 + 1) * 64], a2 = Kc[(s * 3 + 2) * 64];
            const i32x4 c0 = Kc[((s + 2) * 3 + 0) * 64], c1 = Kc[((s + 2) * 3 + 1) * 64], c2 = Kc[((s + 2) * 3 + 2) * 64];
            if (MELLOW_PAX_ABL & 8) { S[0] += __int_as_float(a0[0] ^ a1[1] ^ a2[2] ^ c0[0] ^ c1[1] ^ c2[2]); continue; }
            PAX_MFMA(a2, qp[s][0], S);      PAX_MFMA(c2, qp[s + 2][0], S2);
            PAX_MFMA(a1, qp[s][1], S);      PAX_MFMA(c1, qp[s + 2][1], S2);
            PAX_MFMA(a0, qp[s][2], S);      PAX_MFMA(c0, qp[s + 2][2], S2);
            PAX_MFMA(a1, qp[s][0], S);      PAX_MFMA(c1, qp[s + 2][0], S2);
            PAX_MFMA(a0, qp[s][1], S);      PAX_MFMA(c0, qp[s + 2][1], S2);
            PAX_MFMA(a0, qp[s][0], S);      PAX_MFMA(c0, qp[s + 2][0], S2);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] += S2[r];
        // lane: query ql, keys k0 + (r&3) + 8(r>>2) + 4h
        float p[16];
        if (MELLOW_PAX_ABL & 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = S[r];
            l_run += S[0];
        } else {
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key > qi) S[r] = -INFINITY;          // causal mask (only bites on the diagonal tile)
            tmax = fmaxf(tmax, S[r]);
        }
        tmax = half_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = pa_exp(m_run - m_new);
        float rsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = pa_exp(S[r] - m_new);
            rsum += p[r];
        }
        rsum = half_sum(rsum);
        l_run = l_run * alpha + rsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
        }
        // O^T[d][query] += sum_key V[key][d] P^T[key][query]: registers 8 t .. 8 t + 7 of P are the k-slots of key step t
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float v[8] = {p[8 * t], p[8 * t + 1], p[8 * t + 2], p[8 * t + 3], p[8 * t + 4], p[8 * t + 5], p[8 * t + 6], p[8 * t + 7]};
            i32x4 bq[NP];
            if (MELLOW_PAX_ABL & 4) { for (int z = 0; z < NP; ++z) bq[z] = i32x4{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])}; }
            else pax_split<NP>(v, bq);
            if constexpr (NP == 1) {
                const i32x4 a0 = Vc[(t * 2 + 0) * 64], c0 = Vc[(t * 2 + 1) * 64];
                if (MELLOW_PAX_ABL & 16) { O0[0] += __int_as_float(a0[0] ^ bq[0][0]); O1[0] += __int_as_float(c0[0]); continue; }
                PAX_MFMA(a0, bq[0], O0);      PAX_MFMA(c0, bq[0], O1);
            } else {       // the two d-halves are independent chains: interleave them
                const i32x4 b0 = bq[0], b1 = bq[1], b2 = bq[2];
                const i32x4 a0 = Vc[((t * 2 + 0) * 3 + 0) * 64], a1 = Vc[((t * 2 + 0) * 3 + 1) * 64], a2 = Vc[((t * 2 + 0) * 3 + 2) * 64];
                const i32x4 c0 = Vc[((t * 2 + 1) * 3 + 0) * 64], c1 = Vc[((t * 2 + 1) * 3 + 1) * 64], c2 = Vc[((t * 2 + 1) * 3 + 2) * 64];
                if (MELLOW_PAX_ABL & 16) { O0[0] += __int_as_float(a0[0] ^ a1[1] ^ a2[2] ^ b0[0] ^ b1[1] ^ b2[2]); O1[0] += __int_as_float(c0[0] ^ c1[1] ^ c2[2]); continue; }
                PAX_MFMA(a2, b0, O0);      PAX_MFMA(c2, b0, O1);
                PAX_MFMA(a1, b1, O0);      PAX_MFMA(c1, b1, O1);
                PAX_MFMA(a0, b2, O0);      PAX_MFMA(c0, b2, O1);
                PAX_MFMA(a1, b0, O0);      PAX_MFMA(c1, b0, O1);
                PAX_MFMA(a0, b1, O0);      PAX_MFMA(c0, b1, O1);
                PAX_MFMA(a0, b0, O0);      PAX_MFMA(c0, b0, O1);
            }
        }
        __syncthreads();                                   // done with stage kt & 1; the next tile is visible
    }
    if (qi < T) {
        const float inv = 1.0f / l_run;
        if (o_sc) {       // fp8 mode: the o_proj is gemm_mx8_kernel -- the row as MXFP8 in AMX order (K = 576: blocks 2 hq, 2 hq + 1)
            const int64_t m = (int64_t)b * T + qi;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = O0[r] * inv;
            amx_store_block(o_apb, o_sc, m, hq * 2, 9, 3, v, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = O1[r] * inv;
            amx_store_block(o_apb, o_sc, m, hq * 2 + 1, 9, 3, v, h);
        } else
        if (o_apb) {
            const int64_t m = (int64_t)b * T + qi;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float X[4], Y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { X[j] = O0[8 * gp + j] * inv; Y[j] = O0[8 * gp + 4 + j] * inv; }
                apb_store_quads(o_apb, m, hq * 8 + 2 * gp, 36, X, Y, h);
#pragma unroll
                for (int j = 0; j < 4; ++j) { X[j] = O1[8 * gp + j] * inv; Y[j] = O1[8 * gp + 4 + j] * inv; }
                apb_store_quads(o_apb, m, hq * 8 + 4 + 2 * gp, 36, X, Y, h);
            }
        } else {
            float* orow = o + ((int64_t)b * T + qi) * 576 + hq * 64;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d = 8 * gq + 4 * h;
                *reinterpret_cast<float4*>(orow + d) =
                    make_float4(O0[4 * gq] * inv, O0[4 * gq + 1] * inv, O0[4 * gq + 2] * inv, O0[4 * gq + 3] * inv);
                *reinterpret_cast<float4*>(orow + 32 + d) =
                    make_float4(O1[4 * gq] * inv, O1[4 * gq + 1] * inv, O1[4 * gq + 2] * inv, O1[4 * gq + 3] * inv);
            }
        }
    }
}

// x3 = the bf16-split kernel (the engine's f32x3 mode), else exact fp32 MFMA
void launch_prefill_attention(const float* q, const float* k_cache, const float* v_cache, float* o, void* o_apb, int B, int T,
                              int Tmax, bool x3, hipStream_t s, void* o_scales, bool bf16_once, bool pages16) {
    const int qtiles = (T + 31) / 32;
    uint8_t* sc = reinterpret_cast<uint8_t*>(o_scales);
    if (x3 && bf16_once && pages16) hipLaunchKernelGGL((prefill_attention_x3_kernel<1, true>), dim3(qtiles, 3, B), dim3(PAX_THREADS), 0, s, q, k_cache, v_cache, o, reinterpret_cast<i32x4*>(o_apb), sc, T, Tmax);
    else if (x3 && bf16_once) hipLaunchKernelGGL((prefill_attention_x3_kernel<1, false>), dim3(qtiles, 3, B), dim3(PAX_THREADS), 0, s, q, k_cache, v_cache, o, reinterpret_cast<i32x4*>(o_apb), sc, T, Tmax);
    else if (x3) hipLaunchKernelGGL((prefill_attention_x3_kernel<3, false>), dim3(qtiles, 3, B), dim3(PAX_THREADS), 0, s, q, k_cache, v_cache, o, reinterpret_cast<i32x4*>(o_apb), sc, T, Tmax);
    else hipLaunchKernelGGL(prefill_attention_kernel, dim3(qtiles, 3, B), dim3(256), 0, s, q, k_cache, v_cache, o, reinterpret_cast<i32x4*>(o_apb), sc, T, Tmax);
}

}  // namespace mellow
