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
                                                                float* __restrict__ o, i32x4* __restrict__ o_apb, int T, int Tmax) {
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

void launch_prefill_attention(const float* q, const float* k_cache, const float* v_cache, float* o, void* o_apb, int B, int T,
                              int Tmax, hipStream_t s) {
    const int qtiles = (T + 31) / 32;
    hipLaunchKernelGGL(prefill_attention_kernel, dim3(qtiles, 3, B), dim3(256), 0, s, q, k_cache, v_cache, o, reinterpret_cast<i32x4*>(o_apb), T, Tmax);
}

}  // namespace mellow
