// KV-cached decode step kernels (replace the reference's full re-forward per token, wrapper.py:216-249;
// SURVEY.md §8a A15/A16).  All HBM-bound: per step they stream the LM weights once (538 MB fp32) and
// the KV pages of every live example, so the design goal is coalesced 1 KiB-per-wave-instruction
// streams and as few dependent launches as possible.
//
//  skinny_splitk  P[kc][32][N] = X[32][K-slice] W^T on v_mfma_f32_32x32x2_f32.  The 32 batch rows are exactly
//                 one MFMA tile; weights come straight from HBM in P-layout (kernels.h) — one lane-linear
//                 float4 per lane = four MFMAs, no LDS round trip for the streamed operand.
//  rows_finish    residual add of the split-K slabs + LlamaRMSNorm, one workgroup per batch row.
//  decode_attn    one workgroup per (example, kv head): RoPE of the new q/k, append K/V to the pages,
//                 scores for the 3 query heads sharing the KV head from ONE pass over the K page
//                 (GQA), block softmax in LDS, one pass over the V page.
#include "common.h"
#include "kernels.h"

namespace mellow {

// ----------------------------------------------------------------------------------------------------
// split-K skinny GEMM.  grid (n-tiles, KC, row-blocks), 4 waves; wave w of k-chunk kc owns KPW
// consecutive 8-wide k-tiles and issues ALL of its loads (KPW x 1 KiB of weights) before the first
// MFMA, so a launch has the whole weight matrix in flight at once (a decode layer's matrices are only
// 1.3-7 MB: latency, not bandwidth, is the enemy).  The 4 waves reduce through LDS and the block writes
// one deterministic partial slab P[kc][row][n]; slabs are summed by the row-parallel finish kernel
// (fixed order -> bit-reproducible, no atomics).
// ----------------------------------------------------------------------------------------------------
template <int KPW, int PRO, int KCIN>
__global__ __launch_bounds__(256) void skinny_splitk_kernel(const SkinnyArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];  // 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, kc = blockIdx.y, rb = blockIdx.z;
    const int m = lane & 31, h = lane >> 5;
    const int k8_0 = (kc * 4 + wave) * KPW;
    const float4* wp = reinterpret_cast<const float4*>(a.Wp) + ((int64_t)nt * a.K8p + k8_0) * 64 + lane;
    float4 w[KPW], x[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) w[i] = wp[i * 64];
    if (PRO == PRO_SWIGLU) {
        // X = raw gate/up partial slabs [KCin][rows][ldx] in pair-interleaved 64-column groups:
        // gate at (k/32)*64 + k%32, up 32 columns later;  x = silu(sum gate) * (sum up)
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = (k8_0 + i) * 8 + 4 * h;
            const int c = ((k >> 5) << 6) + (k & 31);
            float4 g4[KCIN], u4[KCIN];
#pragma unroll
            for (int s = 0; s < KCIN; ++s) {   // compile-time slab count: all loads issue together
                const float* xr = a.X + ((int64_t)s * a.slab_rows + rb * 32 + m) * a.ldx + c;
                g4[s] = *reinterpret_cast<const float4*>(xr);
                u4[s] = *reinterpret_cast<const float4*>(xr + 32);
            }
            float4 gt = g4[0], up = u4[0];
#pragma unroll
            for (int s = 1; s < KCIN; ++s) {
                gt.x += g4[s].x; gt.y += g4[s].y; gt.z += g4[s].z; gt.w += g4[s].w;
                up.x += u4[s].x; up.y += u4[s].y; up.z += u4[s].z; up.w += u4[s].w;
            }
            x[i].x = __fmul_rn(siluf_(gt.x), up.x); x[i].y = __fmul_rn(siluf_(gt.y), up.y);
            x[i].z = __fmul_rn(siluf_(gt.z), up.z); x[i].w = __fmul_rn(siluf_(gt.w), up.w);
        }
    } else {
        const float* xr = a.X + (int64_t)(rb * 32 + m) * a.ldx + k8_0 * 8 + 4 * h;
#pragma unroll
        for (int i = 0; i < KPW; ++i) x[i] = *reinterpret_cast<const float4*>(xr + i * 8);
    }
    // keep every load above this point: the whole K-slice of the wave is in flight before the first MFMA
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i].x, x[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i].y, x[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i].z, x[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i].w, x[i].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    // thread -> (row mm, column group gq, half hh): 4 consecutive columns n = 8*gq + 4*hh + j  <-  r = 4*gq + j
    const int mm = tid & 31, hh = (tid >> 5) & 1, gq = tid >> 6;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * gq + j;
        v[j] = (red[(0 * 16 + r) * 64 + mm + 32 * hh] + red[(1 * 16 + r) * 64 + mm + 32 * hh]) +
               (red[(2 * 16 + r) * 64 + mm + 32 * hh] + red[(3 * 16 + r) * 64 + mm + 32 * hh]);
    }
    const int n = nt * 32 + 8 * gq + 4 * hh;
    const int row = rb * 32 + mm;
    if (a.Y && n < a.N)
        *reinterpret_cast<float4*>(a.Y + ((int64_t)kc * a.slab_rows_out + row) * a.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
    if (a.cand_val) {
        // fused arg-max candidates (lm_head, KC == 1): best (value, lowest index) of this 32-column tile per row
        __syncthreads();
        float bv = v[0];
        int bi = n;
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (v[j] > bv) { bv = v[j]; bi = n + j; }
        red[tid] = bv;
        reinterpret_cast<int*>(red)[256 + tid] = bi;
        __syncthreads();
        if (tid < 32) {
            float best = red[tid];
            int idx = reinterpret_cast<int*>(red)[256 + tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const float ov = red[tid + 32 * q];
                const int oi = reinterpret_cast<int*>(red)[256 + tid + 32 * q];
                if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
            }
            const int64_t o = (int64_t)(rb * 32 + tid) * gridDim.x + nt;
            a.cand_val[o] = best;
            a.cand_idx[o] = idx;
        }
    }
}

template <int KPW>
static void launch_skinny_kpw(const SkinnyArgs& a, int KC, hipStream_t s) {
    const dim3 grid((a.N + 31) / 32, KC, a.RB);
    if (a.pro == PRO_SWIGLU) hipLaunchKernelGGL((skinny_splitk_kernel<KPW, PRO_SWIGLU, SK_KC_GU>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((skinny_splitk_kernel<KPW, PRO_PLAIN, 1>), grid, dim3(256), 0, s, a);
}

int skinny_kc_for(int K) {
    // k8 tiles = K/8 must equal KC * 4 waves * KPW with KPW in {2,3,6,9,18}
    const int K8 = K / 8;
    if (K8 == 72) return 9;     // K = 576  -> KPW 2
    if (K8 == 192) return 16;   // K = 1536 -> KPW 3
    return 0;
}

void launch_skinny(const SkinnyArgs& a, hipStream_t s) {
    const int K8 = a.K / 8;
    const int KC = a.kc_out;
    const int kpw = K8 / (KC * 4);
    switch (kpw) {
        case 2: launch_skinny_kpw<2>(a, KC, s); break;
        case 3: launch_skinny_kpw<3>(a, KC, s); break;
        case 6: launch_skinny_kpw<6>(a, KC, s); break;
        case 9: launch_skinny_kpw<9>(a, KC, s); break;
        case 18: launch_skinny_kpw<18>(a, KC, s); break;
        default: break;  // validated by the engine
    }
}

// ---- row-parallel finish: x_out = x_in + sum_kc P[kc]; optional RMSNorm -> xn -------------------------------
// One workgroup per batch row; replaces the residual adds and LlamaRMSNorm of the reference layer.
// KC is a compile-time constant so the KC slab loads are issued together (one L2 round trip, not KC).
template <int KC>
__global__ __launch_bounds__(192) void rows_finish_kernel(const float* __restrict__ x_in, const float* __restrict__ P,
                                                          int64_t slab_stride, float* __restrict__ x_out,
                                                          const float* __restrict__ norm_w, float eps,
                                                          float* __restrict__ xn, int C, int32_t* inc_word) {
    __shared__ float part[3];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = C >> 2;
    // the first finish of a decode step advances the position word: no kernel of the previous step reads it any
    // more, and this kernel does not read it
    if (inc_word && row == 0 && tid == 0) *inc_word = *inc_word + 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < nv) {
        float4 p[KC > 0 ? KC : 1];
        v = reinterpret_cast<const float4*>(x_in + (int64_t)row * C)[tid];
#pragma unroll
        for (int s = 0; s < KC; ++s) p[s] = reinterpret_cast<const float4*>(P + s * slab_stride + (int64_t)row * C)[tid];
#pragma unroll
        for (int s = 0; s < KC; ++s) { v.x += p[s].x; v.y += p[s].y; v.z += p[s].z; v.w += p[s].w; }
        if (x_out && KC > 0) reinterpret_cast<float4*>(x_out + (int64_t)row * C)[tid] = v;
    }
    if (norm_w) {
        float ss = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        ss = wave_sum(ss);
        if ((tid & 63) == 0) part[tid >> 6] = ss;
        __syncthreads();
        const float r = 1.0f / sqrtf((part[0] + part[1] + part[2]) / (float)C + eps);
        if (tid < nv) {
            const float4 w = reinterpret_cast<const float4*>(norm_w)[tid];
            float4 y;
            y.x = __fmul_rn(w.x, __fmul_rn(v.x, r)); y.y = __fmul_rn(w.y, __fmul_rn(v.y, r));
            y.z = __fmul_rn(w.z, __fmul_rn(v.z, r)); y.w = __fmul_rn(w.w, __fmul_rn(v.w, r));
            reinterpret_cast<float4*>(xn + (int64_t)row * C)[tid] = y;
        }
    }
}
void launch_rows_finish(const float* x_in, const float* P, int kc, int64_t slab_stride, float* x_out,
                        const float* norm_w, float eps, float* xn, int rows, int C, int32_t* inc_word, hipStream_t s) {
#define MELLOW_RF(KC) hipLaunchKernelGGL((rows_finish_kernel<KC>), dim3(rows), dim3(192), 0, s, x_in, P, slab_stride, x_out, norm_w, eps, xn, C, inc_word)
    switch (kc) {
        case 0: MELLOW_RF(0); break;
        case SK_KC_O: MELLOW_RF(SK_KC_O); break;       // == SK_KC_QKV
        case SK_KC_DOWN: MELLOW_RF(SK_KC_DOWN); break;
        default: break;   // validated by the engine
    }
#undef MELLOW_RF
}

// ----------------------------------------------------------------------------------------------------
// decode attention (flash-decoding inside one workgroup).  grid (kv_heads=3, B); 1024 threads = 16 waves.
//   qkv split-K slabs P[kc][rows][960] (q: 9 heads x 64 | k: 3 x 64 | v: 3 x 64, no RoPE yet), summed here.
//   position of the new token = *d_pos (number of keys already in the pages).
// Each wave owns an interleaved set of 4-key groups: it issues the K and V loads of up to DA_G groups at
// once (2 KiB per group in flight), computes the 3 GQA heads' scores from ONE pass over K, does its own
// softmax statistics (max / sum) in registers, accumulates P.V, and only at the end the 16 waves'
// (m, l, o) triples are combined through LDS — one barrier instead of a block-wide softmax.
// ----------------------------------------------------------------------------------------------------
constexpr int DA_WAVES = 16;
constexpr int DA_G = 8;          // 4-key groups in flight per wave per chunk (512 keys per chunk per block)

template <int KC>
__global__ __launch_bounds__(DA_WAVES * 64) void decode_attention_kernel(
    const float* __restrict__ qkv_parts, int64_t slab_stride, float* __restrict__ k_cache,
    float* __restrict__ v_cache, const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
    const int32_t* __restrict__ d_pos, float* __restrict__ o, int Tmax) {
    __shared__ __attribute__((aligned(16))) float qs[3 * 64];            // RoPE'd, pre-scaled q
    __shared__ __attribute__((aligned(16))) float knew[64], vnew[64];
    __shared__ __attribute__((aligned(16))) float ored[DA_WAVES * 3 * 64];
    __shared__ float mred[DA_WAVES * 3], lred[DA_WAVES * 3];

    const int g = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = *d_pos;          // keys 0..pos-1 are cached; the new key is key `pos`
    const float* row = qkv_parts + (int64_t)b * 960;
    float* kpage = k_cache + ((int64_t)b * 3 + g) * Tmax * 64;
    float* vpage = v_cache + ((int64_t)b * 3 + g) * Tmax * 64;

    // ---- sum the split-K slabs, RoPE (rotate-half) the 3 query heads and the new key, append to the pages ----
    if (tid < 4 * 32) {
        const int hsel = tid >> 5, i = tid & 31;   // hsel 0..2 = query head 3g+hsel, 3 = new key
        const float c = rope_cos[(int64_t)pos * 32 + i], sn = rope_sin[(int64_t)pos * 32 + i];
        const int col = hsel < 3 ? (3 * g + hsel) * 64 : 576 + g * 64;
        float a1[KC], a2[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            a1[s] = row[s * slab_stride + col + i];
            a2[s] = row[s * slab_stride + col + i + 32];
        }
        float x1 = a1[0], x2 = a2[0];
#pragma unroll
        for (int s = 1; s < KC; ++s) { x1 += a1[s]; x2 += a2[s]; }
        const float o1 = __fadd_rn(__fmul_rn(x1, c), __fmul_rn(-x2, sn));
        const float o2 = __fadd_rn(__fmul_rn(x2, c), __fmul_rn(x1, sn));
        if (hsel < 3) {
            qs[hsel * 64 + i] = o1 * 0.125f;        // head_dim^-0.5 = 1/8 exactly
            qs[hsel * 64 + i + 32] = o2 * 0.125f;
        } else {
            knew[i] = o1; knew[i + 32] = o2;
            kpage[(int64_t)pos * 64 + i] = o1; kpage[(int64_t)pos * 64 + i + 32] = o2;
        }
    } else if (tid < 4 * 32 + 64) {
        const int i = tid - 128;
        float a1[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) a1[s] = row[s * slab_stride + 768 + g * 64 + i];
        float v = a1[0];
#pragma unroll
        for (int s = 1; s < KC; ++s) v += a1[s];
        vnew[i] = v;
        vpage[(int64_t)pos * 64 + i] = v;
    }
    __syncthreads();

    // lane -> (key sub = lane>>4, dim quad = lane&15); one wave instruction covers 4 keys x 64 dims
    const int sub = lane >> 4, quad = lane & 15;
    float4 q4[3];
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) q4[hh] = *reinterpret_cast<const float4*>(qs + hh * 64 + quad * 4);
    const int ngroups = (pos + 3) >> 2;   // groups of 4 cached keys
    float m_run[3] = {-INFINITY, -INFINITY, -INFINITY};
    float l_run[3] = {0.f, 0.f, 0.f};      // per-lane partial (this lane's keys only)
    float4 acc[3];
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int g0 = wave; g0 < ngroups; g0 += DA_WAVES * DA_G) {
        float4 k4[DA_G], v4[DA_G];
#pragma unroll
        for (int u = 0; u < DA_G; ++u) {
            const int t = (g0 + u * DA_WAVES) * 4 + sub;
            const int tc = t < pos ? t : pos - 1;   // pos >= 1 always (a prefix precedes)
            k4[u] = *reinterpret_cast<const float4*>(kpage + (int64_t)tc * 64 + quad * 4);
            v4[u] = *reinterpret_cast<const float4*>(vpage + (int64_t)tc * 64 + quad * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        float sc[DA_G][3];
        float cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int u = 0; u < DA_G; ++u) {
            const int t = (g0 + u * DA_WAVES) * 4 + sub;
#pragma unroll
            for (int hh = 0; hh < 3; ++hh) {
                float sv = q4[hh].x * k4[u].x + q4[hh].y * k4[u].y + q4[hh].z * k4[u].z + q4[hh].w * k4[u].w;
                sv += __shfl_xor(sv, 8, 64);
                sv += __shfl_xor(sv, 4, 64);
                sv += __shfl_xor(sv, 2, 64);
                sv += __shfl_xor(sv, 1, 64);
                sv = t < pos ? sv : -INFINITY;
                sc[u][hh] = sv;
                cmax[hh] = fmaxf(cmax[hh], sv);
            }
        }
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            float cm = cmax[hh];
            cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float m_new = fmaxf(m_run[hh], cm);       // finite: the chunk's first key (t = 4*g0) is valid
            const float alpha = expf(m_run[hh] - m_new);    // exp(-inf) = 0 on the first chunk
            m_run[hh] = m_new;
            float lsum = 0.f;
            float4 a = make_float4(acc[hh].x * alpha, acc[hh].y * alpha, acc[hh].z * alpha, acc[hh].w * alpha);
#pragma unroll
            for (int u = 0; u < DA_G; ++u) {
                const float p = expf(sc[u][hh] - m_new);    // masked keys: exp(-inf) = 0
                lsum += p;
                a.x += p * v4[u].x; a.y += p * v4[u].y; a.z += p * v4[u].z; a.w += p * v4[u].w;
            }
            acc[hh] = a;
            l_run[hh] = l_run[hh] * alpha + lsum;
        }
    }
    // reduce the 4 key-subs of the wave; publish (m, l, o) of the wave
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) {
        float l = l_run[hh];                 // identical across the 16 quads of a sub; sum over the 4 subs
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            acc[hh].x += __shfl_xor(acc[hh].x, off, 64);
            acc[hh].y += __shfl_xor(acc[hh].y, off, 64);
            acc[hh].z += __shfl_xor(acc[hh].z, off, 64);
            acc[hh].w += __shfl_xor(acc[hh].w, off, 64);
        }
        if (sub == 0) *reinterpret_cast<float4*>(ored + (wave * 3 + hh) * 64 + quad * 4) = acc[hh];
        if (lane == 0) { mred[wave * 3 + hh] = m_run[hh]; lred[wave * 3 + hh] = l; }
    }
    __syncthreads();
    if (tid < 192) {
        const int hh = tid >> 6, d = tid & 63;
        float snew = 0.f;                    // score of the new key (q . k_new, both in LDS)
#pragma unroll
        for (int i = 0; i < 64; ++i) snew += qs[hh * 64 + i] * knew[i];
        float M = snew;
#pragma unroll
        for (int w = 0; w < DA_WAVES; ++w) M = fmaxf(M, mred[w * 3 + hh]);
        const float pn = expf(snew - M);
        float L = pn, O = pn * vnew[d];
#pragma unroll
        for (int w = 0; w < DA_WAVES; ++w) {
            const float f = expf(mred[w * 3 + hh] - M);     // waves without keys: m = -inf -> factor 0
            L += lred[w * 3 + hh] * f;
            O += ored[(w * 3 + hh) * 64 + d] * f;
        }
        o[(int64_t)b * 576 + (3 * g + hh) * 64 + d] = O / L;
    }
}

void launch_decode_attention(const float* qkv_parts, int kc, int64_t slab_stride, float* k_cache, float* v_cache,
                             const float* rope_cos, const float* rope_sin, const int32_t* d_pos, float* o, int B,
                             int Tmax, hipStream_t s) {
    (void)kc;  // == SK_KC_QKV, validated by the engine
    hipLaunchKernelGGL((decode_attention_kernel<SK_KC_QKV>), dim3(3, B), dim3(DA_WAVES * 64), 0, s, qkv_parts, slab_stride,
                       k_cache, v_cache, rope_cos, rope_sin, d_pos, o, Tmax);
}

// ---- arg-max with first-index ties (torch.argmax, reference wrapper.py:232) ---------------------------------
// full-row version (taps / mellow_argmax): one workgroup per row, float4 loads
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
            if (vv[j] > best) { best = vv[j]; idx = 4 * i + j; }   // increasing index order: first max wins
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        tokens[b] = idx;
    }
}
void launch_argmax(const float* logits, int B, int V, int64_t ld, int32_t* tokens, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, s, logits, V, ld, tokens);
}
// candidate version: per-row reduction of the lm_head's per-tile (value, index) candidates, fused with the loop
// bookkeeping of reference wrapper.py:232-249: record the token at column (*d_pos - T0 + 1), track stop ids,
// and gather its embedding row as the next step's input (embed_tokens, wrapper.py:237).
__global__ __launch_bounds__(256) void argmax_cand_kernel(const float* __restrict__ cv, const int32_t* __restrict__ ci,
                                                          int n, int32_t* __restrict__ tokens,
                                                          const float* __restrict__ embed, int H, float* __restrict__ x,
                                                          int32_t* __restrict__ out_tokens, int max_len,
                                                          const int32_t* __restrict__ d_pos, int T0, int stop_id,
                                                          int32_t* seen_stop, int32_t* n_seen) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int tok_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
        const float v = cv[(int64_t)b * n + i];
        const int id = ci[(int64_t)b * n + i];
        if (v > best || (v == best && id < idx)) { best = v; idx = id; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        tokens[b] = idx;
        tok_s = idx;
        if (out_tokens) {
            const int step = *d_pos - T0 + 1;
            if (step >= 0 && step < max_len) out_tokens[(int64_t)b * max_len + step] = idx;
            if (idx == stop_id && seen_stop[b] == 0) {
                seen_stop[b] = 1;
                atomicAdd(n_seen, 1);
            }
        }
    }
    if (x) {
        __syncthreads();
        const float4* src = reinterpret_cast<const float4*>(embed + (int64_t)tok_s * H);
        float4* dst = reinterpret_cast<float4*>(x + (int64_t)b * H);
        for (int i = tid; i < H / 4; i += 256) dst[i] = src[i];
    }
}
void launch_argmax_cand(const float* cv, const int32_t* ci, int B, int n, int32_t* tokens, const float* embed, int H,
                        float* x, int32_t* out_tokens, int max_len, const int32_t* d_pos, int T0, int stop_id,
                        int32_t* seen_stop, int32_t* n_seen, hipStream_t s) {
    hipLaunchKernelGGL(argmax_cand_kernel, dim3(B), dim3(256), 0, s, cv, ci, n, tokens, embed, H, x, out_tokens, max_len,
                       d_pos, T0, stop_id, seen_stop, n_seen);
}

// ---- embedding gather of the new tokens + on-device loop bookkeeping (wrapper.py:236-249) ----------------------
__global__ void embed_and_record_kernel(const float* __restrict__ embed, const int32_t* __restrict__ tokens, int H,
                                        float* __restrict__ x, int32_t* __restrict__ out_tokens, int max_len,
                                        const int32_t* __restrict__ d_step, int stop_id, int32_t* seen_stop,
                                        int32_t* n_seen) {
    // grid = B blocks; block b copies embedding row tokens[b] into x[b] and records the token at column *d_step
    const int b = blockIdx.x;
    const int tok = tokens[b];
    if (x) {
        const float4* src = reinterpret_cast<const float4*>(embed + (int64_t)tok * H);
        float4* dst = reinterpret_cast<float4*>(x + (int64_t)b * H);
        for (int i = threadIdx.x; i < H / 4; i += blockDim.x) dst[i] = src[i];
    }
    if (threadIdx.x == 0 && out_tokens) {
        const int step = *d_step;
        if (step < max_len) out_tokens[(int64_t)b * max_len + step] = tok;
        if (tok == stop_id && seen_stop[b] == 0) {
            seen_stop[b] = 1;
            atomicAdd(n_seen, 1);
        }
    }
}
__global__ void advance_kernel(int32_t* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = *p + 1;
}
void launch_advance(int32_t* p, hipStream_t s) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, s, p); }
void launch_embed_and_record(const float* embed, const int32_t* tokens, int B, int H, float* x, int32_t* out_tokens,
                             int max_len, int32_t* d_step, int stop_id, int32_t* seen_stop, int32_t* n_seen,
                             hipStream_t s) {
    hipLaunchKernelGGL(embed_and_record_kernel, dim3(B), dim3(64), 0, s, embed, tokens, H, x, out_tokens, max_len,
                       (const int32_t*)d_step, stop_id, seen_stop, n_seen);
    if (out_tokens) launch_advance(d_step, s);   // separate launch: every block has read *d_step before it moves
}

__global__ void gather_rows_kernel(const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ rows, int C,
                                   float* __restrict__ out, int64_t ld_out) {
    const int r = blockIdx.x;
    const int src = rows ? rows[r] : r;
    for (int i = threadIdx.x; i < C; i += blockDim.x) out[(int64_t)r * ld_out + i] = in[(int64_t)src * ld_in + i];
}
void launch_gather_rows(const float* in, int64_t ld_in, const int32_t* rows, int n, int C, float* out, int64_t ld_out,
                        hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, in, ld_in, rows, C, out, ld_out);
}

__global__ void take_last_kernel(const float* __restrict__ x, int T, int C, float* __restrict__ out) {
    const int b = blockIdx.x;
    const float* src = x + ((int64_t)b * T + (T - 1)) * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) out[(int64_t)b * C + i] = src[i];
}
void launch_take_last(const float* x, int B, int T, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(take_last_kernel, dim3(B), dim3(256), 0, s, x, T, C, out);
}

}  // namespace mellow
