// Developer microbenchmark (GPU box), round 5: would a decode layer with TWO seams (one "attention block" launch + one "MLP block"
// launch, partitioned the tensor-parallel way with workgroups as the ranks) beat today's four launches (23.5 us per layer)?
// Gate set by the round-4 review: build it only if the emulated layer costs <= 17 us.
//
// Emulated per layer (B = 32 rows, hidden 576, 9 q / 3 kv heads of 64, intermediate 1536, context 420 keys, fp32):
//
//   ATTENTION BLOCK, one launch, workgroup = (kv group g of 3, R rows, key split sp of SP):
//     prologue   x_new = x_mid + sum of the S slabs the previous MLP block wrote (R x (1 + S) x 2.3 KB), sum of squares
//     q/k/v      its group's 320 x 576 slice of the norm-folded projection (737 KB per workgroup, the same slice for every row
//                group: L2-resident after the first workgroup of an XCD has pulled it), R GEMVs on the vector ALUs
//     attention  its (row, group) K and V pages (R x 215 KB / SP, streamed once, non-temporal), dot products + exp
//     o_proj     the 192 columns of W_o that belong to its heads (442 KB), R GEMVs -> one 576-float partial per (row, g, sp)
//   MLP BLOCK, one launch, workgroup = slice s of S of the 1536 hidden units, all 32 rows, 4 waves:
//     prologue   x_mid_new = x_new + sum of the 3 SP attention slabs (72 KB x (1 + 3 SP)), in B-fragment order, sum of squares
//     gate/up    its 2 x (1536 / S) x 576 slice (streamed once, nt), fp32 MFMA 32x32x2, K split over the 4 waves, LDS reduce
//     SwiGLU     local;  down: its (1536 / S) rows of W_d (576 x 1536 / S), fp32 MFMA -> one 32 x 576 slab per workgroup
//
// Every launch reads what the previous one wrote; weights and pages come from per-layer regions (538 MB + 620 MB cycle through
// the 256 MB Infinity Cache like a real step).  Values are meaningless (zero weights); byte counts, instruction counts and
// dependencies are the real ones.  Chains of 30 layers are captured into a hipGraph and replayed.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/two_seam_emul.bin tools/microbench/two_seam_emul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ f4v ld_nt(const f4v* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float hsum(f4v v) { return (v[0] + v[1]) + (v[2] + v[3]); }

constexpr int T_KEYS = 420;
constexpr int SMAX = 48;

struct P {
    const f4v* wqkvT;   // [layer][3 groups][576 k][80 float4]    737 KB per (layer, group)
    const f4v* woT;     // [layer][3 groups][192 k][144 float4]   442 KB
    const f4v* kv;      // [layer][32 rows][3 groups][2][T_KEYS][16 float4]
    const f4v* wgu;     // [layer][S slices][n-tile][72 k-tiles][64]   (P-layout, 7 MB per layer)
    const f4v* wd;      // [layer][S slices][18 n-tiles][HS / 8 k-tiles][64]   (3.5 MB per layer)
    f4v* xmid;          // [32][144]
    f4v* mslab;         // [SMAX][32][144]   MLP block output slabs
    f4v* aslab;         // [6][32][144]      attention block output slabs (3 groups x SP)
    float* sink;
};

// ---------------------------------------------------------------------------------------------------------------------------
template <int R, int SP, int NT, int S>
__global__ __launch_bounds__(NT) void k_attn_block(P p, int layer) {
    constexpr int QP = NT / 80, QK = 576 / QP;            // q/k/v: column float4 c = t % 80, k part = t / 80
    constexpr int OP = NT / 144 > 6 ? 6 : NT / 144, OK = 192 / OP;      // o_proj: column float4 c = t % 144, k part
    static_assert(QK * QP == 576 && OK * OP == 192, "parts must divide K");
    __shared__ __attribute__((aligned(16))) float xs[R][576];
    __shared__ __attribute__((aligned(16))) float qkv[R][320];
    __shared__ __attribute__((aligned(16))) float att[R][192];
    __shared__ __attribute__((aligned(16))) f4v red[(QP > OP ? QP * 80 : OP * 144) * R];
    __shared__ float ssp[NT / 64], kvacc[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, rg = blockIdx.y, sp = blockIdx.z;
    // ---- K/V pages of the first row requested up front (as the real kernel does), the rest after the projections
    constexpr int KV4 = 2 * T_KEYS * 16 / SP;              // float4 per (row, group, split)
    constexpr int KVL = (KV4 + NT - 1) / NT;
    constexpr bool EARLY = NT <= 512;                      // (16-wave workgroups have 128 VGPRs per lane: pages requested after the projections)
    f4v kvr[KVL];
    const f4v* page0 = p.kv + ((((int64_t)layer * 32 + rg * R) * 3 + g) * SP + sp) * KV4;
    if (EARLY) {
#pragma unroll
        for (int i = 0; i < KVL; ++i) kvr[i] = ld_nt(page0 + (tid + i * NT < KV4 ? tid + i * NT : KV4 - 1));
    }
    // ---- prologue
    float ss = 0.f;
    if (tid < 144 * R) {
        const int r = tid / 144, c = tid % 144, row = rg * R + r;
        f4v x = p.xmid[row * 144 + c];
#pragma unroll
        for (int s0 = 0; s0 < S; s0 += 12) {
            f4v sl[12];
#pragma unroll
            for (int s = 0; s < 12; ++s) sl[s] = p.mslab[((int64_t)(s0 + s) * 32 + row) * 144 + c];
#pragma unroll
            for (int s = 0; s < 12; ++s) x += sl[s];
        }
        *reinterpret_cast<f4v*>(&xs[r][4 * c]) = x;
        ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        if (g == 0 && sp == 0) p.xmid[row * 144 + c] = x;              // the residual row for the MLP block
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) ssp[wave] = ss;
    __syncthreads();
    float rs = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) rs += ssp[w];
    rs = 1.0f / sqrtf(rs / (576.0f * R) + 1e-5f);
    // ---- q/k/v of the group: W^T [576][80 float4]
    if (tid < QP * 80) {
        const int c = tid % 80, part = tid / 80;
        const f4v* w = p.wqkvT + (((int64_t)layer * 3 + g) * 576 + part * QK) * 80 + c;
        f4v acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f4v{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < QK; k0 += 12) {
            f4v wv[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) wv[j] = w[(k0 + j) * 80];
#pragma unroll
            for (int j = 0; j < 12; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] += wv[j] * xs[r][part * QK + k0 + j];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) red[(part * R + r) * 80 + c] = acc[r];
    }
    __syncthreads();
    if (tid < 80 * R) {
        const int r = tid / 80, c = tid % 80;
        f4v v = red[r * 80 + c];
#pragma unroll
        for (int q = 1; q < QP; ++q) v += red[(q * R + r) * 80 + c];
        *reinterpret_cast<f4v*>(&qkv[r][4 * c]) = v * rs;
    }
    __syncthreads();
    // ---- attention: dot products of the streamed pages with q (3 heads), one exp per key group
    float a = 0.f;
    for (int r = 0; r < R; ++r) {
        if (r > 0 || !EARLY) {
            const f4v* page = page0 + (int64_t)r * 3 * SP * KV4;
#pragma unroll
            for (int i = 0; i < KVL; ++i) kvr[i] = ld_nt(page + (tid + i * NT < KV4 ? tid + i * NT : KV4 - 1));
        }
        const f4v q0 = *reinterpret_cast<const f4v*>(&qkv[r][4 * (lane & 15)]), q1 = *reinterpret_cast<const f4v*>(&qkv[r][64 + 4 * (lane & 15)]),
                  q2 = *reinterpret_cast<const f4v*>(&qkv[r][128 + 4 * (lane & 15)]);
#pragma unroll
        for (int i = 0; i < KVL; ++i) {
            float s0 = hsum(kvr[i] * q0), s1 = hsum(kvr[i] * q1), s2 = hsum(kvr[i] * q2);
            s0 += __shfl_xor(s0, 1, 64); s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 4, 64);
            a += __expf(s0 - 1.f) * kvr[i][0] + __expf(s1 - 1.f) * kvr[i][1] + __expf(s2 - 1.f) * kvr[i][2];
        }
        float ar = a;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ar += __shfl_xor(ar, o, 64);
        if (lane == 0) kvacc[wave] = ar;
        __syncthreads();
        if (tid < 192) {
            float v = qkv[r][tid];
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) v += kvacc[w];
            att[r][tid] = v;
        }
        __syncthreads();
    }
    // ---- partial o_proj: Wo^T slice [192][144 float4]
    if (tid < OP * 144) {
        const int c = tid % 144, part = tid / 144;
        const f4v* w = p.woT + (((int64_t)layer * 3 + g) * 192 + part * OK) * 144 + c;
        f4v acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f4v{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < OK; k0 += 16) {
            f4v wv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[j] = w[(k0 + j) * 144];
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] += wv[j] * att[r][part * OK + k0 + j];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) red[(part * R + r) * 144 + c] = acc[r];
    }
    __syncthreads();
    if (tid < 144 * R) {
        const int r = tid / 144, c = tid % 144;
        f4v v = red[r * 144 + c];
#pragma unroll
        for (int q = 1; q < OP; ++q) v += red[(q * R + r) * 144 + c];
        p.aslab[((int64_t)(g * SP + sp) * 32 + rg * R + r) * 144 + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
template <int S, int NAS>        // NAS = attention slabs to sum (3 x SP)
__global__ __launch_bounds__(256) void k_mlp_block(P p, int layer) {
    constexpr int HS = 1536 / S, NTL = 2 * HS / 32, KD = HS / 8;       // hidden units, gate/up n-tiles, down k-tiles of the slice
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* red = lds;                                              // [4 waves][NTL][16][64]
    f4v* hF = reinterpret_cast<f4v*>(lds + 4 * NTL * 16 * 64);     // [KD][64]
    float* ssq_s = lds + 4 * NTL * 16 * 64 + KD * 64 * 4;          // [4][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    // ---- prologue: this wave's 18 k-tiles of x_mid_new in B-fragment order (x_new + the attention slabs)
    f4v x[18];
    float ssp = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < 18; i0 += 6) {
        f4v sl[6][NAS];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            x[i0 + i] = p.xmid[(wave * 18 + i0 + i) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NAS; ++j) sl[i][j] = p.aslab[(int64_t)j * 4608 + (wave * 18 + i0 + i) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < NAS; ++j) x[i0 + i] += sl[i][j];
            ssp += x[i0 + i][0] * x[i0 + i][0] + x[i0 + i][1] * x[i0 + i][1] + x[i0 + i][2] * x[i0 + i][2] + x[i0 + i][3] * x[i0 + i][3];
            if (s == 0) p.xmid[(wave * 18 + i0 + i) * 64 + lane] = x[i0 + i];       // the residual stream for the next attention block
        }
    }
    ssp += __shfl_xor(ssp, 32, 64);
    if (lane < 32) ssq_s[wave * 32 + lane] = ssp;
    // ---- gate/up slice
    const f4v* wg = p.wgu + ((int64_t)layer * S + s) * (NTL * 72 * 64) + (wave * 18) * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        f4v w[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) w[i] = ld_nt(wg + (nt * 72 + i) * 64);
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i][j], x[i][j], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NTL + nt) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    // ---- SwiGLU: 32 rows x HS hidden units, thread -> 4 consecutive hidden units of one row (F-layout float4)
    for (int e = tid; e < KD * 64; e += 256) {
        const int kt = e >> 6, ln = e & 63, m = ln & 31;
        const float r2 = 1.0f / sqrtf((ssq_s[m] + ssq_s[32 + m] + ssq_s[64 + m] + ssq_s[96 + m]) / 576.0f + 1e-5f);
        f4v h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hu = kt * 8 + (ln >> 5) * 4 + j;          // hidden unit of the slice: gate row hu, up row HS + hu
            const int gt = hu >> 5, gr = hu & 31, ut = (HS + hu) >> 5, ur = (HS + hu) & 31;
            float gv = 0.f, uv = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                gv += red[((w * NTL + gt) * 16 + (gr >> 3) * 4 + (gr & 3)) * 64 + m + 32 * ((gr >> 2) & 1)];
                uv += red[((w * NTL + ut) * 16 + (ur >> 3) * 4 + (ur & 3)) * 64 + m + 32 * ((ur >> 2) & 1)];
            }
            gv *= r2; uv *= r2;
            h[j] = gv / (1.0f + __expf(-gv)) * uv;
        }
        hF[e] = h;
    }
    __syncthreads();
    // ---- partial down: wave -> n-tiles wave, wave + 4, ...; K = HS
    const f4v* wd = p.wd + ((int64_t)layer * S + s) * (18 * KD * 64) + lane;
    f4v hx[KD];
#pragma unroll
    for (int k = 0; k < KD; ++k) hx[k] = hF[k * 64 + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int nt = wave + 4 * q;
        if (nt < 18) {
            f4v w[KD];
#pragma unroll
            for (int k = 0; k < KD; ++k) w[k] = ld_nt(wd + (nt * KD + k) * 64);
            f16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int k = 0; k < KD; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k][j], hx[k][j], acc, 0, 0, 0);
            // D[n = 8 (r / 4) + 4 (lane / 32) + r % 4][m = lane % 32] -> row-major slab, float4 of 4 consecutive n
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                p.mslab[((int64_t)s * 32 + (lane & 31)) * 144 + nt * 8 + r4 * 2 + (lane >> 5)] = f4v{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
        }
    }
}

template <int R, int SP, int NT, int S>
static int run(P p, hipStream_t st, const char* name, int what /* 3 both, 1 attention only, 2 MLP only */) {
    const int LAYERS = 30, REPS = 100;
    constexpr int mlp_lds = (4 * (2 * (1536 / S) / 32) * 16 * 64 + (1536 / S / 8) * 64 * 4 + 128) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_block<S, 3 * SP>), hipFuncAttributeMaxDynamicSharedMemorySize, mlp_lds));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < LAYERS; ++l) {
        if (what & 1) hipLaunchKernelGGL((k_attn_block<R, SP, NT, S>), dim3(3, 32 / R, SP), dim3(NT), 0, st, p, l);
        if (what & 2) hipLaunchKernelGGL((k_mlp_block<S, 3 * SP>), dim3(S), dim3(256), mlp_lds, st, p, l);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-78s %8.2f us per layer\n", name, ms * 1e3 / (REPS * LAYERS));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int LAYERS = 30;
    P p;
    void *a, *b, *c, *d, *e;
    const size_t n_qkv = (size_t)LAYERS * 3 * 576 * 80 * 16, n_o = (size_t)LAYERS * 3 * 192 * 144 * 16,
                 n_kv = (size_t)LAYERS * 32 * 3 * 2 * T_KEYS * 16 * 16, n_gu = (size_t)LAYERS * 3072 * 576 * 4, n_d = (size_t)LAYERS * 1536 * 576 * 4;
    CK(hipMalloc(&a, n_qkv)); CK(hipMemset(a, 0, n_qkv));
    CK(hipMalloc(&b, n_o)); CK(hipMemset(b, 0, n_o));
    CK(hipMalloc(&c, n_kv)); CK(hipMemset(c, 0, n_kv));
    CK(hipMalloc(&d, n_gu)); CK(hipMemset(d, 0, n_gu));
    CK(hipMalloc(&e, n_d)); CK(hipMemset(e, 0, n_d));
    p.wqkvT = (const f4v*)a; p.woT = (const f4v*)b; p.kv = (const f4v*)c; p.wgu = (const f4v*)d; p.wd = (const f4v*)e;
    CK(hipMalloc(&p.xmid, 4608 * 16)); CK(hipMemset(p.xmid, 0, 4608 * 16));
    CK(hipMalloc(&p.mslab, (size_t)SMAX * 4608 * 16)); CK(hipMemset(p.mslab, 0, (size_t)SMAX * 4608 * 16));
    CK(hipMalloc(&p.aslab, 6 * 4608 * 16)); CK(hipMemset(p.aslab, 0, 6 * 4608 * 16));
    CK(hipMalloc(&p.sink, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("two-seam decode layer, emulated (B = 32, 420 keys, fp32); today's four-launch layer: 23.5 us; gate: <= 17 us\n");
    // attention block geometries x hidden slices
    if (run<1, 1, 1024, 24>(p, st, "attn (row, g) 96 wg x 16 waves  + mlp 24 slices of 64 hidden", 3)) return 1;
    if (run<1, 2, 512, 24>(p, st, "attn (row, g, 2 key splits) 192 wg x 8 waves + mlp 24 slices", 3)) return 1;
    if (run<2, 1, 1024, 24>(p, st, "attn (2 rows, g) 48 wg x 16 waves + mlp 24 slices", 3)) return 1;
    if (run<2, 2, 512, 24>(p, st, "attn (2 rows, g, 2 key splits) 96 wg x 8 waves + mlp 24 slices", 3)) return 1;
    if (run<4, 2, 512, 24>(p, st, "attn (4 rows, g, 2 key splits) 48 wg x 8 waves + mlp 24 slices", 3)) return 1;
    if (run<1, 1, 1024, 48>(p, st, "attn (row, g) 96 wg x 16 waves  + mlp 48 slices of 32 hidden", 3)) return 1;
    if (run<1, 2, 512, 48>(p, st, "attn (row, g, 2 key splits) 192 wg x 8 waves + mlp 48 slices", 3)) return 1;
    if (run<2, 2, 512, 48>(p, st, "attn (2 rows, g, 2 key splits) 96 wg x 8 waves + mlp 48 slices", 3)) return 1;
    // each launch of the pair on its own (chains of 30 identical launches): where the time is
    if (run<1, 1, 1024, 24>(p, st, "  attention block only: (row, g) 96 wg x 16 waves, sums 24 slabs", 1)) return 1;
    if (run<1, 2, 512, 24>(p, st, "  attention block only: (row, g, split) 192 wg x 8 waves, sums 24 slabs", 1)) return 1;
    if (run<2, 2, 512, 24>(p, st, "  attention block only: (2 rows, g, split) 96 wg x 8 waves, sums 24 slabs", 1)) return 1;
    if (run<1, 2, 512, 48>(p, st, "  attention block only: (row, g, split) 192 wg x 8 waves, sums 48 slabs", 1)) return 1;
    if (run<1, 1, 1024, 24>(p, st, "  MLP block only: 24 slices, sums 3 attention slabs", 2)) return 1;
    if (run<1, 2, 512, 24>(p, st, "  MLP block only: 24 slices, sums 6 attention slabs", 2)) return 1;
    if (run<1, 1, 1024, 48>(p, st, "  MLP block only: 48 slices, sums 3 attention slabs", 2)) return 1;
    if (run<1, 2, 512, 48>(p, st, "  MLP block only: 48 slices, sums 6 attention slabs", 2)) return 1;
    return 0;
}
