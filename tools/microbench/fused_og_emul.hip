// Developer microbenchmark (GPU box), round 4: would the 3-launch decode layer pay?  The data movement, matrix work and
// dependencies of the o_proj -> gate/up pair of a decode layer (B = 32, hidden 576, intermediate 1536), emulated in two forms
// inside a chain of dependent launches (a captured hipGraph of 30 "layers", like a decode step):
//
//   TWO   o_proj-shaped launch  (144 workgroups x 12 waves: 36 KB of weights, 36 KB of attention partials + (m, l), residual;
//                                12 x v_mfma_f32_16x16x4_f32 per wave; 12-way LDS reduction; writes 8 rows x 16 columns + a statistic)
//         gate/up-shaped launch (192 workgroups x 4 waves: 36 KB of weights, 72 KB of x_mid, the 36 statistic partials of its rows;
//                                72 MFMAs per wave; 4-way reduction; SwiGLU; writes 32 rows x 8 hidden units)
//   ONE   a single launch: the 144 o_proj-shaped workgroups first (lowest block ids, dispatched first) + 192 gate/up workgroups
//         on the COMPOSED operand [Wg' | Wg' Wo]: 72 KB of weights, x (72 KB) and BOTH attention partials (144 KB), 144 MFMAs per
//         wave; the SwiGLU needs the RMS statistic of x_mid, which the o_proj-shaped workgroups of the SAME launch publish as
//         tagged 8-byte granules (write-through) and every gate/up workgroup polls (L2-bypassing loads) before its epilogue.
//   Both are followed by a qkv2-shaped launch (252 workgroups x 4 waves, 48 KB of weights + its activations) that reads what the
//   pair wrote, so every launch of the chain has a real producer and a real consumer.
// Weights stream from fresh regions (non-temporal loads), like the 538 MB a decode step touches once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/fused_og_emul.bin tools/microbench/fused_og_emul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ f4v ld_nt(const f4v* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ f4v mfma4x(f4v acc, f4v w, f4v x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], x[3], acc, 0, 0, 0);
    return acc;
}

struct Bufs {
    const f4v* w;          // weight pool
    int64_t region_f4;     // f4 per layer region
    f4v* att;              // attention partials: 2 splits x 32 rows x 576  (float4: 2 * 4608)
    float* ml;             // (m, l) pairs
    f4v* xres;             // residual, 32 x 576
    f4v* xmid;             // o_proj output (32 x 576), 3 copies like the real kernel are not emulated: one
    float* ssq;            // [32][40]
    unsigned long long* stat;   // tagged granules {value, tag}: [36 tiles][32 rows]
    f4v* h;                // gate/up output 32 x 1536
    f4v* out;              // qkv2-shaped output
    unsigned* err;         // [0] sweeps that gave up, [1] last tag seen by a sweep that gave up
};

// ---- the o_proj-shaped work: one workgroup = (n16 tile nt of 36, row quarter part of 4), 12 waves x 3 k16-tiles ----
template <int WAVES>
__device__ __forceinline__ void oproj_work(const Bufs& b, int layer, int nt, int part, unsigned tag, bool publish, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f4v* wp = b.w + (int64_t)layer * b.region_f4 + (int64_t)nt * (36 * 64) + lane;
    constexpr int TP = 36 / WAVES;
    f4v w[TP], o0[TP], o1[TP];
#pragma unroll
    for (int i = 0; i < TP; ++i) {
        const int t = wave + WAVES * i;
        w[i] = ld_nt(wp + t * 64);
        o0[i] = b.att[(int64_t)(t * 2 + (part >> 1)) * 64 + lane];
        o1[i] = b.att[4608 + (int64_t)(t * 2 + (part >> 1)) * 64 + lane];
    }
    const float m = b.ml[(lane & 15) * 4 + (part & 1)];
    const f4v xr = b.xres[(nt * 4 + part) * 16 + (tid & 15)];
    __builtin_amdgcn_sched_barrier(0);
    f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TP; ++i) {
        f4v x = o0[i] * m + o1[i];
        acc = mfma4x(acc, w[i], x);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (tid < 64) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int wv = 0; wv < WAVES; ++wv)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += red[(wv * 4 + r) * 64 + tid];
        const f4v y = {xr[0] + v[0], xr[1] + v[1], xr[2] + v[2], xr[3] + v[3]};
        b.xmid[(nt * 4 + part) * 32 + (tid & 31)] = y;
        float ss = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        if ((tid & 3) == 0) {
            const int row = part * 8 + (tid >> 2) % 8;
            if (publish) {
                // tagged granule, write-through: {value, tag} in one 8-byte store
                const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(ss);
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(b.stat + nt * 32 + row), "v"(g) : "memory");
            } else {
                b.ssq[row * 40 + nt] = ss;
            }
        }
    }
}

// ---- the gate/up-shaped work: workgroup = n16 tile nt of 192; 4 waves x (KT16 / 4) k16-tiles; X = x_mid (K = 576) or, COMPOSED,
//      [x | att split 0 | att split 1] with K = 1152 of weights (the two partials are merged on load like the o_proj does) ----
template <bool COMPOSED, bool POLL = true>
__device__ __forceinline__ void gateup_work(const Bufs& b, int layer, int nt, unsigned tag, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TPW = COMPOSED ? 18 : 9;
    const f4v* wp = b.w + (int64_t)layer * b.region_f4 + 1048576 + (int64_t)nt * ((COMPOSED ? 72 : 36) * 64) + lane;
    f4v w[TPW], x0[TPW], x1[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave * TPW + i;
        w[i] = ld_nt(wp + t * 64);
        if (!COMPOSED) {
            x0[i] = b.xmid[(t * 2) * 64 + lane];
            x1[i] = b.xmid[(t * 2 + 1) * 64 + lane];
        } else if (t < 36) {
            x0[i] = b.xres[(t * 2) * 64 + lane];
            x1[i] = b.xres[(t * 2 + 1) * 64 + lane];
        } else {
            const int u = t - 36;
            x0[i] = b.att[(u * 2) * 64 + lane] + b.att[4608 + (u * 2) * 64 + lane];
            x1[i] = b.att[(u * 2 + 1) * 64 + lane] + b.att[4608 + (u * 2 + 1) * 64 + lane];
        }
    }
    f4v s4[9];
    if (!COMPOSED && tid < 64) {
#pragma unroll
        for (int j = 0; j < 9; ++j) s4[j] = reinterpret_cast<const f4v*>(b.ssq + ((tid >> 1) & 31) * 40)[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        acc0 = mfma4x(acc0, w[i], x0[i]);
        acc1 = mfma4x(acc1, w[i], x1[i]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(wave * 8 + r) * 64 + lane] = acc0[r];
        red[(wave * 8 + 4 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
    if (tid < 64) {
        const int m = tid >> 1;
        float ss = 0.f;
        if (COMPOSED && !POLL) {
            ss = 1.0f;          // timing only: what the fused launch costs WITHOUT the in-launch hand-off of the statistic
        } else if (COMPOSED) {
            // the statistic of x_mid comes from the o_proj-shaped workgroups of THIS launch: thread (row m, half q) sweeps 18 of the
            // row's 36 tagged granules with independent L2-bypassing loads and repeats the sweep until every tag is this launch's
            const int q = tid & 1;
            unsigned long long g[18];
            int spins = 0;
            bool all;
            do {
#pragma unroll
                for (int j = 0; j < 18; ++j)
                    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(g[j]) : "v"(b.stat + (q * 18 + j) * 32 + m) : "memory");
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]), "+v"(g[8]),
                               "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]), "+v"(g[13]), "+v"(g[14]), "+v"(g[15]), "+v"(g[16]), "+v"(g[17])
                             :: "memory");
                all = true;
#pragma unroll
                for (int j = 0; j < 18; ++j) all = all && (unsigned)(g[j] >> 32) == tag;
            } while (!all && ++spins < 4096);
            if (!all) { atomicAdd(b.err, 1u); b.err[1] = (unsigned)(g[0] >> 32); b.err[2] = tag; }
#pragma unroll
            for (int j = 0; j < 18; ++j) ss += __uint_as_float((unsigned)g[j]);
            ss += __shfl_xor(ss, 1, 64);
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j) ss += (s4[j][0] + s4[j][1]) + (s4[j][2] + s4[j][3]);
        }
        const float r2 = 1.0f / sqrtf(ss / 576.0f + 1e-5f);
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gv = 0.f, uv = 0.f;
            for (int wv = 0; wv < 4; ++wv) { gv += red[(wv * 8 + r) * 64 + tid]; uv += red[(wv * 8 + 4 + r) * 64 + tid]; }
            gv *= r2; uv *= r2;
            hv[r] = gv / (1.0f + __expf(-gv)) * uv;
        }
        b.h[nt * 64 + tid] = f4v{hv[0], hv[1], hv[2], hv[3]};
    }
}

__global__ __launch_bounds__(768) void k_oproj(Bufs b, int layer) {
    __shared__ float red[12 * 4 * 64];
    oproj_work<12>(b, layer, blockIdx.x % 36, blockIdx.x / 36, 0u, false, red);
}
__global__ __launch_bounds__(256) void k_gateup(Bufs b, int layer) {
    __shared__ float red[4 * 8 * 64];
    gateup_work<false>(b, layer, blockIdx.x, 0u, red);
}
// ONE launch (256-thread blocks: a launch has one block size): blocks [0, 144) = o_proj-shaped on 4 waves x 9 k16-tiles,
// blocks [144, 336) = gate/up on the composed operand
__global__ __launch_bounds__(256) void k_fused(Bufs b, int layer, unsigned tag) {
    __shared__ float red[4 * 8 * 64];
    if (blockIdx.x < 144) oproj_work<4>(b, layer, blockIdx.x % 36, blockIdx.x / 36, tag, true, red);
    else gateup_work<true>(b, layer, blockIdx.x - 144, tag, red);
}
__global__ __launch_bounds__(256) void k_fused_nopoll(Bufs b, int layer, unsigned tag) {
    __shared__ float red[4 * 8 * 64];
    if (blockIdx.x < 144) oproj_work<4>(b, layer, blockIdx.x % 36, blockIdx.x / 36, tag, true, red);
    else gateup_work<true, false>(b, layer, blockIdx.x - 144, tag, red);
}
// the launch after the pair (qkv2-shaped): 252 workgroups x 4 waves x 12 k8-tiles of weights + h / x_mid
__global__ __launch_bounds__(256) void k_next(Bufs b, int layer) {
    __shared__ float red[4 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f4v* wp = b.w + (int64_t)layer * b.region_f4 + 2097152 + (int64_t)blockIdx.x * (48 * 64) + lane;
    f4v w[12], x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        w[i] = ld_nt(wp + (wave * 12 + i) * 64);
        x[i] = blockIdx.x < 60 ? b.xmid[((wave * 12 + i) % 72) * 64 + lane] : b.h[(((blockIdx.x - 60) / 48) * 48 + wave * 12 + i) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i][j], x[i][j], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = 4 * (tid >> 6) + j;
        float s = 0.f;
        for (int wv = 0; wv < 4; ++wv) s += red[(wv * 16 + q) * 64 + (tid & 63)];
        v[j] = s;
    }
    b.out[blockIdx.x * 256 + tid] = f4v{v[0], v[1], v[2], v[3]};
    // the attention of the next layer would consume this; here the next layer's o_proj-shaped launch reads att / xres, which
    // this launch refreshes so that the chain stays dependent
    if (blockIdx.x < 36) { b.att[blockIdx.x * 256 + tid] = f4v{v[0], 1.f, 1.f, 1.f}; b.xres[(blockIdx.x * 256 + tid) % 4608] = f4v{v[1], 1.f, 1.f, 1.f}; }
}

int main() {
    const int LAYERS = 30, REPS = 200;
    setvbuf(stdout, nullptr, _IONBF, 0);
    Bufs b;
    const int64_t region_f4 = 4194304;     // 64 MB per layer region (o_proj 1.3 MB @0, gate/up [composed 14.2 MB] @16 MB, next 12 MB @32 MB)
    f4v* w;
    CK(hipMalloc(&w, region_f4 * LAYERS * 16));
    CK(hipMemset(w, 0, region_f4 * LAYERS * 16));
    b.w = w; b.region_f4 = region_f4;
    CK(hipMalloc(&b.att, 2 * 4608 * 16)); CK(hipMemset(b.att, 0, 2 * 4608 * 16));
    CK(hipMalloc(&b.ml, 4096)); CK(hipMemset(b.ml, 0, 4096));
    CK(hipMalloc(&b.xres, 4608 * 16)); CK(hipMemset(b.xres, 0, 4608 * 16));
    CK(hipMalloc(&b.xmid, 4608 * 16)); CK(hipMemset(b.xmid, 0, 4608 * 16));
    CK(hipMalloc(&b.ssq, 32 * 40 * 4)); CK(hipMemset(b.ssq, 0, 32 * 40 * 4));
    CK(hipMalloc(&b.stat, 36 * 32 * 8)); CK(hipMemset(b.stat, 0, 36 * 32 * 8));
    CK(hipMalloc(&b.h, 192 * 64 * 16)); CK(hipMemset(b.h, 0, 192 * 64 * 16));
    CK(hipMalloc(&b.out, 252 * 256 * 16));
    CK(hipMalloc(&b.err, 64)); CK(hipMemset(b.err, 0, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 4; ++mode) {       // 0 TWO, 1 ONE, 2 TWO again (box drift), 3 ONE without the statistic hand-off (timing only)
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < LAYERS; ++l) {
            if (mode == 1) {
                hipLaunchKernelGGL(k_fused, dim3(336), dim3(256), 0, s, b, l, (unsigned)(l + 1));
            } else if (mode == 3) {
                hipLaunchKernelGGL(k_fused_nopoll, dim3(336), dim3(256), 0, s, b, l, (unsigned)(l + 1));
            } else {
                hipLaunchKernelGGL(k_oproj, dim3(144), dim3(768), 0, s, b, l);
                hipLaunchKernelGGL(k_gateup, dim3(192), dim3(256), 0, s, b, l);
            }
            hipLaunchKernelGGL(k_next, dim3(252), dim3(256), 0, s, b, l);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        // the tags of ONE must differ between replays: the granules carry the layer number, and every layer's granules are
        // rewritten by its own launch before they are polled, so a stale tag of the previous replay (same value!) would be accepted:
        // clear them between replays (cheap memset node outside the timed chain would distort; instead time REPS replays, each
        // preceded by a memsetAsync of 9 KB -- the same for both modes)
        for (int i = 0; i < 5; ++i) { CK(hipMemsetAsync(b.stat, 0, 36 * 32 * 8, s)); CK(hipGraphLaunch(ge, s)); }
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < REPS; ++i) { CK(hipMemsetAsync(b.stat, 0, 36 * 32 * 8, s)); CK(hipGraphLaunch(ge, s)); }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned he[4] = {0, 0, 0, 0};
        CK(hipMemcpy(he, b.err, 16, hipMemcpyDeviceToHost));
        printf("%-40s %8.3f us per layer (%d layers x %d replays; sweeps that gave up: %u, tag seen %u wanted %u)\n",
               mode == 1 ? "ONE fused o_proj + gate/up launch" : mode == 3 ? "ONE, statistic hand-off left out" : "TWO launches (o_proj | gate/up)", ms * 1e3 / (REPS * LAYERS), LAYERS, REPS, he[0], he[1], he[2]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
