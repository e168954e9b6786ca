// Developer microbenchmark (GPU box), round 3: one decode GEMM stage of a PERSISTENT kernel, emulated with the real data
// movement of the gate/up projection (the launch it would replace costs 5.3 us traced):
//   256 workgroups x 256 threads; per stage each workgroup
//     1. requests its 36 KB of weights (non-temporal loads, a fresh region per stage: 9.4 MB x 64 regions = 600 MB, like the
//        540 MB of weights a step streams) BEFORE it waits,
//     2. arrives at the device-wide barrier (per-XCD counters + a generation word, grid_sync2.hip) and waits,
//     3. reads the whole 72 KB activation matrix the previous stage's workgroups wrote (288 B each), in one of three ways:
//          BYP   loads with sc0 sc1 (bypass the non-coherent L2s)
//          SC1   loads with sc1
//          INV   one `buffer_inv sc1` per workgroup, then plain loads (served by this XCD's L2 after the first miss)
//     4. 72 x v_mfma_f32_16x16x4_f32 per wave on the loaded registers, a 4-way LDS reduction,
//     5. writes its 288 B of output with write-through stores (sc0 sc1), tagged with the stage number: a stale read is counted.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/stage_emul.bin tools/microbench/stage_emul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_wt(float4* p, float4 v) {
    const f4v r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ unsigned ld_word(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Ctl {
    unsigned global;
    unsigned pad0[31];
    unsigned gen;
    unsigned pad1[31];
    unsigned xcd[8 * 32];
};

constexpr int NWG = 256, WPT = 9, XPT = 18;   // float4 per thread: weights 9 (36 KB / workgroup), activations 18 (72 KB)

template <int MODE, bool PREFETCH>   // 0 BYP, 1 SC1, 2 INV
__global__ __launch_bounds__(256) void stage_kernel(Ctl* c, const float4* __restrict__ weights, int64_t region_f4, int regions,
                                                    float4* act, int stages, float* out, unsigned* err) {
    __shared__ float red[4 * 8 * 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned bad = 0;
    float keep = 0.f;
    for (int s = 1; s <= stages; ++s) {
        const float4* wreg = weights + (int64_t)(s % regions) * region_f4 + (int64_t)b * (256 * WPT);
        f4v w[WPT];
        if (PREFETCH) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) w[i] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(wreg) + i * 256 + tid);
        }
        // ---- arrive + wait ----
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&c->xcd[(b & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(NWG / 8) * (unsigned)s - 1u) {
                const unsigned g = __hip_atomic_fetch_add(&c->global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g == 8u * (unsigned)s - 1u) __hip_atomic_store(&c->gen, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            while (ld_word(&c->gen) < (unsigned)s)
                if (++spins > (1 << 22)) { *err = 1; break; }
            if (MODE == 2) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!PREFETCH) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) w[i] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(wreg) + i * 256 + tid);
        }
        // ---- activations of the previous stage: buffer (s-1)&1, 256 workgroups x 18 float4 ----
        const float4* ain = act + (int64_t)((s - 1) & 1) * (NWG * XPT);
        f4v x[XPT];
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x[i]) : "v"(ain + i * 256 + tid) : "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[i]) : "v"(ain + i * 256 + tid) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < XPT; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[i]) : "v"(ain + i * 256 + tid) : "memory");
        }
        // the loads above are invisible to the compiler's wait counting: tie every destination to the wait
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                       "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17])
                     :: "memory");
        f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][j], x[2 * i][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][j], x[2 * i + 1][j], acc1, 0, 0, 0);
            }
        }
        if (s > 1) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) bad += (x[i][0] != (float)(s - 1));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(wave * 8 + r) * 64 + lane] = acc0[r];
            red[(wave * 8 + 4 + r) * 64 + lane] = acc1[r];
        }
        __syncthreads();
        if (tid < XPT) {
            float v = 0.f;
            for (int wv = 0; wv < 4; ++wv) v += red[(wv * 8 + (tid & 7)) * 64 + tid];
            keep += v;
            float4* aout = act + (int64_t)(s & 1) * (NWG * XPT) + b * XPT + tid;
            st_wt(aout, make_float4((float)s, v, 2.f, 3.f));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (bad) atomicAdd(err + 1, bad);
    if (keep == 123.456f) out[0] = keep;
}

template <int MODE, bool PF>
static void run(const char* name, Ctl* ctl, const float4* weights, int64_t region_f4, int regions, float4* act, float* out, unsigned* err) {
    const int stages = 4000;
    hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(err, 0, 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((stage_kernel<MODE, PF>), dim3(NWG), dim3(256), 0, 0, ctl, weights, region_f4, regions, act, stages, out, err);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned e[2] = {0, 0};
    hipMemcpy(e, err, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.3f us per stage (timeouts %u, stale reads %u)\n", name, ms * 1e3 / stages, e[0], e[1]);
}

int main() {
    Ctl* ctl;
    unsigned* err;
    float* out;
    float4 *weights, *act;
    const int regions = 64;
    const int64_t region_f4 = (int64_t)NWG * 256 * WPT;
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&err, 8); hipMalloc(&out, 4);
    hipMalloc(&weights, region_f4 * regions * 16);
    hipMemset(weights, 0, region_f4 * regions * 16);
    hipMalloc(&act, 2 * NWG * XPT * 16);
    hipMemset(act, 0, 2 * NWG * XPT * 16);
    run<0, true>("BYP, weights prefetched", ctl, weights, region_f4, regions, act, out, err);
    run<1, true>("SC1, weights prefetched", ctl, weights, region_f4, regions, act, out, err);
    run<2, true>("INV, weights prefetched", ctl, weights, region_f4, regions, act, out, err);
    run<0, false>("BYP, weights after the wait", ctl, weights, region_f4, regions, act, out, err);
    run<1, false>("SC1, weights after the wait", ctl, weights, region_f4, regions, act, out, err);
    run<2, false>("INV, weights after the wait", ctl, weights, region_f4, regions, act, out, err);
    return 0;
}
