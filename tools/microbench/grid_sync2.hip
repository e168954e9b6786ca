// Developer microbenchmark (GPU box), round 3: the cheapest device-wide hand-off this part allows, to decide whether a
// persistent decode kernel (weights of the next stage prefetched while waiting) can beat a launch boundary + first-load latency
// (1.5 + ~2 us).  grid_sync.hip measured the naive form (one counter, __threadfence() on both sides): 10..40 us.
// Here: payload written with write-through stores (sc0 sc1) and read with L2-bypassing loads (sc0 sc1), NO fence (a fence is
// an L2 write-back + invalidate on a part with 8 non-coherent L2s), relaxed agent-scope atomics, and three arrival schemes:
//   flat    every workgroup adds to one counter and polls it
//   tree    per-XCD counter (workgroup id mod 8 = XCD); the last arriver of an XCD adds to the global counter; all poll the global one
//   flag    as tree, but the last global arriver writes a generation word that everybody polls (polling a word nobody adds to)
// Each stage: write slice, arrive, wait, read ANOTHER workgroup's slice (other XCD) and check it.  err != 0 = stale data seen.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/grid_sync2.bin tools/microbench/grid_sync2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_wt(float4* p, float4 v) {
    const f4v r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ float4 ld_byp(const float4* p) {
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned ld_word(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Ctl {
    unsigned global;        // arrivals (flat) or XCD-leader arrivals (tree/flag)
    unsigned pad0[31];
    unsigned gen;           // generation word (flag)
    unsigned pad1[31];
    unsigned xcd[8 * 32];   // per-XCD counters, one 128-byte line each
};

template <int MODE>   // 0 flat, 1 tree, 2 flag
__global__ __launch_bounds__(256) void sync_kernel(Ctl* c, float* payload, int stages, int payload_f4, float* out, unsigned* err,
                                                   int sleep) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    const int per_xcd = nb / 8;
    float acc = 0.f;
    unsigned bad = 0;
    for (int s = 1; s <= stages; ++s) {
        float4* mine = reinterpret_cast<float4*>(payload) + ((int64_t)(s & 1) * nb + b) * payload_f4;
        for (int i = tid; i < payload_f4; i += 256) st_wt(mine + i, make_float4((float)s, 1.f, 2.f, 3.f));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // write-through stores acknowledged
        __syncthreads();
        if (tid == 0) {
            if (MODE == 0) {
                __hip_atomic_fetch_add(&c->global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)nb * (unsigned)s;
                int spins = 0;
                while (ld_word(&c->global) < target) {
                    if (sleep) __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { *err = 1; break; }
                }
            } else {
                const unsigned old = __hip_atomic_fetch_add(&c->xcd[(b & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)per_xcd * (unsigned)s - 1u) {
                    const unsigned g = __hip_atomic_fetch_add(&c->global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (MODE == 2 && g == 8u * (unsigned)s - 1u)
                        __hip_atomic_store(&c->gen, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int spins = 0;
                if (MODE == 1) {
                    while (ld_word(&c->global) < 8u * (unsigned)s) {
                        if (sleep) __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 22)) { *err = 1; break; }
                    }
                } else {
                    while (ld_word(&c->gen) < (unsigned)s) {
                        if (sleep) __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 22)) { *err = 1; break; }
                    }
                }
            }
        }
        __syncthreads();
        const int src = (b + 37) % nb;
        const float4* theirs = reinterpret_cast<const float4*>(payload) + ((int64_t)(s & 1) * nb + src) * payload_f4;
        for (int i = tid; i < payload_f4; i += 256) {
            const float4 v = ld_byp(theirs + i);
            acc += v.x;
            if (v.x != (float)s) bad = 1;
        }
    }
    if (bad) *err = 2;
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    Ctl* ctl;
    unsigned* err;
    float *payload, *out;
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&err, 4); hipMalloc(&out, 4);
    hipMalloc(&payload, (size_t)1024 * 65536);
    const int stages = 2000;
    const char* names[3] = {"flat", "tree", "flag"};
    for (int mode = 0; mode < 3; ++mode)
        for (int sleep = 0; sleep < 2; ++sleep)
            for (int nb : {64, 256}) {
                for (int pf4 : {0, 64, 512}) {                 // 0 B, 1 KiB, 8 KiB per workgroup per stage
                    hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(err, 0, 4);
                    hipEvent_t a, b;
                    hipEventCreate(&a); hipEventCreate(&b);
                    hipEventRecord(a);
                    if (mode == 0) hipLaunchKernelGGL(sync_kernel<0>, dim3(nb), dim3(256), 0, 0, ctl, payload, stages, pf4, out, err, sleep);
                    if (mode == 1) hipLaunchKernelGGL(sync_kernel<1>, dim3(nb), dim3(256), 0, 0, ctl, payload, stages, pf4, out, err, sleep);
                    if (mode == 2) hipLaunchKernelGGL(sync_kernel<2>, dim3(nb), dim3(256), 0, 0, ctl, payload, stages, pf4, out, err, sleep);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms = 0;
                    hipEventElapsedTime(&ms, a, b);
                    unsigned e = 0;
                    hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
                    printf("%s sleep %d workgroups %4d payload %6d B: %7.3f us per hand-off (err %u)\n", names[mode], sleep, nb,
                           pf4 * 16, ms * 1e3 / stages, e);
                }
            }
    return 0;
}
