// Developer microbenchmark (GPU box): cost of a device-wide dependency hand-off through memory on MI355X (8 XCDs with
// private L2s): N resident workgroups run `stages` rounds of { write a payload, release-fence, atomic counter += 1,
// spin until the counter reaches nblocks * round, acquire-fence, read another workgroup's payload }.
// Answers whether a persistent flag-synchronised decode kernel could beat ~2 us per dependent launch.
// Measured (profiles/r01_microbench.txt): 10 us (64 workgroups) .. 40 us (256) per hand-off -- agent-scope atomics on one
// address serialise at ~70 ns each and every fence is an L2 write-back/invalidate -- so the answer is no.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/grid_sync.bin tools/microbench/grid_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void sync_kernel(unsigned* counter, float* payload, int stages, int payload_f4, float* out,
                                                   unsigned* err) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int s = 1; s <= stages; ++s) {
        // payload: this workgroup's slice
        for (int i = tid; i < payload_f4; i += 256)
            reinterpret_cast<float4*>(payload)[((int64_t)(s & 1) * nb + b) * payload_f4 + i] = make_float4((float)s, 1.f, 2.f, 3.f);   // double-buffered: a workgroup may run one stage ahead of its reader
        __threadfence();                                                              // release: every wave drains and writes back its stores
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)nb * (unsigned)s;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *err = 1; break; }                       // never hang the box
            }
        }
        __syncthreads();
        __threadfence();                                                              // acquire: drop stale lines before reading
        // consume a neighbour's slice (written on another CU / possibly another XCD)
        const int src = (b + 37) % nb;
        for (int i = tid; i < payload_f4; i += 256) {
            const float4 v = reinterpret_cast<const float4*>(payload)[((int64_t)(s & 1) * nb + src) * payload_f4 + i];
            acc += v.x;
            if (v.x != (float)s) *err = 2;
        }
        __syncthreads();
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    unsigned *counter, *err;
    float *payload, *out;
    hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&out, 4);
    hipMalloc(&payload, (size_t)1024 * 65536);
    const int stages = 2000;
    for (int nb : {64, 256, 512}) {
        for (int pf4 : {0, 64, 1024}) {                 // 0 B, 1 KiB, 16 KiB per workgroup per stage
            hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(sync_kernel, dim3(nb), dim3(256), 0, 0, counter, payload, stages, pf4, out, err);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            unsigned e = 0;
            hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            printf("workgroups %4d payload %6d B: %7.3f us per hand-off (err %u)\n", nb, pf4 * 16, ms * 1e3 / stages, e);
        }
    }
    return 0;
}
