// Developer microbenchmark (GPU box): the main loop of gemm_f32_kernel<2,2,*,32> (128x128x32 tile, 4 waves, 2 workgroups
// per CU) rebuilt with switchable parts, to find which part keeps the MFMA pipe below the 155 TFLOP/s that a pure MFMA
// loop reaches (mfma_peak.hip).  Same instruction mix per k-tile: 64 MFMA, 16 ds_read_b128, 8 ds_write_b128,
// 8 global_load_dwordx4, 1 barrier.  Addresses are synthetic (linear), results are meaningless.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/gemm_ablate.bin tools/microbench/gemm_ablate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { F_GLOAD = 1, F_STORE = 2, F_BARRIER = 4, F_LDSREAD = 8, F_STORE_LATE = 16, F_L2HOT = 32, F_DEEP = 64, F_STAMP = 128, F_DIRECT = 256 };
__device__ unsigned long long g_stamp[8];

#define MFMA16(w0, w1, a0, a1)                                                                        \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, a0.x, acc[0], 0, 0, 0);                       \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, a1.x, acc[1], 0, 0, 0);                       \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, a0.x, acc[2], 0, 0, 0);                       \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, a1.x, acc[3], 0, 0, 0);                       \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, a0.y, acc[0], 0, 0, 0);                       \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, a1.y, acc[1], 0, 0, 0);                       \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, a0.y, acc[2], 0, 0, 0);                       \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, a1.y, acc[3], 0, 0, 0);                       \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, a0.z, acc[0], 0, 0, 0);                       \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, a1.z, acc[1], 0, 0, 0);                       \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, a0.z, acc[2], 0, 0, 0);                       \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, a1.z, acc[3], 0, 0, 0);                       \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, a0.w, acc[0], 0, 0, 0);                       \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, a1.w, acc[1], 0, 0, 0);                       \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, a0.w, acc[2], 0, 0, 0);                       \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, a1.w, acc[3], 0, 0, 0);

template <int FL>
__global__ __launch_bounds__(256) void loop_kernel(const f32x4* __restrict__ A, const f32x4* __restrict__ W, float* out, int KT) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];   // [2 stages][A 1024 f4 | W 1024 f4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // a private 1 MiB stream per workgroup and operand (HBM-bound), or with F_L2HOT 16 streams shared by all workgroups
    // (L2 hits, like the real kernel whose panels are re-read by the other tiles of the same row/column)
    const int stream = (FL & F_L2HOT) ? (blockIdx.x & 15) : blockIdx.x;
    const f32x4* ap = A + (int64_t)stream * 1024 * 64 + tid;
    const f32x4* wp = W + (int64_t)stream * 1024 * 64 + tid;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 ra[4], rw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { ra[q] = ap[q * 256]; rw[q] = wp[q * 256]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { smem[q * 256 + tid] = ra[q]; smem[1024 + q * 256 + tid] = rw[q]; smem[2048 + q * 256 + tid] = ra[q]; smem[3072 + q * 256 + tid] = rw[q]; }
    __syncthreads();
    f32x4 xa0 = smem[(2 * wm) * 64 + lane], xa1 = smem[(2 * wm + 1) * 64 + lane];
    f32x4 xw0 = smem[1024 + (2 * wn) * 64 + lane], xw1 = smem[1024 + (2 * wn + 1) * 64 + lane];
    f32x4 ya0 = xa0, ya1 = xa1, yw0 = xw0, yw1 = xw1;
    unsigned long long w_vm = 0, w_bar = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (FL & F_DIRECT) {
            // global -> LDS without a VGPR round trip: lane i of the wave lands at base + 16 i
            f32x4* dst = smem + (cur ^ 1) * 2048 + wave * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_global_load_lds(ap + (kt & 63) * 1024 + q * 256, dst + q * 256, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(wp + (kt & 63) * 1024 + q * 256, dst + 1024 + q * 256, 16, 0, 0);
            }
        } else if (FL & F_GLOAD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { ra[q] = ap[(kt & 63) * 1024 + q * 256]; rw[q] = wp[(kt & 63) * 1024 + q * 256]; }
        }
        const f32x4* Ac = smem + cur * 2048 + (2 * wm) * 64 + lane;
        const f32x4* Wc = smem + cur * 2048 + 1024 + (2 * wn) * 64 + lane;
        f32x4* An = smem + (cur ^ 1) * 2048;
        if (FL & F_LDSREAD) { xa0 = Ac[0]; xa1 = Ac[64]; xw0 = Wc[0]; xw1 = Wc[64]; }
#pragma unroll
        for (int k8 = 0; k8 < 4; k8 += 2) {
            if (FL & F_LDSREAD) { ya0 = Ac[(k8 + 1) * 4 * 64]; ya1 = Ac[((k8 + 1) * 4 + 1) * 64]; yw0 = Wc[(k8 + 1) * 4 * 64]; yw1 = Wc[((k8 + 1) * 4 + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
            MFMA16(xw0, xw1, xa0, xa1)
            if (k8 + 2 < 4) {
                if (FL & F_LDSREAD) { xa0 = Ac[(k8 + 2) * 4 * 64]; xa1 = Ac[((k8 + 2) * 4 + 1) * 64]; xw0 = Wc[(k8 + 2) * 4 * 64]; xw1 = Wc[((k8 + 2) * 4 + 1) * 64]; }
            } else if ((FL & F_STORE) && !(FL & F_STORE_LATE)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { An[q * 256 + tid] = ra[q]; An[1024 + q * 256 + tid] = rw[q]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            MFMA16(yw0, yw1, ya0, ya1)
        }
        unsigned long long t0 = 0;
        if (FL & F_STAMP) {
            t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            w_vm += __builtin_readcyclecounter() - t0;
            t0 = __builtin_readcyclecounter();
        }
        if ((FL & F_STORE) && (FL & F_STORE_LATE)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { An[q * 256 + tid] = ra[q]; An[1024 + q * 256 + tid] = rw[q]; }
        }
        if (FL & F_DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FL & F_BARRIER) __syncthreads();
        if (FL & F_STAMP) w_bar += __builtin_readcyclecounter() - t0;
    }
    if ((FL & F_STAMP) && blockIdx.x == gridDim.x / 2 && tid == 0) {
        g_stamp[0] = w_vm; g_stamp[1] = w_bar; g_stamp[2] = __builtin_readcyclecounter() - t_begin; g_stamp[3] = KT;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    s += ra[0].x + rw[0].x;
    if (s == 123.456f) out[0] = s;
}


// 8-wave ping-pong: one workgroup of 512 threads per CU, tile 256 x 128 x 32, 96 KB LDS.  Waves 0-3 (group 0) and 4-7
// (group 1) share the SIMDs pairwise and alternate: one group issues its 64 MFMAs while the other refills LDS, issues its
// global prefetch and waits at the barrier.
template <int FL>
__global__ __launch_bounds__(512) void pingpong_kernel(const f32x4* __restrict__ A, const f32x4* __restrict__ W, float* out, int KT) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];   // [2 stages][A 2048 f4 | W 1024 f4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, group = wave >> 2;
    const int stream = blockIdx.x & 15;
    const f32x4* ap = A + (int64_t)stream * 1024 * 64 + tid;
    const f32x4* wp = W + (int64_t)stream * 1024 * 64 + tid;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 ra[4], rw[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = ap[q * 512];
#pragma unroll
    for (int q = 0; q < 2; ++q) rw[q] = wp[q * 512];
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) smem[st * 3072 + q * 512 + tid] = ra[q];
#pragma unroll
        for (int q = 0; q < 2; ++q) smem[st * 3072 + 2048 + q * 512 + tid] = rw[q];
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (group == ph) {
                const f32x4* Ac = smem + cur * 3072 + (2 * wm) * 64 + lane;            // A fragments: [k8][8 m-tiles][64]
                const f32x4* Wc = smem + cur * 3072 + 2048 + (2 * wn) * 64 + lane;     // W fragments: [k8][4 n-tiles][64]
                f32x4 xa0 = Ac[0], xa1 = Ac[64], xw0 = Wc[0], xw1 = Wc[64];
#pragma unroll
                for (int k8 = 0; k8 < 4; k8 += 2) {
                    const f32x4 ya0 = Ac[(k8 + 1) * 8 * 64], ya1 = Ac[((k8 + 1) * 8 + 1) * 64];
                    const f32x4 yw0 = Wc[(k8 + 1) * 4 * 64], yw1 = Wc[((k8 + 1) * 4 + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    MFMA16(xw0, xw1, xa0, xa1)
                    if (k8 + 2 < 4) { xa0 = Ac[(k8 + 2) * 8 * 64]; xa1 = Ac[((k8 + 2) * 8 + 1) * 64]; xw0 = Wc[(k8 + 2) * 4 * 64]; xw1 = Wc[((k8 + 2) * 4 + 1) * 64]; }
                    __builtin_amdgcn_sched_barrier(0);
                    MFMA16(yw0, yw1, ya0, ya1)
                }
            } else {
                f32x4* An = smem + (cur ^ 1) * 3072;
                if (FL & F_STORE) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) An[q * 512 + tid] = ra[q];
#pragma unroll
                    for (int q = 0; q < 2; ++q) An[2048 + q * 512 + tid] = rw[q];
                }
                if (FL & F_GLOAD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ra[q] = ap[(kt & 31) * 2048 + q * 512];
#pragma unroll
                    for (int q = 0; q < 2; ++q) rw[q] = wp[(kt & 31) * 2048 + q * 512];
                }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    s += ra[0].x + rw[0].x;
    if (s == 123.456f) out[0] = s;
}
template <int FL>
static void run_pp(const char* name, const f32x4* A, const f32x4* W, float* out, int blocks, int KT) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pingpong_kernel<FL>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((pingpong_kernel<FL>), dim3(blocks), dim3(512), 98304, 0, A, W, out, KT);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((pingpong_kernel<FL>), dim3(blocks), dim3(512), 98304, 0, A, W, out, KT);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double fl = (double)blocks * 8 * KT * 64 * (32.0 * 32 * 2 * 2);
    printf("%-58s blocks %5d KT %3d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, KT, ms, fl / ms / 1e9);
}

template <int FL>
static void run(const char* name, const f32x4* A, const f32x4* W, float* out, int blocks, int KT) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<FL>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((loop_kernel<FL>), dim3(blocks), dim3(256), 65536, 0, A, W, out, KT);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((loop_kernel<FL>), dim3(blocks), dim3(256), 65536, 0, A, W, out, KT);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double fl = (double)blocks * 4 * KT * 64 * (32.0 * 32 * 2 * 2);
    printf("%-58s blocks %5d KT %3d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, KT, ms, fl / ms / 1e9);
    if (FL & F_STAMP) {
        unsigned long long h[8];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
        printf("      per k-tile: total %.0f clocks, wait-for-global %.0f, lds-write+barrier %.0f\n", (double)h[2] / h[3], (double)h[0] / h[3], (double)h[1] / h[3]);
    }
}

int main() {
    const int blocks = 2560;
    f32x4 *A, *W;
    float* out;
    hipMalloc(&A, (size_t)blocks * 1024 * 64 * 16);
    hipMalloc(&W, (size_t)blocks * 1024 * 64 * 16);
    hipMemset(A, 0, (size_t)blocks * 1024 * 64 * 16);
    hipMemset(W, 0, (size_t)blocks * 1024 * 64 * 16);
    hipMalloc(&out, 4);
    for (int KT : {18, 64}) {
        run<0>("mfma only", A, W, out, blocks, KT);
        run<F_BARRIER>("mfma + barrier", A, W, out, blocks, KT);
        run<F_LDSREAD>("mfma + lds reads", A, W, out, blocks, KT);
        run<F_LDSREAD | F_BARRIER>("mfma + lds reads + barrier", A, W, out, blocks, KT);
        run<F_LDSREAD | F_STORE | F_BARRIER>("mfma + lds reads + lds writes(mid) + barrier", A, W, out, blocks, KT);
        run<F_LDSREAD | F_STORE | F_STORE_LATE | F_BARRIER>("mfma + lds reads + lds writes(end) + barrier", A, W, out, blocks, KT);
        run<F_GLOAD>("mfma + global loads", A, W, out, blocks, KT);
        run<F_GLOAD | F_LDSREAD | F_STORE | F_BARRIER>("full loop (writes mid)", A, W, out, blocks, KT);
        run<F_GLOAD | F_LDSREAD | F_STORE | F_STORE_LATE | F_BARRIER>("full loop (writes end)", A, W, out, blocks, KT);
        run<F_L2HOT | F_GLOAD | F_LDSREAD | F_STORE | F_BARRIER>("full loop, L2-hot operands (writes mid)", A, W, out, blocks, KT);
        run<F_L2HOT | F_GLOAD | F_LDSREAD | F_STORE | F_STORE_LATE | F_BARRIER>("full loop, L2-hot operands (writes end)", A, W, out, blocks, KT);
        run<F_STAMP | F_L2HOT | F_GLOAD | F_LDSREAD | F_STORE | F_STORE_LATE | F_BARRIER>("full loop, L2-hot (writes end), stamped", A, W, out, blocks, KT);
        run<F_STAMP | F_LDSREAD | F_STORE | F_STORE_LATE | F_BARRIER>("no global loads (writes end), stamped", A, W, out, blocks, KT);
        run<F_L2HOT | F_DIRECT | F_LDSREAD | F_BARRIER>("full loop, L2-hot, direct-to-LDS loads", A, W, out, blocks, KT);
        run<F_DIRECT | F_LDSREAD | F_BARRIER>("full loop, private streams, direct-to-LDS loads", A, W, out, blocks, KT);
        run_pp<F_STORE | F_GLOAD>("8-wave ping-pong 256x128, full loop, L2-hot", A, W, out, blocks / 2, KT);
        run_pp<0>("8-wave ping-pong 256x128, mfma + lds reads + barriers", A, W, out, blocks / 2, KT);
        run_pp<F_STORE>("8-wave ping-pong 256x128, + lds writes only", A, W, out, blocks / 2, KT);
        run_pp<F_GLOAD>("8-wave ping-pong 256x128, + global loads only", A, W, out, blocks / 2, KT);
    }
    return 0;
}
