// Developer microbenchmark (GPU box): sustained v_mfma_f32_32x32x2_f32 / 16x16x4_f32 rate with no memory traffic,
// at 1, 2 and 4 waves per SIMD.  Answers "what fraction of the 157.3 TFLOP/s datasheet figure can any kernel reach".
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_peak.bin tools/microbench/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma32_kernel(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void mfma16_kernel(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_fp8_kernel(float* out, int iters, long a, long b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters, int a0, int b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bf16x8 a = __builtin_bit_cast(bf16x8, i32x4{a0, a0, a0, a0}), b = __builtin_bit_cast(bf16x8, i32x4{b0, b0, b0, b0});
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void mfma_f8f6f4_kernel(float* out, int iters, int a0, int b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    i32x8 a = {a0, a0, a0, a0, a0, a0, a0, a0}, b = {b0, b0, b0, b0, b0, b0, b0, b0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, 127, 0, 127);   // fp8 x fp8, unit scales
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <typename F>
static double time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD = workgroups of 4 waves per CU
        const int blocks = cus * wps;
        {
            const double ms = time_ms([&] { hipLaunchKernelGGL((mfma32_kernel<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            const double fl = (double)blocks * 4 * iters * 16 * 4 * (32.0 * 32 * 2 * 2);
            printf("32x32x2 f32, 4 acc, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
        {
            const double ms = time_ms([&] { hipLaunchKernelGGL((mfma32_kernel<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            const double fl = (double)blocks * 4 * iters * 16 * 1 * (32.0 * 32 * 2 * 2);
            printf("32x32x2 f32, 1 acc (dependent chain), %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
        {
            const double ms = time_ms([&] { hipLaunchKernelGGL(mfma_fp8_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0x3838383838383838L, 0x3838383838383838L); });
            const double fl = (double)blocks * 4 * iters * 16 * 4 * (32.0 * 32 * 16 * 2);
            printf("32x32x16 fp8 (e4m3), 4 acc, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
        {
            const double ms = time_ms([&] { hipLaunchKernelGGL(mfma_bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0x3f803f80, 0x3f803f80); });
            const double fl = (double)blocks * 4 * iters * 16 * 4 * (32.0 * 32 * 16 * 2);
            printf("32x32x16 bf16, 4 acc, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
        {
            const double ms = time_ms([&] { hipLaunchKernelGGL(mfma_f8f6f4_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0x38383838, 0x38383838); });
            const double fl = (double)blocks * 4 * iters * 16 * 4 * (32.0 * 32 * 64 * 2);
            printf("32x32x64 f8f6f4 (fp8 x fp8, scaled), 4 acc, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
        {
            const double ms = time_ms([&] { hipLaunchKernelGGL(mfma16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            const double fl = (double)blocks * 4 * iters * 16 * 4 * (16.0 * 16 * 4 * 2);
            printf("16x16x4 f32, 4 acc, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
