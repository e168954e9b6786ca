// Developer probe (GPU box): which (row, k-block) does lane p's scale byte of v_mfma_scale_f32_32x32x64_f8f6f4 apply to?
// A = 1.0 everywhere; B[j][k] = 1.0 (k < 32), 0.5 (k >= 32); every scale 127 (x1) except lane p of scale_a (or scale_b) = 128 (x2).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mx8_probe.bin tools/microbench/mx8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OPSEL>
__global__ void k(float* D, int p, int which, int byte) {
    const int l = threadIdx.x, r = l & 31, kh = l >> 5;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = 0x38383838; b[j] = kh ? 0x30303030 : 0x38383838; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    int wa = 0x7F7F7F7F, wb = 0x7F7F7F7F;
    if (l == p) { if (which == 0) wa += 1 << (8 * byte); else wb += 1 << (8 * byte); }
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, wa, OPSEL, wb);
    for (int i = 0; i < 16; ++i) D[(8 * (i >> 2) + 4 * kh + (i & 3)) * 32 + r] = c[i];
}
int main() {
    float* D; hipMalloc(&D, 4096);
    float h[1024];
    for (int which = 0; which < 2; ++which)
        for (int p : {0, 1, 5, 31, 32, 33, 63})
            for (int byte = 0; byte < 4; ++byte)
                for (int opsel = 0; opsel < 4; ++opsel) {
                    if (opsel == 0) k<0><<<1, 64>>>(D, p, which, byte);
                    else if (opsel == 1) k<1><<<1, 64>>>(D, p, which, byte);
                    else if (opsel == 2) k<2><<<1, 64>>>(D, p, which, byte);
                    else k<3><<<1, 64>>>(D, p, which, byte);
                    hipMemcpy(h, D, 4096, hipMemcpyDeviceToHost);
                    int n = 0, i0 = -1, j0 = -1, i1 = -1, j1 = -1; float v = 0;
                    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (h[i * 32 + j] != 48.f) { if (!n) { i0 = i; j0 = j; v = h[i * 32 + j]; } i1 = i; j1 = j; ++n; }
                    if (n) printf("scale_%c lane %2d byte %d opsel %d: %4d elements changed, rows %d..%d cols %d..%d, value %.0f (80 = k-block 0 doubled, 64 = k-block 1)\n", which ? 'b' : 'a', p, byte, opsel, n, i0, i1, j0, j1, v);
                }
    printf("base value D[0][0] unscaled expectation 48\n");
    return 0;
}
