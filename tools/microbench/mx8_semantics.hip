// Developer microtest (GPU box), round 6: operand and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 inputs, checked
// against a host evaluation.  Found with mx8_probe.hip and verified here: lane l = (row or column l & 31, half h = l >> 5) supplies
// 32 bytes; byte t of them is k = 32 (t / 16) + 16 h + t % 16 (the instruction is two k32 steps back to back, each in the order of the
// K = 32 forms); scale VGPR byte `opsel` of lane (row, h) is the E8M0 scale (2^(s - 127)) of the row's k-block h = k 32 h .. 32 h + 31,
// i.e. of bytes 16 h .. 16 h + 15 of BOTH lanes of the row -- not of the lane's own 32 bytes;
// C/D as every 32x32 MFMA: D[i = 8 (r >> 2) + 4 (l >> 5) + (r & 3)][j = l & 31] with A rows i, B columns j.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mx8_semantics.bin tools/microbench/mx8_semantics.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* D, int mode) {
    const int l = threadIdx.x, r = l & 31, kh = l >> 5;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = *reinterpret_cast<const int*>(A + r * 64 + kh * 32 + 4 * j);
        b[j] = *reinterpret_cast<const int*>(B + r * 64 + kh * 32 + 4 * j);     // B stored as [col][k]
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    // scale words: byte 0 unused garbage, byte 1 = the real scale when opsel = 1 (checks the byte select)
    // scale words: byte q holds scale set q (sa + 64 q): the host checks which set the hardware used for each opsel
    int wa = 0, wb = 0;
    for (int q = 0; q < 4; ++q) { wa |= sa[64 * q + r * 2 + kh] << (8 * q); wb |= sb[64 * q + r * 2 + kh] << (8 * q); }
    if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);            // scales folded away
    else if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, wa, 0, wb);
    else if (mode == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, wa, 1, wb);
    else if (mode == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, wa, 2, wb);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, wa, 3, wb);
    for (int i = 0; i < 16; ++i) D[(8 * (i >> 2) + 4 * kh + (i & 3)) * 32 + r] = c[i];
}
static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 15 && m == 7) x = NAN;
    else if (e == 0) x = ldexpf((float)m, -9);
    else x = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
int main(int argc, char**) {
    const bool full = argc > 1;      // any argument: e4m3 bytes over the whole finite range (|x| <= 448) instead of |x| <= 2
    uint8_t hA[32 * 64], hB[32 * 64], hsa[256], hsb[256];
    uint32_t s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int i = 0; i < 2048; ++i) { hA[i] = (rnd() & 0x80) | (rnd() % (full ? 0x7F : 0x40)); hB[i] = (rnd() & 0x80) | (rnd() % (full ? 0x7F : 0x40)); }
    for (int i = 0; i < 256; ++i) { hsa[i] = 120 + rnd() % 12; hsb[i] = 125 + rnd() % 5; }
    uint8_t *A, *B, *sa, *sb; float* D;
    hipMalloc(&A, 2048); hipMalloc(&B, 2048); hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&D, 4096);
    hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
    hipMemcpy(sa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(sb, hsb, 256, hipMemcpyHostToDevice);
    // e4m3 values drawn small (|x| <= 2) so that fp32 accumulation noise is far below a wrong-scale error
    for (int mode = 0; mode < 5; ++mode) {
        k<<<1, 64>>>(A, B, sa, sb, D, mode);
        float hD[1024];
        hipMemcpy(hD, D, 4096, hipMemcpyDeviceToHost);
        for (int q = 0; q < (mode ? 4 : 1); ++q) {
            double worst = 0, mag = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0;
                    for (int kk = 0; kk < 64; ++kk) {
                        double fa = e4m3(hA[i * 64 + kk]), fb = e4m3(hB[j * 64 + kk]);
                        if (mode) { fa *= ldexp(1.0, hsa[64 * q + i * 2 + (kk % 32) / 16] - 127); fb *= ldexp(1.0, hsb[64 * q + j * 2 + (kk % 32) / 16] - 127); }
                        ref += fa * fb;
                    }
                    worst = fmax(worst, fabs(ref - hD[i * 32 + j]));
                    mag = fmax(mag, fabs(ref));
                }
            printf("mode %d (%s) vs scale set %d: max |D - ref| = %.3e, max |ref| = %.3e  -> %s\n", mode, mode ? "scaled, opsel = mode - 1" : "scale operands 0 (unscaled form)", q, worst, mag,
                   worst <= 1e-4 * mag ? "OK (within the fp8 pipe's own accumulation precision, ~4e-5 of the largest element)" : "mismatch");
        }
    }
    return 0;
}
