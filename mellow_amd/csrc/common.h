// Shared device/host helpers for the Mellow gfx950 engine.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mellow {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MELLOW_WAVE 64

// ---- cross-lane reductions without LDS traffic ---------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip + s_waitcnt) on gfx950.  A 16-lane all-reduce is
// instead four DPP-modified VALU ops: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
// (after the two quad steps every quad is uniform, so the mirrors exchange quads / 8-lane halves).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {       // all 16 lanes of a DPP row end with the row's sum
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
// lane ^ 16 inside each 32-lane half (ds_swizzle bit mode: and 0x1F, or 0, xor 0x10)
__device__ __forceinline__ float swz_xor16(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
// combine the two 32-lane halves: v_permlane32_swap gives every lane its own and the other half's value
__device__ __forceinline__ float half_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += swz_xor16(v);
    return half_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, swz_xor16(v));
    return half_max(v);
}

// torch.argmax order (reference wrapper.py:232): NaN compares as the maximum, and among equal values (or several NaNs)
// the lowest index wins.  With (-inf, INT_MAX) as the start value the result is always a valid index.
__device__ __forceinline__ bool arg_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

// exact-erf GELU (nn.GELU() default, reference htsat.py:121 / mellow.py:50)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x / (1.0f + expf(-x)); }

// ---- fp32 = exact sum of three bf16 (gemm_bf16x3.hip) ------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = static_cast<__bf16>(a);                       // round to nearest even (v_cvt_pk_bf16_f32)
    const float r1 = a - static_cast<float>(h);       // exact
    m = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(m);      // exact, at most 8 significant bits
    l = static_cast<__bf16>(r2);                      // exact
}
__device__ __forceinline__ void split8(const float (&v)[8], i32x4& p0, i32x4& p1, i32x4& p2) {
#ifdef MELLOW_X3_FAKESPLIT      // developer A/B build only: wrong numbers, no VALU work (is the split what bounds the loop?)
    p0 = i32x4{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
    p1 = i32x4{__float_as_int(v[4]), __float_as_int(v[5]), __float_as_int(v[6]), __float_as_int(v[7])};
    p2 = p0;
    return;
#endif
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        split3(v[j], a, b, c);
        h[j] = a; m[j] = b; l[j] = c;
    }
    p0 = __builtin_bit_cast(i32x4, h);
    p1 = __builtin_bit_cast(i32x4, m);
    p2 = __builtin_bit_cast(i32x4, l);
}
// "APB": an activation matrix [M][K] pre-split for the x3q GEMM, in the order its LDS stage wants it.  16-byte slot index of
// (row m, columns 8*k8 .. 8*k8+7, piece pc): ((m/128 * K/16 + k8/2) * 12 + pc * 4 + (m/32)%4) * 64 + m%32 + 32 * (k8 % 2)
__device__ __forceinline__ int64_t apb_slot(int64_t m, int k8, int KT) {
    return (((m >> 7) * KT + (k8 >> 1)) * 12 + ((m >> 5) & 3)) * 64 + (m & 31) + 32 * (k8 & 1);
}
__device__ __forceinline__ void apb_store8(i32x4* apb, int64_t m, int k8, int KT, const float (&w)[8]) {
    i32x4 p0, p1, p2;
    split8(w, p0, p1, p2);
    i32x4* o = apb + apb_slot(m, k8, KT);
    o[0] = p0; o[4 * 64] = p1; o[8 * 64] = p2;
}
// The MFMA epilogues hold a row's columns as quads: lane (h = lane / 32, row = lane % 32) owns columns 8*gq + 4*h + (0..3) of
// quad group gq.  X = the quad of an even group, Y = of the next (odd) group: one v_permlane32_swap per value gives the lower
// half-wave all eight columns of the even group and the upper half-wave those of the odd group.
__device__ __forceinline__ void apb_store_quads(i32x4* apb, int64_t m, int k8_even, int KT, const float (&X)[4], const float (&Y)[4], int h) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[j]), __float_as_uint(Y[j]), false, false);
        w[j] = __uint_as_float(r[0]);
        w[4 + j] = __uint_as_float(r[1]);
    }
    apb_store8(apb, m, k8_even + h, KT, w);
}

// ---- "AMX": an activation matrix [M][K] as MXFP8 (e4m3 elements, one E8M0 scale per 32 consecutive k of a row) in the order the
// LDS stage of gemm_mx8_kernel wants it (gemm_fp8.hip has the derivation of the operand order from the instruction) -------------
// 16-byte slot of (row m, k64 step s, 32-block j of the step, 16-k half h of the block)
__device__ __forceinline__ int64_t amx_slot(int64_t m, int s, int j, int h, int KT64) {
    return (((((m >> 7) * KT64 + s) * 4 + ((m >> 5) & 3)) * 2 + j) * 64) + (m & 31) + 32 * h;
}
// byte address of the scale of (row m, block 2 s + j): word [panel][s / 4][m-tile][lane = m % 32 + 32 j], byte s % 4
__device__ __forceinline__ int64_t amx_scale_byte(int64_t m, int s, int j, int KQ) {
    return (((((m >> 7) * KQ + (s >> 2)) * 4 + ((m >> 5) & 3)) * 64) + (m & 31) + 32 * j) * 4 + (s & 3);
}
// biased exponent e of the block scale 2^(e - 127) = the smallest power of two with amax / scale <= 448 (no element clips);
// an all-zero block takes 2^0.  inverse = 2^(127 - e) exactly.
__device__ __forceinline__ int mx8_exponent(float amax) {
    const uint32_t b = __float_as_uint(amax * (1.0f / 448.0f));
    int e = (int)((b >> 23) & 0xFF) + ((b & 0x7FFFFF) ? 1 : 0);
    e = e < 1 ? 1 : (e > 253 ? 253 : e);
    return amax > 0.f ? e : 127;
}
__device__ __forceinline__ float mx8_inverse(int e) { return __uint_as_float((uint32_t)(254 - e) << 23); }
__device__ __forceinline__ int mx8_pack4(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
}
// The MFMA epilogues (and the attention epilogues) hold one 32-column block of a row in a lane PAIR: lane (row, h = lane / 32) owns
// columns 8 g + 4 h + (0..3), g = 0..3, as v[4 g .. 4 g + 3].  Block scale from both lanes' values, quantise, then two dword-level
// v_permlane32_swap give the lower lane columns 0..15 and the upper lane columns 16..31: each stores one whole 16-byte slot.
__device__ __forceinline__ void amx_store_block(i32x4* img, uint8_t* sc, int64_t m, int kb, int KT64, int KQ, const float (&v)[16], int h) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = half_max(amax);
    const int e = mx8_exponent(amax);
    const float inv = mx8_inverse(e);
    int w[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) w[g] = mx8_pack4(v[4 * g] * inv, v[4 * g + 1] * inv, v[4 * g + 2] * inv, v[4 * g + 3] * inv);
    // lower lane: (own g0, partner g0, own g1, partner g1) = columns 0..15; upper lane: (partner g2, own g2, partner g3, own g3) = 16..31
    auto r0 = __builtin_amdgcn_permlane32_swap((uint32_t)w[0], (uint32_t)w[2], false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap((uint32_t)w[1], (uint32_t)w[3], false, false);
    const int s = kb >> 1, j = kb & 1;
    img[amx_slot(m, s, j, h, KT64)] = i32x4{(int)r0[0], (int)r0[1], (int)r1[0], (int)r1[1]};
    if (h == 0) sc[amx_scale_byte(m, s, j, KQ)] = (uint8_t)e;
}

// the same for a lane that already holds 16 CONSECUTIVE columns of its row: y = columns 32 kb + 16 h .. + 15 (norm kernels, quantiser)
__device__ __forceinline__ void amx_store16(i32x4* img, uint8_t* sc, int64_t m, int kb, int KT64, int KQ, const float (&y)[16], int h) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(y[i]));
    amax = half_max(amax);
    const int e = mx8_exponent(amax);
    const float inv = mx8_inverse(e);
    i32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = mx8_pack4(y[4 * q] * inv, y[4 * q + 1] * inv, y[4 * q + 2] * inv, y[4 * q + 3] * inv);
    const int s = kb >> 1, j = kb & 1;
    img[amx_slot(m, s, j, h, KT64)] = w;
    if (h == 0) sc[amx_scale_byte(m, s, j, KQ)] = (uint8_t)e;
}

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): logical ids that are consecutive land on
// the same XCD so tiles sharing an operand panel hit one L2.  Bijective for any block count.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

}  // namespace mellow
