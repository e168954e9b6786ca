// Epilogues of the MFMA GEMMs, shared by the fp32 kernel (gemm_f32.hip) and the fp8 kernel (gemm_fp8.hip): both issue
// v_mfma 32x32 tiles as D[n][m] (weight = A operand, activation = B operand), so a lane owns ONE output row
// m_local = lane & 31 of each 32-row tile and, in accumulator quad gq, the four consecutive columns 8 gq + 4 (lane >> 5).
// acc[ni][mi] = 32x32 tile (n-tile 2 wn + ni, m-tile 2 wm + mi) of the 64x64 wave tile.
#pragma once
#include "common.h"
#include "kernels.h"

namespace mellow {

__device__ __forceinline__ void store4_bf16(__bf16* dst, const float (&v)[4]) {
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    bf16x4_ h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = static_cast<__bf16>(v[j]);
    *reinterpret_cast<bf16x4_*>(dst) = h;
}

// C16: EPI_LINEAR stores C as bf16 (GemmArgs::c16).  AMX: a C3 output is an AMX image (GemmArgs::c3_fmt == 1: the fp8 kernel), else an APB
// one (the f32x3 kernels).  Compile-time forms: each GEMM family carries only its own hand-over code and registers.
template <int WN, int EPI, bool C16 = false, bool AMX = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], int pm, int pn, int wm, int wn,
                                              int lane, int BM, int BN, const float* rs_rows = nullptr) {
    // rs_rows (optional): the row scales of this workgroup's BM rows, computed by the kernel BEFORE its main loop (LDS)
    // ---- epilogue: lane owns row m_local = lane&31 of each m-tile, columns 8g + 4h + (0..3) ----------
    const int h = lane >> 5;
    // Everything the epilogue READS from memory is fetched up front, in one round of latency.  Left inside the store loops the
    // loads are serialised by the compiler behind the stores they might alias (the residual IS the output buffer of the in-place
    // GEMMs): sixteen load -> add -> store rounds per lane, 15-20 us of a 60 us o_proj launch (round 4).
    float4 pre_a[2][2][4], pre_b[2][4];         // residual (or rotary cos / sin) per output quad; bias per column quad
    int64_t pre_crow[2];
    if constexpr (EPI == EPI_LINEAR) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = pm * BM + wm * 64 + mi * 32 + (lane & 31);
            pre_crow[mi] = m;
            if (m < g.M && g.crow_map) pre_crow[mi] = (int64_t)(m / g.rows_in) * g.rows_out + g.crow_map[m % g.rows_in];
        }
        if (g.bias) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = pn * BN + wn * 64 + ni * 32 + 8 * gq + 4 * h;
                    pre_b[ni][gq] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (col < g.N) pre_b[ni][gq] = *reinterpret_cast<const float4*>(g.bias + col);
                }
        }
        if (g.resid) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = pm * BM + wm * 64 + mi * 32 + (lane & 31);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int col = pn * BN + wn * 64 + ni * 32 + 8 * gq + 4 * h;
                        pre_a[mi][ni][gq] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (m < g.M && col < g.N) pre_a[mi][ni][gq] = *reinterpret_cast<const float4*>(g.resid + pre_crow[mi] * g.ldr + col);
                    }
            }
        }
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        const int P = pn * WN + wn;
        if (P < g.q_heads + g.kv_heads) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = pm * BM + wm * 64 + mi * 32 + (lane & 31);
                const int t = (m < g.M ? m : g.M - 1) % g.T;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    pre_a[mi][0][gq] = *reinterpret_cast<const float4*>(g.rope_cos + (int64_t)t * 32 + 8 * gq + 4 * h);
                    pre_a[mi][1][gq] = *reinterpret_cast<const float4*>(g.rope_sin + (int64_t)t * 32 + 8 * gq + 4 * h);
                }
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = pm * BM + wm * 64 + mi * 32 + (lane & 31);
        if (m >= g.M) continue;
        if (g.rs_ssq) {
            // RMSNorm folded into this GEMM: the norm weight sits in the weight columns, the row statistic comes from the
            // producer's per-64-column partial sums (fixed order), and (x W'^T) r = (x r) W'^T up to fp32 rounding
            float r;
            if (rs_rows) {
                r = rs_rows[wm * 64 + mi * 32 + (lane & 31)];
            } else {
                const float* sp = g.rs_ssq + (int64_t)m * g.rs_parts;
                float ss = 0.f;
                for (int p = 0; p < g.rs_parts; ++p) ss += sp[p];
                r = 1.0f / sqrtf(ss / g.rs_dim + g.rs_eps);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ni][mi][q] *= r;
        }
        if constexpr (EPI == EPI_LINEAR) {
            const int64_t crow = pre_crow[mi];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = pn * BN + wn * 64 + ni * 32 + 8 * gq + 4 * h;
                    if (col >= g.N) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * gq + j];
                    if (g.bias) {
                        const float4 b = pre_b[ni][gq];
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (g.act == ACT_GELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                    } else if (g.act == ACT_SIGMOID) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = sigmoidf_(v[j]);
                    }
                    if (g.resid) {
                        const float4 r = pre_a[mi][ni][gq];
                        v[0] = r.x + v[0]; v[1] = r.y + v[1]; v[2] = r.z + v[2]; v[3] = r.w + v[3];
                    }
                    if (C16) { if (g.C) store4_bf16(reinterpret_cast<__bf16*>(g.C) + crow * g.ldc + col, v); }
                    else if (g.C) *reinterpret_cast<float4*>(g.C + crow * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);   // (null: only the split copy is wanted)
                    if (g.C3) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ni][mi][4 * gq + j] = v[j];          // kept for the split store below
                    }
                }
            }
            if (g.C3) {
                // the stored row, pre-split for the next x3q GEMM, and its sum of squares over this wave's 64 columns
                const int P = pn * WN + wn;
                if (AMX) {           // fp8 mode: the consumer is gemm_mx8_kernel -- MXFP8 in AMX order (common.h), one scale per
                    if (P * 64 < g.N) {        // 32 columns = per (ni) accumulator tile of the lane pair; N % 32 == 0 on this path
                        float ss = 0.f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            if (P * 64 + ni * 32 >= g.N) continue;
                            float v[16];
#pragma unroll
                            for (int q = 0; q < 16; ++q) { v[q] = acc[ni][mi][q]; ss += v[q] * v[q]; }
                            amx_store_block(reinterpret_cast<i32x4*>(g.C3), g.C3s, m, P * 2 + ni, g.c3_kt64, (g.c3_kt64 + 3) >> 2, v, h);
                        }
                        ss = half_sum(ss);
                        if (h == 0 && g.ssq_out) g.ssq_out[(int64_t)m * g.ssq_parts + P] = ss;
                    }
                } else
                if (!AMX && P * 64 < g.N) {            // wave-uniform (N is a multiple of 64 on this path)
                    float ss = 0.f;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            float X[4], Y[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                X[j] = acc[ni][mi][8 * gp + j];
                                Y[j] = acc[ni][mi][8 * gp + 4 + j];
                                ss += X[j] * X[j] + Y[j] * Y[j];
                            }
                            apb_store_quads(reinterpret_cast<i32x4*>(g.C3), m, P * 8 + ni * 4 + 2 * gp, g.N >> 4, X, Y, h);
                        }
                    ss = half_sum(ss);                    // the two half-waves hold the two column halves of the row
                    if (h == 0 && g.ssq_out) g.ssq_out[(int64_t)m * g.ssq_parts + P] = ss;
                }
            }
        } else {
            // pair epilogues: n-tile 2*wn holds the first half of the pair, 2*wn+1 the second
            const int P = pn * WN + wn;  // 64-column group index
            if constexpr (EPI == EPI_SWIGLU) {
                if (AMX && g.C3) {       // fp8 mode: silu(gate) * up as MXFP8 in AMX order; the wave's 32 columns are one block
                    if (P * 32 >= g.N) continue;
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = __fmul_rn(siluf_(acc[0][mi][q]), acc[1][mi][q]);
                    amx_store_block(reinterpret_cast<i32x4*>(g.C3), g.C3s, m, P, g.c3_kt64, (g.c3_kt64 + 3) >> 2, v, h);
                    continue;
                }
                if (!AMX && g.C3) {       // the consumer is an x3q GEMM: write silu(gate) * up pre-split in APB order (N % 16 == 0)
                    if (P * 32 >= g.N) continue;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        float X[4], Y[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            X[j] = __fmul_rn(siluf_(acc[0][mi][8 * gp + j]), acc[1][mi][8 * gp + j]);
                            Y[j] = __fmul_rn(siluf_(acc[0][mi][8 * gp + 4 + j]), acc[1][mi][8 * gp + 4 + j]);
                        }
                        apb_store_quads(reinterpret_cast<i32x4*>(g.C3), m, P * 4 + 2 * gp, g.N >> 4, X, Y, h);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int i0 = 8 * gq + 4 * h;  // 0..31 within the pair
                float x1[4], x2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { x1[j] = acc[0][mi][4 * gq + j]; x2[j] = acc[1][mi][4 * gq + j]; }
                if constexpr (EPI == EPI_POWER) {
                    // re^2 + im^2 with separate roundings, like torch's real**2 + imag**2
                    const int col = P * 32 + i0;
                    if (col >= g.N) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __fadd_rn(__fmul_rn(x1[j], x1[j]), __fmul_rn(x2[j], x2[j]));
                    *reinterpret_cast<float4*>(g.C + (int64_t)m * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (EPI == EPI_SWIGLU) {
                    const int col = P * 32 + i0;
                    if (col >= g.N) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __fmul_rn(siluf_(x1[j]), x2[j]);
                    *reinterpret_cast<float4*>(g.C + (int64_t)m * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (EPI == EPI_LOGMEL) {
                    // not a pair epilogue: both tiles are plain mel columns
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int col = P * 64 + ni * 32 + i0;
                        if (col >= g.N) continue;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float x = ni == 0 ? x1[j] : x2[j];
                            x = x < 1e-10f ? 1e-10f : x;      // torch.clamp(min=1e-10): NaN stays NaN (fmaxf would drop it)
                            x = __fmul_rn(10.0f, log10f(x));
                            if (g.apply_bn) x = __fadd_rn(__fmul_rn(x, g.bn_alpha[col + j]), g.bn_beta[col + j]);
                            v[j] = x;
                        }
                        *reinterpret_cast<float4*>(g.C + (int64_t)m * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else if constexpr (EPI == EPI_QKV_ROPE) {
                    // P = head slot: [0,q_heads) query heads, then kv_heads key heads, then kv_heads value heads
                    const int b = m / g.T, t = m % g.T;
                    if (P < g.q_heads + g.kv_heads) {
                        const float4 c4 = pre_a[mi][0][gq], s4 = pre_a[mi][1][gq];        // (t, i0 = 8 gq + 4 h: fetched up front)
                        const float c[4] = {c4.x, c4.y, c4.z, c4.w};
                        const float sn[4] = {s4.x, s4.y, s4.z, s4.w};
                        float o1[4], o2[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // q*cos + rotate_half(q)*sin, products rounded separately (HF apply_rotary_pos_emb)
                            o1[j] = __fadd_rn(__fmul_rn(x1[j], c[j]), __fmul_rn(-x2[j], sn[j]));
                            o2[j] = __fadd_rn(__fmul_rn(x2[j], c[j]), __fmul_rn(x1[j], sn[j]));
                        }
                        if (g.kv16) {      // fp8 mode: the key goes straight into the bf16 page, the query into a bf16 row (rounded once, RNE)
                            __bf16* d16 = P < g.q_heads ? reinterpret_cast<__bf16*>(g.q_out) + (int64_t)m * (g.q_heads * 64) + P * 64
                                                        : reinterpret_cast<__bf16*>(g.k_cache) + (((int64_t)b * g.kv_heads + (P - g.q_heads)) * g.Tmax + t) * 64;
                            store4_bf16(d16 + i0, o1);
                            store4_bf16(d16 + 32 + i0, o2);
                        } else {
                        float* dst;
                        if (P < g.q_heads) dst = g.q_out + (int64_t)m * (g.q_heads * 64) + P * 64;
                        else dst = g.k_cache + (((int64_t)b * g.kv_heads + (P - g.q_heads)) * g.Tmax + t) * 64;
                        *reinterpret_cast<float4*>(dst + i0) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                        *reinterpret_cast<float4*>(dst + 32 + i0) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                        }
                    } else if (P < g.q_heads + 2 * g.kv_heads) {
                        if (g.kv16) {
                            __bf16* d16 = reinterpret_cast<__bf16*>(g.v_cache) + (((int64_t)b * g.kv_heads + (P - g.q_heads - g.kv_heads)) * g.Tmax + t) * 64;
                            store4_bf16(d16 + i0, x1);
                            store4_bf16(d16 + 32 + i0, x2);
                        } else {
                        float* dst = g.v_cache + (((int64_t)b * g.kv_heads + (P - g.q_heads - g.kv_heads)) * g.Tmax + t) * 64;
                        *reinterpret_cast<float4*>(dst + i0) = make_float4(x1[0], x1[1], x1[2], x1[3]);
                        *reinterpret_cast<float4*>(dst + 32 + i0) = make_float4(x2[0], x2[1], x2[2], x2[3]);
                        }
                    }
                }
            }
        }
    }
}

}  // namespace mellow
