// Host-side launcher declarations for every HIP kernel of the engine.  Each launcher only enqueues
// on the given stream; none of them synchronises or allocates (hipGraph-capture safe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mellow {

// ----------------------------------------------------------------------------------------------------
// Packed weight layout ("P-layout"): a row-major Linear weight W[N][K] is stored as
//   P[nt][k8][lane][4]   nt = n/32, k8 = k/8, lane = (n%32) + 32*((k%8)/4), j = k%4
// i.e. one 1 KiB chunk per (32 rows x 8 k) tile holding, for each of the 64 lanes, the float4 that is
// directly the A operand of four v_mfma_f32_32x32x2_f32 (k-pairs (k0+j, k0+4+j), j=0..3).
// N is zero-padded to a multiple of 128 rows, K to a multiple of 32.
// ----------------------------------------------------------------------------------------------------
void launch_pack_weight(const float* w, int N, int K, int64_t ldw, float* out, int NP, int KP, hipStream_t s);
// pairs-interleaved: packed row 64*j + i       = w0[32*j + i]  (i < 32)
//                    packed row 64*j + 32 + i  = w1[32*j + i]
// (used for re/im of the DFT and gate/up of the SwiGLU so a wave holds both halves of a pair)
void launch_pack_weight_pairs(const float* w0, const float* w1, int N, int K, int64_t ldw, float* out, int NP,
                              int KP, hipStream_t s);

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SIGMOID = 2 };
enum { EPI_LINEAR = 0, EPI_POWER = 1, EPI_LOGMEL = 2, EPI_SWIGLU = 3, EPI_QKV_ROPE = 4 };
enum { A_PLAIN = 0, A_FRAMES = 1 };

struct GemmArgs {
    // C[M][N] (+epilogue) = A[M][K] * W[Nw][K]^T, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32
    const float* A = nullptr;
    int64_t lda = 0;
    int M = 0;
    int K = 0;          // multiple of 32; A must be readable for K columns (pad columns may hold anything
                        // finite when the matching packed weight columns are zero)
    int a_mode = A_PLAIN;
    int fpc = 1;               // A_FRAMES: rows per clip
    int64_t clip_stride = 0;   // A_FRAMES: floats between clips
    int hop = 0;               // A_FRAMES: floats between consecutive rows of a clip
    const float* Wp = nullptr;  // packed weight
    int Nw = 0;                 // logical weight rows (before 128-padding)
    int N = 0;                  // output columns that may be stored (multiple of 4)
    float* C = nullptr;
    int64_t ldc = 0;
    const float* bias = nullptr;   // length >= roundup(Nw,128) or null
    const float* resid = nullptr;  // optional, added after bias/activation; row mapping of C
    int64_t ldr = 0;
    const int32_t* crow_map = nullptr;  // C row = (m / rows_in) * rows_out + crow_map[m % rows_in]
    int rows_in = 1, rows_out = 1;
    int act = ACT_NONE;
    int epi = EPI_LINEAR;
    // EPI_LOGMEL
    const float* bn_alpha = nullptr;
    const float* bn_beta = nullptr;
    int apply_bn = 0;
    // EPI_QKV_ROPE (rows m = b*T + t)
    float* q_out = nullptr;     // [M][q_heads*64]
    float* k_cache = nullptr;   // [B][kv_heads][Tmax][64] of this layer
    float* v_cache = nullptr;
    const float* rope_cos = nullptr;  // [max_pos][32]
    const float* rope_sin = nullptr;
    int T = 1, Tmax = 1, q_heads = 9, kv_heads = 3;
};
void launch_gemm(const GemmArgs& a, hipStream_t s);
double gemm_flops(const GemmArgs& a);

// ---- skinny (decode) split-K GEMM: P[kc][32*RB][N] = X[32*RB][K-slice kc] * W^T -----------------------------
enum { PRO_PLAIN = 0, PRO_SWIGLU = 2 };
// split-K factors of the decode GEMMs (compile-time: slab sums are fully unrolled): K/8 = KC * 4 waves * KPW
constexpr int SK_KC_QKV = 9, SK_KC_O = 9, SK_KC_GU = 3, SK_KC_DOWN = 16;
struct SkinnyArgs {
    const float* X = nullptr;   // PRO_PLAIN: row-major [rows][ldx]; PRO_SWIGLU: kc_in slabs [kc_in][slab_rows][ldx]
    int64_t ldx = 0;
    int K = 0;                  // logical K; K/8 must equal kc_out * 4 * {2,3,6,9,18}
    const float* Wp = nullptr;
    int K8p = 0;                // packed k8 stride = roundup(K,32)/8
    int N = 0;                  // logical outputs (n-tiles = ceil(N/32)); multiple of 4
    float* Y = nullptr;         // partial slabs [kc_out][slab_rows_out][ldy] (may be null when only candidates wanted)
    int64_t ldy = 0;
    int RB = 1;                 // row blocks of 32
    int pro = PRO_PLAIN;
    int kc_in = 1;              // PRO_SWIGLU: input slabs to sum
    int slab_rows = 0;          // rows per input slab
    int kc_out = 1;             // split-K factor = output slabs
    int slab_rows_out = 0;      // rows per output slab
    float* cand_val = nullptr;  // optional fused arg-max candidates [rows][n-tiles] (kc_out must be 1)
    int32_t* cand_idx = nullptr;
};
void launch_skinny(const SkinnyArgs& a, hipStream_t s);
// x_out[r] = x_in[r] + sum_{s<kc} P[s][r]  (x_out may be null);  xn[r] = norm_w * (x_out[r] * rsqrt(mean^2+eps)) if norm_w
// inc_word (may be null): a device int32 incremented by one (the decode position word, see engine.cpp)
void launch_rows_finish(const float* x_in, const float* P, int kc, int64_t slab_stride, float* x_out,
                        const float* norm_w, float eps, float* xn, int rows, int C, int32_t* inc_word, hipStream_t s);

// ---- front-end ---------------------------------------------------------------------------------------
void launch_reflect_pad(const float* wav, int n_clips, int64_t n_samples, float* out, int64_t padded_len, int pad,
                        hipStream_t s);
// logmel_bn [n_src][src_frames][64] -> x0 [n_virtual][4096][96]; virtual clip v = src*n_crops + crop reads
// frames [crop*crop_hop, +crop_len) resampled (bicubic, align_corners) to 1024
void launch_fold_patch_embed(const float* logmel_bn, int n_src, int src_frames, int n_crops, int crop_hop,
                             int crop_len, const float* conv_w, const float* conv_b, const float* ln_w,
                             const float* ln_b, float* x0, hipStream_t s);

// ---- normalisation -----------------------------------------------------------------------------------
// out[m] = LN(in[src(m)]) over C (eps 1e-5).  row_map (may be null): src = (m/ntok)*ntok + row_map[m%ntok]
void launch_layernorm(const float* in, float* out, int M, int C, const float* w, const float* b,
                      const int32_t* row_map, int ntok, hipStream_t s);
// patch merging gather + LN(4C): in [n][R*R][C] -> out [n][(R/2)^2][4C]
void launch_merge_layernorm(const float* in, float* out, int n, int R, int C, const float* w, const float* b,
                            hipStream_t s);
void launch_rmsnorm(const float* in, float* out, int M, int C, const float* w, float eps, hipStream_t s);

// ---- Swin window attention -----------------------------------------------------------------------------
// qkv [M][3C] rows in window order; out [M][C] window order.  bias_exp [nH][64][64]; mask [nW][64][64] or null
void launch_window_attention(const float* qkv, float* out, int M, int C, int nH, const float* bias_exp,
                             const float* mask, int nW, hipStream_t s);

// ---- encoder tail -------------------------------------------------------------------------------------
// y [n][64][768] (post final LN) -> latent [n][768] written into emb rows (n*emb_rows_stride) and im2col
// A_ts [n*32][4608] with k = (cf*3+dt)*768 + ch
void launch_tail_latent_im2col(const float* y, int n, float* latent, int64_t latent_stride, float* a_ts,
                               hipStream_t s);
// mean over crops: in [n][n_crops][len] -> out [n][out_stride...] (out row stride given)
void launch_crop_average(const float* in, int n, int n_crops, int64_t len, int64_t in_stride, float* out,
                         int64_t out_stride, hipStream_t s);
void launch_gelu(const float* in, float* out, int64_t n, hipStream_t s);
// prefix [B][389][576] from proj33 [2B][33][576] (clips 0..B-1 = audio1, B..2B-1 = audio2)
void launch_prefix_assemble(const float* proj33, const float* embed, const int32_t* ids, int B, int text_len,
                            int sep_id, float* prefix, hipStream_t s);
// audio129 [n][129][576] from proj33 [n][33][576] (tap / mellow_encode)
void launch_downsample33(const float* proj33, int n, float* out, hipStream_t s);

// ---- LM attention --------------------------------------------------------------------------------------
// causal GQA flash attention over the KV pages written by the QKV epilogue.  q [B*T][576]; o [B*T][576]
void launch_prefill_attention(const float* q, const float* k_cache, const float* v_cache, float* o, int B, int T,
                              int Tmax, hipStream_t s);
// one decode step: qkv split-K slabs [kc][rows][960] (q|k|v, no RoPE yet) at position *d_pos; sums the slabs,
// applies RoPE, appends K/V to the pages, attends over pos+1 keys.  o [32*RB][576]
void launch_decode_attention(const float* qkv_parts, int kc, int64_t slab_stride, float* k_cache, float* v_cache,
                             const float* rope_cos, const float* rope_sin, const int32_t* d_pos, float* o, int B,
                             int Tmax, hipStream_t s);

// ---- sampling / bookkeeping ------------------------------------------------------------------------------
void launch_argmax(const float* logits, int B, int V, int64_t ld, int32_t* tokens, hipStream_t s);
// per-row arg-max over the lm_head's per-tile candidates; optionally (out_tokens != null) records the token at
// column (*d_pos - T0 + 1) and tracks stop ids; optionally (x != null) gathers embed[token] into x[b]
void launch_argmax_cand(const float* cand_val, const int32_t* cand_idx, int B, int n_tiles, int32_t* tokens,
                        const float* embed, int H, float* x, int32_t* out_tokens, int max_len, const int32_t* d_pos,
                        int T0, int stop_id, int32_t* seen_stop, int32_t* n_seen, hipStream_t s);
// x[b] = embed[tokens[b]] (x may be null); records tokens[b] into out_tokens[b][*d_step] (out_tokens may be null),
// updates seen_stop / n_seen, then advances *d_step (when recording)
void launch_embed_and_record(const float* embed, const int32_t* tokens, int B, int H, float* x, int32_t* out_tokens,
                             int max_len, int32_t* d_step, int stop_id, int32_t* seen_stop, int32_t* n_seen,
                             hipStream_t s);
void launch_advance(int32_t* p, hipStream_t s);
void launch_gather_rows(const float* in, int64_t ld_in, const int32_t* rows, int n, int C, float* out, int64_t ld_out,
                        hipStream_t s);
// rows b*T + (T-1) of x -> out[b]
void launch_take_last(const float* x, int B, int T, int C, float* out, hipStream_t s);

}  // namespace mellow
