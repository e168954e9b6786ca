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
    int c16 = 0;                // EPI_LINEAR: C is a bf16 matrix (ldc in elements): the output is stored rounded once (fp8 mode: q/k/v of the Swin blocks)
    int kv16 = 0;               // fp8 mode: k_cache / v_cache are bf16 pages and q_out bf16 rows (same element order, 2-byte elements): stored rounded once
    // EPI_SWIGLU: when set, the output is written pre-split in APB order (common.h) for an x3q consumer instead of to C.
    // EPI_LINEAR (round 4, norm-free chaining of the f32x3 LM prefill): when set, the stored value (accumulator + bias +
    // residual) is written to C AND pre-split to C3, and ssq_out[m * ssq_parts + P] receives its sum of squares over the
    // 64-column group P = column / 64 -- the RMS statistic of the NEXT normalisation, which the consuming GEMM applies as a row
    // scale of its accumulators (rs_*), the norm weight being folded into that GEMM's weight columns at load time.
    void* C3 = nullptr;
    // split-K (f32x3 kernels, launches of <= 256 output tiles): workspace of up to 512 fp32 partial tiles of 128 x 128; null =
    // never split.  The split count depends on (M, N, K) only; a second launch sums the partials in split order.
    float* sk_ws = nullptr;
    int sk_tiles = 0, sk_max = 0;
    bool no_x3w = false;        // engine option "x3w" = 0: K = 96 launches stay on the LDS-DMA tile kernel (developer A/B, bit-identical)
    float* ssq_out = nullptr;
    int ssq_parts = 0;
    // consumer side: every accumulator of row m is multiplied by 1 / sqrt(sum_p rs_ssq[m * rs_parts + p] / rs_dim + rs_eps)
    // before the epilogue arithmetic (RoPE / SwiGLU / store)
    const float* rs_ssq = nullptr;
    int rs_parts = 0;
    float rs_dim = 1.f, rs_eps = 0.f;
    // fp8 mode (gemm_fp8.hip): activation as an AMX image (MXFP8: e4m3 + one E8M0 scale per 32 k, common.h), weight as a WMX image
    // with one fp32 scale per packed row; fp32 accumulate, C = (A8 . W8^T) * w_scale[n] (+epilogue).  (A8 / W8 / lda8 also carry
    // the pre-split operands of the f32x3 kernels.)
    const uint8_t* A8 = nullptr;      // AMX data, roundup(M, 128) rows x lda8 bytes
    int64_t lda8 = 0;                 // K rounded up to 64
    const uint32_t* a_sc = nullptr;   // AMX scale words
    const uint8_t* W8 = nullptr;      // WMX weight
    const float* w_scale = nullptr;   // [roundup(Nw,128)] per packed weight row
    // format of C3 (the output handed over in the NEXT GEMM's operand order): 0 = APB (bf16 x 3, f32x3 mode), 1 = AMX (fp8 mode,
    // with C3s = its scale bytes and c3_kt64 = the consumer's k64 steps)
    int c3_fmt = 0;
    uint8_t* C3s = nullptr;
    int c3_kt64 = 0;
};
void launch_gemm(const GemmArgs& a, hipStream_t s);
// ---- fp8 (e4m3) GEMM path: BASELINE config 5 (gemm_fp8.hip) -------------------------------------------------
// fp32 row-major activation -> AMX image (roundup(M, 128) x roundup(K, 64) bytes) + scale words (roundup(M, 128) x ceil(K / 256) x 8
// bytes): the standalone form, for GEMM inputs whose producer does not emit AMX itself.  K % 4 == 0.
void launch_quant_mx8(const float* A, int64_t lda, int M, int K, void* img, void* sc, hipStream_t s);
// from an fp32 P-layout weight (NP x KP, already padded / pair-interleaved) to a WMX image (NP x roundup(KP, 64) bytes) + row scales
void launch_pack_fp8(const float* Wp, int NP, int KP, uint8_t* W8, float* w_scale, hipStream_t s);
void launch_gemm_fp8(const GemmArgs& a, hipStream_t s);   // a_mode == A_PLAIN
// ---- fp32 GEMM on the bf16 matrix pipe by exact 3-way operand splitting (gemm_bf16x3.hip) ----------------------
// operands travel in GemmArgs::A8 (A3 [M][K/8][3][8 bf16], lda8 = 16-byte units per row = 3 K/8) and GemmArgs::W8 (PB)
void launch_pack_bf16x3(const float* Wp, int NP, int KP, void* PB, hipStream_t s);
void launch_gemm_bf16x3_fused(const GemmArgs& a, hipStream_t s);
// pre-split A in fragment order ("APB", A8) + PB weight (W8), both staged by LDS-DMA
void launch_split_rows_apb(const float* A, int64_t lda, int M, int K, void* out, hipStream_t s);
void launch_gemm_bf16x3_apb(const GemmArgs& a, hipStream_t s);   // A split in registers (no pre-pass), K % 32 == 0, a_mode == A_PLAIN
double gemm_flops(const GemmArgs& a);

// Allow `bytes` of dynamic LDS (> 64 KiB) for a kernel function on the CURRENT device: remembered per (function, device) under a
// lock (an engine per GPU, pools of host threads), return code reported on stderr.  No-op up to 64 KiB.
void set_max_dynamic_lds(const void* fn, size_t bytes);

// ---- decode step (decode.hip) --------------------------------------------------------------------------------
// split factors are compile-time so that partial-sum ("slab") loads are fully unrolled and issued together
constexpr int DEC_KC_QKV = 8;    // qkv: 72 k-tiles = 8 chunks x 3 waves x 3  (30 x 8 = 240 workgroups <= 256 CUs)
#ifndef MELLOW_DEC_KC_DOWN
#define MELLOW_DEC_KC_DOWN 8
#endif
constexpr int DEC_KC_DOWN = MELLOW_DEC_KC_DOWN;   // down: 192 k-tiles = 8 chunks x 4 waves x 6
#ifndef MELLOW_DEC_TS
#define MELLOW_DEC_TS 2
#endif
constexpr int DEC_TS = MELLOW_DEC_TS;   // key splits of the decode attention at ONE row block (merged by the o_proj prologue); the buffers are sized for it
// Batches of two or more row blocks run ONE split in the default mode (DecArgs::ts): the launch already has >= 192 workgroups, a
// second split only repeats the prologue and doubles the partials the o_proj ingests (same box, decode per 63 steps at B = 64:
// 63.9 -> 60.6 ms; tokens unchanged).  Only in the f32x3 mode, whose last bits depend on the batch size anyway (mellow_hip.h, ABI
// minor 1): the exact-fp32 mode keeps its promise of batch-size-independent arithmetic, and in the fp8 mode a row's TOKENS must not
// depend on the batch around it (an e4m3 rounding downstream turns the 1e-7 of another summation order into another token:
// test_fp8_mode_end_to_end) -- there the split count is DEC_TS for every batch size.
#ifndef MELLOW_DEC_TS_MULTI
#define MELLOW_DEC_TS_MULTI 1
#endif
constexpr int DEC_TS_MULTI = MELLOW_DEC_TS_MULTI;
inline int dec_key_splits(int RB, bool fixed) { return RB <= 1 || fixed ? DEC_TS : DEC_TS_MULTI; }
// dec_qkv2_kernel (down projection of layer l + q/k/v of layer l+1 in one launch): k-chunks of the h part, waves, derived counts
#ifndef MELLOW_Q2_HC
#define MELLOW_Q2_HC 4
#endif
#ifndef MELLOW_Q2_WAVES
#define MELLOW_Q2_WAVES 12
#else
#define MELLOW_Q2_WAVES_FORCED 1
#endif
constexpr int Q2_HC = MELLOW_Q2_HC, Q2_WAVES = MELLOW_Q2_WAVES;
constexpr int Q2_NPQ = 2 + Q2_HC;             // qkv slabs the fused kernel emits: 2 chunks of x_mid + Q2_HC chunks of h
constexpr int Q2_K8 = 72 + 192;               // k-tiles of a Wq2 row: [W' (576 columns) | W' Wd (1536 columns)]
constexpr int Q2_BLOCKS = 60 + 48 * Q2_HC;    // workgroups per row block
static_assert(Q2_NPQ <= DEC_KC_QKV && Q2_HC <= DEC_KC_DOWN, "the fused kernel reuses the slab buffers of the split-K kernels");
// engine option "decode_x3": bit 1 the lm_head on the streaming f32x3 kernel; bits 2 AND 4 together the layer launches (gate/up and the
// fused down + q/k/v exchange pre-split activations, so they switch as a pair: engine_lm.cpp ensure_lm)
enum { DEC_X3_HEAD = 1, DEC_X3_GATEUP = 2, DEC_X3_QKV = 4, DEC_X3_ALL = 7 };
struct DecArgs {
    int rows = 0, RB = 0;          // padded batch rows (multiple of 32), row blocks
    int Tmax = 0;
    float eps = 1e-5f;
    int32_t* d_pos = nullptr;      // device position word: index of the token being processed
    int first = 0;                 // set on the first qkv launch of a step: stage rope_cur (and, with inc_pos, advance d_pos)
    int inc_pos = 0;
    int gs = 1;                    // attention: 4-key groups per key split (fixed per launch; last split takes the rest)
    int ts = DEC_TS;               // key splits of this batch's attention launches: dec_key_splits(RB)
    float* rope_cur = nullptr;     // [64] cos | sin of the current position
    float* ssq1 = nullptr;         // [rows][DEC_KC_QKV] per-k-chunk sums of squares of x_new (qkv kernel -> attention)
    const float* rope_cos = nullptr;
    const float* rope_sin = nullptr;
    // residual stream: "mid" = layer input base (embedding rows / previous layer's x_mid), F32-layout
    float* xmidF = nullptr;
    float* xnewR = nullptr;        // x after the previous down projection, row-major, materialised by the qkv kernel
    float* dslabF = nullptr;       // down split-K slabs [DEC_KC_DOWN] x F32-layout
    int64_t slabF_stride4 = 0;     // float4 elements between F-layout slabs
    float* pq = nullptr;           // qkv split-K slabs [DEC_KC_QKV][rows][960]
    float* attF16 = nullptr;       // attention partial outputs [ts][RB][36][2][64][4] (F16-layout)
    float* att_ml = nullptr;       // (running max, sum of weights) of every key split: [9 heads][rows][ts][2]
    float* ssq = nullptr;          // [rows][40] per-o_proj-tile sums of squares of x_mid
    float* xmidF16 = nullptr;      // x_mid in F16-layout (gate/up operand)
    float* guF = nullptr;          // h = SwiGLU(gate, up) [RB][192][64][4] (F32-layout B operand of the down projection)
    float* xnF = nullptr;          // final-normed x, F32-layout (lm_head operand)
    // f32x3 layer kernels (row blocks >= the engine's threshold): x_mid pre-split by the o_proj for the fused down + q/k/v launch
    // (F3-32) and for gate/up (F3-16, replaces xmidF16), h pre-split by gate/up (F3-32, 96 pairs); null = the fp32 kernels
    void *xmid3_32 = nullptr, *xmid3_16 = nullptr, *h3 = nullptr;
    void* xn3 = nullptr;           // f32x3 mode: the same pre-split (F3-32, decode.hip) instead of xnF; null = fp32 xnF
    // per-row-block early exit (reference stop rule, batches of more than one 32-row block): blk_live[rb] == 0 once every
    // row of block rb has produced the stop id -- its workgroups return at once.  Null = never skip (one block / fixed length).
    const int32_t* blk_live = nullptr;
    // copy of blk_live taken by the final-norm launch of the step: what the step's arg-max tests, so that the rows of a block
    // whose last row stops in THIS arg-max launch all still record this step's token, whatever order its workgroups run in
    int32_t* blk_snap = nullptr;
    // row migration (same mode): row_of_slot[s] = the example whose state lives in batch slot s, or -1 for an empty slot.  The
    // slot addresses everything a step computes (activations, slabs); the example addresses what persists (its KV pages, its
    // token record, its stop flag).  dec_compact_kernel repacks the rows that are still running into the lowest slots
    // whenever that empties a whole 32-row block.
    const int32_t* row_of_slot = nullptr;
    int kv16 = 0;                  // fp8 mode: the K/V page pointers of launch_dec_attn address the bf16 shadow pages (decode.hip, KV16)
    int a8 = 0;                    // fp8 mode: launches that get e4m3 weights also quantise their activations (fp8 matrix pipe)
    int x3 = 0;                    // f32x3 mode: bit mask of the decode GEMM launches that run on the bf16 matrix pipe with operands split in
                                   // registers (DEC_X3_*; decode.hip); 0 = the exact fp32 MFMA kernels
    int dbg_seq = -1;              // -DMELLOW_KDEBUG builds: index of this launch in the step (span stamps of tools/kdebug.py), else unused
    float* logits = nullptr;       // [rows][vocab] (may be null)
    float* cand_val = nullptr; int32_t* cand_idx = nullptr;   // [rows][vocab/32]
};
// The five GEMM launchers take an optional `wscale`: when non-null, the weight pointer addresses the e4m3 copy of the same
// layout (launch_pack_dec_fp8: one 4-byte word per float4 slot) and wscale holds one factor per packed weight row
void launch_dec_qkv(const DecArgs& a, const float* Wp_folded, int K8p, int kcd, hipStream_t s, const float* wscale = nullptr);
void launch_pack_dec_fp8(const float* Wp, int tiles, int slots_per_tile, int rows_per_tile, void* out, float* scale, hipStream_t s);
// fused = the projected values come from launch_dec_qkv2 (Q2_NPQ slabs; the attention also forms x_new from the down slabs)
void launch_dec_attn(const DecArgs& a, float* k_cache, float* v_cache, bool fused, hipStream_t s);
// n fp32 values (n % 8 == 0) -> bf16, round to nearest even (the K/V shadow pages of the fp8 mode)
void launch_kv_to_bf16(const float* src, void* dst, int64_t n, hipStream_t s);
// down projection of a layer + q/k/v projection of the next one: Wq2 = P-layout [30][Q2_K8] of [W'_{l+1} | W'_{l+1} Wd_l],
// Wd = this layer's down weight in P-layout (K8p = 192)
void launch_dec_qkv2(const DecArgs& a, const float* Wq2, const float* Wd, hipStream_t s);
// f32x3 forms for any number of row blocks (operands a.xmid3_32 / a.h3 / a.xmid3_16 pre-split by the o_proj / gate-up launches)
void launch_dec_qkv2x3(const DecArgs& a, const float* Wq2, const float* Wd, hipStream_t s);
void dec_prepare_lds_attributes();      // once per device, outside any stream capture: dynamic-LDS limits of the kernels above
void launch_dec_gateup3(const DecArgs& a, const float* Wp16n_folded_pairs, hipStream_t s);
// the same launch on e4m3 weights: the unfused layer's q/k/v copy (72 k-tiles per n-tile), the composed W' Wd (192) and the
// down copy, one scale per packed row each (launch_pack_dec_fp8)
void launch_dec_qkv2_w8(const DecArgs& a, const float* Wx8, const float* sc_x, const float* Wh8, const float* sc_h,
                        const float* Wd8, const float* sc_d, hipStream_t s);
// C[M][N] = A[M][K] . B[K][N], fp64 accumulate, rounded once to fp32 (row-major device buffers)
void launch_compose_f64(const float* A, const float* B, float* C, int M, int N, int K, hipStream_t s);
int dec_attn_chunk_groups(bool kv16);   // 4-key groups one attention workgroup covers per pass
void launch_dec_oproj(const DecArgs& a, const float* Wp16, hipStream_t s, const float* wscale = nullptr);
void launch_dec_gateup(const DecArgs& a, const float* Wp16_folded_pairs, hipStream_t s, const float* wscale = nullptr);
void launch_dec_down(const DecArgs& a, const float* Wp, int K8p, hipStream_t s, const float* wscale = nullptr);
void launch_dec_final_norm(const DecArgs& a, const float* norm_w, int kcd, hipStream_t s);
bool dec_head3r_fits(int vocab);     // the streaming f32x3 lm_head tiles this vocabulary (else xn3 must stay null: fp32 xnF + the fp32 kernel)
void launch_dec_lm_head(const DecArgs& a, const float* Wp, int K8p, int vocab, hipStream_t s, const float* wscale = nullptr);
// Generation-loop bookkeeping that lives on the device (reference wrapper.py:232-249), written by the arg-max kernel:
// the token is recorded at column (*d_pos - T0 + 1) of out_tokens, rows that produced the stop id are counted once, and
// the LAST row to arrive publishes (step ticket << 32 | rows that have stopped) to a host-visible word, so the host
// loop follows the stop rule without ever synchronising the stream.
struct LoopArgs {
    int32_t* out_tokens = nullptr;       // engine-owned [rows][params[0]] (null: taps / single-step calls record nothing)
    const int32_t* params = nullptr;     // device {max_len, stop_id}: graph replays serve any value
    int32_t* seen_stop = nullptr;        // [rows] 0/1
    int32_t* n_seen = nullptr;
    int32_t* arrive = nullptr;           // rows that finished this step's arg-max (reset by the last one)
    int32_t* ticket = nullptr;           // arg-max launches since the start of the call
    int32_t* blk_left = nullptr;         // [row blocks] rows of the block that have not produced the stop id yet
    int32_t* blk_live = nullptr;         // [row blocks] cleared by the row that brings blk_left to 0 (DecArgs::blk_live)
    const int32_t* blk_snap = nullptr;   // [row blocks] blk_live as it was before this step's arg-max (DecArgs::blk_snap)
    int32_t* row_of_slot = nullptr;      // [rows] example of each batch slot (DecArgs::row_of_slot); null: slot == example
    int32_t* n_compactions = nullptr;    // repacks done during the call (diagnostic)
    unsigned long long* host_progress = nullptr;   // mapped host memory
    int T0 = 0;                          // prefix length
};
// per-row arg-max over the lm_head candidates (torch.argmax order: NaN = maximum, lowest index on ties); with
// write_x it gathers embed[token] as the next step's residual stream
void launch_dec_argmax(const DecArgs& a, int B, int n_tiles, int32_t* tokens, const float* embed, int write_x,
                       const LoopArgs& loop, hipStream_t s);
// after the arg-max of a step (early-exit mode only): if the rows that have not produced the stop id yet fit into fewer 32-row
// blocks than are live, move them (their next-step residual rows) to the lowest slots, rewrite row_of_slot / blk_left / blk_live
void launch_dec_compact(const DecArgs& a, int B, const LoopArgs& loop, hipStream_t s);
// residual stream <- rows of `in`: row_ids[b] (embedding gather) or, when row_ids == null, row b*T_last + T_last-1
void launch_dec_load_rows(const DecArgs& a, int B, const float* in, int64_t ld, const int32_t* row_ids, int T_last,
                          int n_src /* rows of `in` that row_ids may address */, hipStream_t s);
void launch_pack_weight16(const float* w, int N, int K, float* out, hipStream_t s);
// the same 16-row tiles with eight consecutive k per lane (f32x3 gate/up, decode.hip P16N); K % 32 == 0
void launch_pack_weight16n(const float* w, int N, int K, float* out, hipStream_t s);
// developer instrumentation: device buffer of 64 uint64 slots stamped by workgroup 0 of the decode kernels (null = off)
void set_kernel_debug_buffer(uint64_t* p);
void set_gemm_debug_buffer(uint64_t* p);

// ---- front-end ---------------------------------------------------------------------------------------
// device twin of the host resampler (A0): wT = transposed polyphase bank [klen][nw]
void launch_resample(const float* x, int n_clips, int64_t n_in, const float* wT, int orig, int nw, int klen, int width,
                     float* out, int64_t n_out, hipStream_t s);
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
// A1 as a 1024-point FFT per frame (stft_fft.hip): power[m][0..512] = |DFT(win * frame m)|^2, columns 513..543 zero.  win [1024];
// tw1 [16][64] complex = exp(-2 pi i b k1 / 1024), tw2 [4][16] complex = exp(-2 pi i d e / 64)
void launch_stft_fft_power(const float* wpad, int fpc, int64_t clip_stride, int hop, int M, const float* win, const float* tw1,
                           const float* tw2, float* power, hipStream_t s);
void launch_rmsnorm(const float* in, float* out, int M, int C, const float* w, float eps, hipStream_t s);
// the same numbers, written pre-split in APB order (C % 16 == 0) for the x3q GEMM
// row-map LayerNorm, output pre-split in APB order (C % 16 == 0, C <= 768; out holds roundup(M, 128) rows)
// out_scales != nullptr (fp8 mode): out_apb receives the AMX image (MXFP8, common.h) and out_scales its scale bytes instead
void launch_layernorm_apb(const float* in, void* out_apb, int M, int C, const float* w, const float* b, const int32_t* row_map,
                          int ntok, hipStream_t s, void* out_scales = nullptr);
void launch_rmsnorm_apb(const float* in, void* out_apb, int M, int C, const float* w, float eps, hipStream_t s, void* out_scales = nullptr);

// ---- Swin window attention -----------------------------------------------------------------------------
// qkv [M][3C] rows in window order; out [M][C] window order.  bias_exp [nH][64][64]; mask [nW][64][64] or null
void launch_window_attention(const float* qkv, float* out, int M, int C, int nH, const float* bias_exp,
                             const float* mask, int nW, hipStream_t s, void* out_apb = nullptr, bool qkv16 = false);

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
                            int sep_id, int vocab, float* prefix, unsigned long long* bad_id_word, hipStream_t s);
// out[i] = table[ids[i]] (rows of `width` floats, width % 4 == 0; ids clamped to [0, n_rows))
void launch_gather_rows(const float* table, int width, const int32_t* ids, int n, int n_rows, float* out, hipStream_t s);
// out [B][n][576] = in [B][T][576] rows from_pos .. from_pos + n - 1
void launch_gather_span(const float* in, int B, int T, int from_pos, int n, float* out, hipStream_t s);
// zero the token slots [t0, t1) of `pages` KV pages of Tmax x 64 floats each
void launch_clear_page_slots(float* cache, int64_t pages, int Tmax, int t0, int t1, hipStream_t s, bool pages16 = false);
// audio129 [n][129][576] from proj33 [n][33][576] (tap / mellow_encode)
void launch_downsample33(const float* proj33, int n, float* out, hipStream_t s);

// ---- LM attention --------------------------------------------------------------------------------------
// causal GQA flash attention over the KV pages written by the QKV epilogue.  q [B*T][576]; o [B*T][576]
// o_apb != nullptr: the output [B*T][576] is written pre-split in APB order instead of to o
// x3: both matrix products as exact 3-way bf16 splits on v_mfma_f32_32x32x16_bf16 (the f32x3 mode); else exact fp32 MFMA
// bf16_once (fp8 mode): operands rounded once to bf16 instead of split exactly in three (plain bf16 flash attention)
// o_scales != nullptr (fp8 mode): o_apb receives the AMX image (MXFP8, K = 576: 9 k64 steps) and o_scales its scale bytes
void launch_prefill_attention(const float* q, const float* k_cache, const float* v_cache, float* o, void* o_apb, int B, int T,
                              int Tmax, bool x3, hipStream_t s, void* o_scales = nullptr, bool bf16_once = false, bool pages16 = false);
// ---- misc ------------------------------------------------------------------------------------------------------
void launch_argmax(const float* logits, int B, int V, int64_t ld, int32_t* tokens, hipStream_t s);

}  // namespace mellow
