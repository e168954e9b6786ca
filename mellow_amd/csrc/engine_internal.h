// Internal header of the host side of libmellow_hip.so (engine*.cpp): the engine object, its weights, and the helpers the
// translation units share.  Not part of the C ABI (include/mellow_hip.h is).
//   engine.cpp          errors, allocation helpers, create / destroy / fork / load_tensor, precision, profiler read-out
//   engine_weights.cpp  mellow_engine_finalize: every reference checkpoint key -> device layouts (P / PB / P16 / e4m3, composed decode weights)
//   engine_encoder.cpp  GEMM dispatch, front-end + HTSAT encoder (A1-A13), the taps mellow_logmel / mellow_encode / mellow_resample
//   engine_lm.cpp       KV pages, LM prefill (A15), the decode step, mellow_prefix / lm taps, mellow_generate (A16)
//   engine_dev.cpp      developer entry points (GEMM timing / debug taps, kernel stamps)
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mellow_hip.h"
#include "kernels.h"

using namespace mellow;

int fail(const char* fmt, ...);
struct mellow_engine;
int apply_options(mellow_engine* e);            // engine.cpp: option table + precision mode -> the engine's resolved fields
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)            \
    do {                     \
        int _r = (expr);     \
        if (_r) return _r;   \
    } while (0)

// ---- model constants (reference mellow/model/config.py:1-10, htsat.py:599-606) ---------------------------
static const int kDepths[4] = {2, 2, 6, 2};
static const int kHeads[4] = {4, 8, 16, 32};
static const int kWin = 8;
static const int kHop = 320, kNfft = 1024, kNfreq = 513, kMel = 64;
static const int kClasses = 527, kEncOut = 768, kProj = 576;
static const int kLongCrop = 689, kLongHop = 344;
static const char* ENC = "audio_encoder.base.htsat.";
static const char* C2L = "audio_encoder.base.c2l.";
static const char* PRJ = "audio_encoder.projection.";
static const char* LMK = "caption_decoder.lm.";

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ---- profiler families --------------------------------------------------------------------------------------
enum { PF_GEMM = 0, PF_SKINNY, PF_PREFILL_ATTN, PF_DECODE_ATTN, PF_WINDOW_ATTN, PF_NORM, PF_MISC, PF_COUNT };
static const char* kFamilyNames[PF_COUNT] = {"gemm_f32_mfma", "skinny_gemm_m32", "prefill_attention",
                                             "decode_attention", "window_attention", "norm", "misc"};

struct HostTensor {
    std::vector<char> data;
    std::vector<int64_t> shape;
    int dtype = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
    const float* f() const { return reinterpret_cast<const float*>(data.data()); }
};

struct Packed {  // a P-layout weight
    float* p = nullptr;
    int N = 0, K = 0, NP = 0, KP = 0;
    int Nw = 0;  // logical packed rows (pairs: 64*ceil(N/32))
};

struct SwinBlockW {
    float *n1w, *n1b, *n2w, *n2b;
    Packed qkv, proj, fc1, fc2;
    float *qkv_b, *proj_b, *fc1_b, *fc2_b;
    float* bias_exp;  // [nH][64][64]
    float* mask;      // [nW][64][64] or null
};
struct MergeW {
    float *nw, *nb;
    Packed red;
};
struct LMLayerW {
    Packed qkv, o, gateup, down;       // prefill (plain weights, P-layout); `down` is also the decode operand
    Packed qkv_f, gateup_f;            // decode: RMSNorm weight folded into the columns (W'[n][k] = W[n][k]*ln[k])
    float* o16 = nullptr;              // decode: P16 layout (16-row tiles) for the complete-output o_proj
    float* gu16n = nullptr;            // decode, f32x3 layer kernels: the same tiles in P16N order (eight consecutive k per lane)
    float* gu16 = nullptr;             // decode: folded gate/up, P16 layout, tile = 8 gate rows + the 8 matching up rows
    // decode, layers >= 1: [W'_l | W'_l Wd_{l-1}] (960 x (576 + 1536), P-layout, the product formed in fp64 at load time): the
    // operand of dec_qkv2_kernel, which runs the down projection of layer l-1 and the q/k/v projection of layer l as one launch
    float* qkv2 = nullptr;
    float *q2h8 = nullptr, *q2h_sc = nullptr;    // fp8 mode: the composed part W'_l . Wd_{l-1} alone, e4m3 + one scale per packed row
    // fp8 mode: e4m3 copies of the four decode operands in the same slot order (one 4-byte word per float4 slot) and one
    // scale per packed weight row (launch_pack_dec_fp8)
    float *qkv8 = nullptr, *qkv_sc = nullptr, *o8 = nullptr, *o_sc = nullptr, *gu8 = nullptr, *gu_sc = nullptr, *dn8 = nullptr,
          *dn_sc = nullptr;
    float *in_ln, *post_ln;
};

struct ProfRec {
    int fam;
    hipEvent_t a, b;
    double flops, bytes;
    int M = 0, N = 0, K = 0, epi = 0;     // GEMM launches only (developer shape report)
};

struct mellow_engine {
    mellow_config_t cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    // explicit configuration (mellow_engine_set_option, engine.cpp): key -> value as given by the caller.  The library reads NO
    // environment variable; apply_options() turns this table (+ the precision mode) into the fields below, and
    // mellow_engine_describe() reports the resolved values.
    std::map<std::string, std::string> opts;
    int mode = MELLOW_PRECISION_F32X3;
    bool x3_stft = true, stft_fft = true, x3_apb = true, x3_attn = true, x3w = true, row_migration = true, decode_fuse = true;
    int arena_mb = 3400;
    int prefill_parts = 2;                      // parts of the split LM prefill (option "prefill_split")
    bool streams_probed = false;                // a probe found fewer overlapping streams than asked for (ensure_prefill_streams): re-probed after 16 calls
    int probe_backoff = 0, prefill_parts_ok = 1, prefill_parts_want = 2;
    hipStream_t stream2[3] = {nullptr, nullptr, nullptr};      // further streams of the split LM prefill (run_prefill)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool finalized = false;
    bool owns_weights = true;                 // false for a context made by mellow_engine_fork: weight memory belongs to its parent
    std::map<std::string, HostTensor> host;   // until finalize
    std::vector<void*> allocs;                // everything hipMalloc'd for weights
    char* arena = nullptr;                    // one big allocation the weights are carved from
    size_t arena_size = 0, arena_used = 0;

    // encoder weights
    Packed dft, mel;
    // f32x3 mode: the STFT as a real FFT when the checkpoint's conv weights are window[n] * cos / sin(2 pi k n / 1024) (checked
    // element by element at load time); fft_win == nullptr: the DFT GEMM on the checkpoint's weights
    float *fft_win = nullptr, *fft_tw1 = nullptr, *fft_tw2 = nullptr;
    float *bn_alpha = nullptr, *bn_beta = nullptr;
    float *pe_w = nullptr, *pe_b = nullptr, *pe_nw = nullptr, *pe_nb = nullptr;
    std::vector<SwinBlockW> blocks[4];
    MergeW merge[3];
    int32_t* win_map[4][2] = {{nullptr}};     // [stage][shifted]
    float *fn_w = nullptr, *fn_b = nullptr;
    Packed tscam, c2l, lin1, lin2;
    float *tscam_b = nullptr, *c2l_b = nullptr, *pln_w = nullptr, *pln_b = nullptr;
    int32_t* emb_row_map = nullptr;           // {1..32}
    // LM
    float* embed = nullptr;                   // row-major [V][H]
    Packed lm_head;
    std::vector<LMLayerW> layers;
    float* final_norm = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;

    // workspaces (grow-only)
    struct Buf {
        float* p = nullptr;
        size_t cap = 0;
    };
    Buf wavcat, wpad, power, logmel, X0, X1, T, QKV, H, ats, fpx, fpxavg, latv, emb33, e1, gbuf, sbuf, proj33;
    Buf lm_x, lm_xn, lm_q, lm_o, lm_h, kcache, vcache;
    Buf kcache16, vcache16;      // fp8 mode: bf16 shadow of the pages for the decode step (half the floats of kcache / vcache)
    bool kv16_direct = false;    // ... and the prefill writes them itself (q/k/v epilogue) and reads them (bf16-once attention): no fp32 pages, no conversion pass
    bool kv16 = false;           // fp8 mode default; option "fp8_kv16" = 0 keeps the decode step on the fp32 pages
    Buf lm_xn3, lm_o3, lm_h3;                  // f32x3 mode: the GEMM inputs of LM prefill, pre-split by their producers (APB order)
    Buf lm_ssq;                                // ... and the per-row sum-of-squares partials of the residual stream (norm-free chaining)
    Buf dec;                                   // one arena for the decode-step buffers (DecArgs)
    Buf dlogits, cand;
    DecArgs da;
    int32_t *d_tokens = nullptr, *d_step = nullptr, *d_pos = nullptr, *d_seen = nullptr, *d_nseen = nullptr;
    int32_t *d_arrive = nullptr, *d_ticket = nullptr, *d_params = nullptr;   // loop bookkeeping words (LoopArgs)
    int32_t *d_blk_left = nullptr, *d_blk_live = nullptr;                    // per-row-block early exit (32 blocks max)
    int32_t *d_row_of_slot = nullptr, *d_ncompact = nullptr;                 // row migration (kernels.h, DecArgs::row_of_slot)
    int last_compactions = 0;
    const void* graph_blk = nullptr;                                         // DecArgs::blk_live the graphs were captured with
    const void* graph_rows = nullptr;                                        // DecArgs::row_of_slot likewise
    unsigned long long* h_progress = nullptr;  // mapped host word the arg-max kernel publishes (ticket << 32 | rows stopped) to
    unsigned long long* d_progress = nullptr;  // its device alias
    std::map<std::pair<int, int>, float*> resample_banks;   // (orig, new) gcd-reduced -> device polyphase bank [klen][new]
    int32_t h_params[2] = {0, 0};              // staging of d_params {max_len, stop id}
    int32_t h_blk[64] = {0};                   // staging of d_blk_left[32] | d_blk_live[32]
    std::vector<int32_t> h_ident;              // staging of d_row_of_slot
    int last_steps_enqueued = 0;               // decode steps (incl. the prefill's token) the last generate call enqueued
    Buf out_tok;                               // engine-owned token record [rows][max_len] (stable address: graph-safe)
    int kv_B = 0, kv_Tmax = 0;                // current page geometry
    int cur_B = 0, cur_pos = 0;               // host mirror of the decode state
    int32_t h_pos_word = 0;                   // staging for the device position word

    // taps
    bool taps_on = false;
    std::map<std::string, Buf> taps;
    std::map<std::string, int64_t> tap_numel;

    // graph
    bool use_graph = true;
    // fp8 GEMM mode (BASELINE config 5): every packed weight with KP % 64 == 0 also gets a P8 copy + per-row scales,
    // looked up by the fp32 packed pointer when a GEMM is issued; activations are quantised per row right before the GEMM
    bool fp8 = false;
    bool fp8_decode = false;                     // fp8 mode: the decode kernels read e4m3 weights too (option "fp8_decode")
    bool fp8_decode_act = false;                 // ... and quantise their activations: fp8 matrix pipe (option "fp8_decode_act")
    bool fp8_attn_bf16 = true;                   // fp8 mode: prefill attention on operands rounded once to bf16 (option "fp8_attn_bf16" = 0: the exact 3-way split)
    bool fp8_prefill = true;                     // fp8 mode: e4m3 GEMMs in encoder + prefill (option "fp8_prefill" = 0: a test isolating the decode weights)
    float *head8 = nullptr, *head_sc = nullptr;  // e4m3 lm_head for the decode step
    int dec_x3_min_rb = 2;                       // ... and the fewest 32-row blocks at which the layer GEMM launches take their f32x3 forms (option "decode_x3_min_rb")
    int dec_x3 = 0;                              // f32x3 mode: DEC_X3_* mask of the decode GEMM launches on the bf16 pipe (option "decode_x3", developer A/B)
    int f32x3_terms = 0;                         // 0 = off; 6 = fp32 GEMMs on the bf16 pipe by exact 3-way operand splitting (six partial products)
    // f32x3 LM prefill without RMSNorm launches: the o_proj / down GEMMs write their output pre-split + its sum of squares,
    // the q/k/v and gate/up GEMMs run on norm-folded weights and scale their accumulators by the row statistic (run_prefill).
    // option "prefill_fuse_norm" = 0: the two-launch form (developer A/B).
    bool prefill_fuse_norm = true;
    int enc_apb_stages = 0x1CC;                   // f32x3 mode: Swin stages (bit st) whose LayerNorms / fc1 hand their output over pre-split (APB) to x3q GEMMs
    Buf sk_ws;                                   // split-K workspace of the f32x3 GEMMs (512 partial tiles of 128 x 128 fp32);
    bool sk_enable = false;                      // ... only launches of the encoder chain (one stream) may use it: run_encoder switches it on
    int sk_max = 8;                              // ... largest split count (option "splitk"; < 2 = never split)
    Buf enc_a3, enc_h3;                          // ... the two pre-split operands (LayerNorm output; GELU(fc1) output)
    uint64_t dbg_spans[960] = {};
    int dbg_seq0 = -1;                           // developer stamps (mellow_dev_kdebug): first launch index of a decode step, -1 = off
    int dec_fuse_max_rb = 1;                     // decode: largest number of 32-row blocks that runs the fused down + q/k/v launch (fp32 weights)
    mellow_engine* parent = nullptr;             // a fork: the context whose weights it shares
    int n_forks = 0, prefill_parts_saved = 2;    // a parent: live forks; its own split setting, restored when the last fork goes
    bool decode_only_weight = false;             // set while packing weights only the decode kernels read: no bf16x3 / fp8 copy
    std::unordered_map<const float*, void*> bf_w;   // fp32 packed pointer -> PB copy
    struct Fp8W { uint8_t* w8; float* scale; };
    std::unordered_map<const float*, Fp8W> fp8_w;
    Buf a8, a8_scale;     // quantised A operand of the GEMM in flight (bytes / floats, carved from float buffers)
    hipGraphExec_t step_exec = nullptr;
    hipGraphExec_t step_exec8 = nullptr;      // the same step captured 8 times in a row (the step is position-independent)
    int step_exec_B = -1, step_exec_Tmax = -1;
    const void* graph_out_tok = nullptr;      // the graphs bake buffer addresses in; max_len / stop id travel in d_params

    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t ev_phase[4] = {nullptr, nullptr, nullptr, nullptr};
    float phase_ms[3] = {0, 0, 0};
};

// ---- helpers shared by the translation units (engine.cpp unless noted) ------------------------------------------------------
int ensure(mellow_engine* e, mellow_engine::Buf& b, size_t floats);
int dev_alloc(mellow_engine* e, float** out, size_t floats);
int upload(mellow_engine* e, float** out, const float* src, size_t floats, size_t alloc_floats = 0);
hipEvent_t next_event(mellow_engine* e);
int tap(mellow_engine* e, const char* name, const float* src, int64_t n);
std::vector<std::string> build_required(const mellow_config_t* cfg);
const std::vector<std::string>& default_required();
bool is_ignored_key(const std::string& k);
void window_map_host(int R, int shift, int32_t* out);
void pack_weight_host(const float* w, int N, int K, int NP, int KP, float* out);
void rope_tables_host(float theta, int head_dim, int P, float* c, float* s);
int alloc_state_words(mellow_engine* e);
struct ProfScope {
    mellow_engine* e;
    ProfRec r;
    bool on;
    ProfScope(mellow_engine* e_, int fam, double flops, double bytes) : e(e_), on(e_->prof_on) {
        if (on) {
            r.fam = fam;
            r.flops = flops;
            r.bytes = bytes;
            r.a = next_event(e);
            r.b = next_event(e);
            hipEventRecord(r.a, e->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            hipEventRecord(r.b, e->stream);
            e->prof.push_back(r);
        }
    }
};
// engine_encoder.cpp
int run_gemm(mellow_engine* e, const GemmArgs& a);
int run_gemm_apb(mellow_engine* e, const GemmArgs& a, const void* a3, hipStream_t st = nullptr, const void* a3_scales = nullptr);
GemmArgs lin(const float* A, int64_t lda, int M, const Packed& w, float* C, int64_t ldc, const float* bias);
int run_encoder(mellow_engine* e, const float* wav, int n, int64_t n_samples, int want_logmel_only, int apply_bn, float* logmel_out);
// engine_lm.cpp
static inline int rb_of(int B) { return (B + 31) / 32; }
static inline size_t kv_layer_floats(const mellow_engine* e) { return (size_t)e->kv_B * 3 * e->kv_Tmax * 64; }
// loop bookkeeping fused into the arg-max kernel (reference wrapper.py:232-249): only mellow_generate records
struct RecordArgs {
    bool embed_next = false;
};
int ensure_lm(mellow_engine* e, int B, int T, int Tmax, int ctx_end = 0);
int clear_page_tails(mellow_engine* e, int T, int t_end);
LoopArgs loop_args(mellow_engine* e);
int run_lm_head(mellow_engine* e, int B, int pending_kcd, const RecordArgs* rec);
int run_prefill(mellow_engine* e, int B, int T, const RecordArgs* rec, bool all_positions = false);
int enqueue_decode_layer_range(mellow_engine* e, int B, int l_begin, int l_end, bool inc_pos);
int enqueue_decode_layers(mellow_engine* e, int B, const RecordArgs* rec);
int ensure_prefill_streams(mellow_engine* e);      // creates + probes the split prefill's side streams; may lower e->prefill_parts to 1
int encode_pair_to_prefix(mellow_engine* e, const float* a1, const float* a2, int64_t n_samples, const int32_t* ids, int B, float* prefix_out);
void clear_bad_id(mellow_engine* e);
int check_bad_id(mellow_engine* e);
