// Host side of libmellow_hip.so: the C ABI of include/mellow_hip.h.
// Owns weights (re-tiled into MFMA fragment order at load), KV pages, workspaces, the private HIP
// stream, the captured decode-step hipGraph and the event-based profiler.
//
// Reference seams replaced (soham97/mellow):
//   model construction + load_state_dict + .to(cuda)      wrapper.py:59-88      -> create/load/finalize
//   Mellow.generate_prefix_inference                      mellow.py:100-108     -> run_encoder + prefix
//   MellowWrapper._generate_batch                         wrapper.py:197-249    -> mellow_generate
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

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
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
    int prefill_parts = 2;                      // parts of the split LM prefill (MELLOW_PREFILL_SPLIT, read when the engine is created)
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
    Buf lm_xn3, lm_o3, lm_h3;                  // f32x3 mode: the GEMM inputs of LM prefill, pre-split by their producers (APB order)
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
    bool fp8_decode = false;                     // fp8 mode: the decode kernels read e4m3 weights too (off: MELLOW_FP8_DECODE=0)
    bool fp8_decode_act = false;                 // ... and quantise their activations: fp8 matrix pipe (off: MELLOW_FP8_DECODE_ACT=0)
    bool fp8_prefill = true;                     // fp8 mode: e4m3 GEMMs in encoder + prefill (off: MELLOW_FP8_PREFILL=0, a test isolating the decode weights)
    float *head8 = nullptr, *head_sc = nullptr;  // e4m3 lm_head for the decode step
    int f32x3_terms = 0;                         // 0 = off; 6 / 9 = fp32 GEMMs on the bf16 pipe by exact 3-way operand splitting
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

// ---- small helpers ---------------------------------------------------------------------------------------------
static int ensure(mellow_engine* e, mellow_engine::Buf& b, size_t floats) {
    if (b.cap >= floats) return 0;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    HIPCHK(hipMalloc(&b.p, floats * sizeof(float)));
    b.cap = floats;
    return 0;
}
// All weights are carved out of ONE large device allocation: a decode step touches every weight byte exactly once
// (538 MB), and hundreds of small hipMallocs leave the address translation with small page fragments.
static int dev_alloc(mellow_engine* e, float** out, size_t floats) {
    const size_t bytes = (floats * sizeof(float) + 4095) / 4096 * 4096;
    if (!e->arena) {
        size_t want = (size_t)3400 << 20;
        const char* env = getenv("MELLOW_ARENA_MB");
        if (env) want = (size_t)atol(env) << 20;
        if (want) {
            void* p = nullptr;
            if (hipMalloc(&p, want) == hipSuccess) {
                e->arena = reinterpret_cast<char*>(p);
                e->arena_size = want;
                e->allocs.push_back(p);
            } else {
                (void)hipGetLastError();
            }
        }
    }
    if (e->arena && e->arena_used + bytes <= e->arena_size) {
        *out = reinterpret_cast<float*>(e->arena + e->arena_used);
        e->arena_used += bytes;
        return 0;
    }
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, bytes));
    e->allocs.push_back(p);
    *out = reinterpret_cast<float*>(p);
    return 0;
}
static int upload(mellow_engine* e, float** out, const float* src, size_t floats, size_t alloc_floats = 0) {
    if (alloc_floats < floats) alloc_floats = floats;
    CHK(dev_alloc(e, out, alloc_floats));
    if (alloc_floats > floats) HIPCHK(hipMemsetAsync(*out, 0, alloc_floats * sizeof(float), e->stream));
    HIPCHK(hipMemcpyAsync(*out, src, floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
static hipEvent_t next_event(mellow_engine* e) {
    if (e->ev_used == e->ev_pool.size()) {
        hipEvent_t ev;
        hipEventCreate(&ev);
        e->ev_pool.push_back(ev);
    }
    return e->ev_pool[e->ev_used++];
}
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

static int tap(mellow_engine* e, const char* name, const float* src, int64_t n) {
    if (!e->taps_on) return 0;
    auto& b = e->taps[name];
    CHK(ensure(e, b, (size_t)n));
    HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    e->tap_numel[name] = n;
    return 0;
}

// ---- required keys ---------------------------------------------------------------------------------------------------
static std::vector<std::string> build_required(const mellow_config_t* cfg) {
    std::vector<std::string> k;
    std::string E = ENC;
    k.push_back(E + "spectrogram_extractor.stft.conv_real.weight");
    k.push_back(E + "spectrogram_extractor.stft.conv_imag.weight");
    k.push_back(E + "logmel_extractor.melW");
    for (const char* s : {"weight", "bias", "running_mean", "running_var"}) k.push_back(E + "bn0." + s);
    k.push_back(E + "patch_embed.proj.weight");
    k.push_back(E + "patch_embed.proj.bias");
    k.push_back(E + "patch_embed.norm.weight");
    k.push_back(E + "patch_embed.norm.bias");
    for (int s = 0; s < 4; ++s) {
        const int R = 64 >> s;
        for (int b = 0; b < kDepths[s]; ++b) {
            std::string p = E + "layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
            if ((b % 2 == 1) && R > kWin) k.push_back(p + "attn_mask");
            for (const char* t : {"norm1.weight", "norm1.bias", "attn.relative_position_bias_table",
                                  "attn.relative_position_index", "attn.qkv.weight", "attn.qkv.bias",
                                  "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
                                  "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"})
                k.push_back(p + t);
        }
        if (s < 3) {
            std::string p = E + "layers." + std::to_string(s) + ".downsample.";
            k.push_back(p + "reduction.weight");
            k.push_back(p + "norm.weight");
            k.push_back(p + "norm.bias");
        }
    }
    k.push_back(E + "norm.weight");
    k.push_back(E + "norm.bias");
    k.push_back(E + "tscam_conv.weight");
    k.push_back(E + "tscam_conv.bias");
    k.push_back(std::string(C2L) + "weight");
    k.push_back(std::string(C2L) + "bias");
    k.push_back(std::string(PRJ) + "linear1.weight");
    k.push_back(std::string(PRJ) + "linear2.weight");
    k.push_back(std::string(PRJ) + "layer_norm.weight");
    k.push_back(std::string(PRJ) + "layer_norm.bias");
    std::string L = LMK;
    k.push_back(L + "model.embed_tokens.weight");
    const int nl = cfg ? cfg->num_layers : 30;
    for (int l = 0; l < nl; ++l) {
        std::string p = L + "model.layers." + std::to_string(l) + ".";
        for (const char* t : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                              "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                              "mlp.down_proj.weight", "input_layernorm.weight", "post_attention_layernorm.weight"})
            k.push_back(p + t);
    }
    k.push_back(L + "model.norm.weight");
    return k;
}
static const std::vector<std::string>& default_required() {
    static std::vector<std::string> k = build_required(nullptr);
    return k;
}
static bool is_ignored_key(const std::string& k) {
    std::string E = ENC;
    return k == E + "bn0.num_batches_tracked" || k == E + "head.weight" || k == E + "head.bias" ||
           k == std::string(LMK) + "lm_head.weight";  // tied to embed_tokens
}

// ---- host-only helpers -----------------------------------------------------------------------------------------------
static void window_map_host(int R, int shift, int32_t* out) {
    // window-order row m = widx*64 + i*8 + j  <-  token ((hs+shift)%R)*R + (ws+shift)%R, (hs,ws) = window coords
    const int ws = R < kWin ? R : kWin;
    const int nwc = R / ws;
    for (int wr = 0; wr < nwc; ++wr)
        for (int wc = 0; wc < nwc; ++wc)
            for (int i = 0; i < ws; ++i)
                for (int j = 0; j < ws; ++j) {
                    const int hs = wr * ws + i, wsx = wc * ws + j;
                    const int m = (wr * nwc + wc) * ws * ws + i * ws + j;
                    out[m] = ((hs + shift) % R) * R + (wsx + shift) % R;
                }
}
static void pack_weight_host(const float* w, int N, int K, int NP, int KP, float* out) {
    const int K8 = KP / 8;
    for (int nt = 0; nt < NP / 32; ++nt)
        for (int k8 = 0; k8 < K8; ++k8)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = nt * 32 + (lane & 31), k = k8 * 8 + 4 * (lane >> 5) + j;
                    out[(((int64_t)nt * K8 + k8) * 64 + lane) * 4 + j] = (n < N && k < K) ? w[(int64_t)n * K + k] : 0.f;
                }
}

// Rotary tables the way transformers' LlamaRotaryEmbedding builds them: inv_freq = 1 / theta^(2i/d) and the angle
// position * inv_freq are fp32 (the powers below reproduce torch's fp32 inv_freq bit for bit, tests/test_abi_cpu.py); cos / sin
// are evaluated in double and rounded once, i.e. correctly rounded fp32 -- torch's vectorised fp32 cos/sin are within 1 ulp of
// that, and a binding that needs torch's own last bit loads "mellow.rope_cos/sin" instead (include/mellow_hip.h).
static void rope_tables_host(float theta, int head_dim, int P, float* c, float* s) {
    const int half = head_dim / 2;
    for (int i = 0; i < half; ++i) {
        const float x = (float)(2 * i) / (float)head_dim;
        const float inv = 1.0f / (float)pow((double)theta, (double)x);
        for (int p = 0; p < P; ++p) {
            const float a = inv * (float)p;
            c[(size_t)p * half + i] = (float)cos((double)a);
            s[(size_t)p * half + i] = (float)sin((double)a);
        }
    }
}

extern "C" {

int mellow_abi_version(void) { return MELLOW_ABI_VERSION; }
int mellow_host_rope_tables(float theta, int head_dim, int max_pos, float* cos_out, float* sin_out) {
    if (!(theta > 0.f) || head_dim <= 0 || head_dim % 2 || max_pos <= 0 || !cos_out || !sin_out) return fail("bad rope table arguments");
    rope_tables_host(theta, head_dim, max_pos, cos_out, sin_out);
    return 0;
}
const char* mellow_last_error(void) { return g_err.c_str(); }
int mellow_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int mellow_engine_num_required(void) { return (int)default_required().size(); }
const char* mellow_engine_required_key(int i) {
    const auto& k = default_required();
    if (i < 0 || i >= (int)k.size()) return nullptr;
    return k[i].c_str();
}
int mellow_prof_num_families(void) { return PF_COUNT; }
const char* mellow_prof_family_name(int i) { return (i >= 0 && i < PF_COUNT) ? kFamilyNames[i] : nullptr; }

int mellow_host_window_map(int R, int shift, int32_t* out) {
    if (R <= 0 || (R % 8 && R > 8) || !out) return fail("bad window map arguments");
    window_map_host(R, shift, out);
    return 0;
}
int mellow_host_pack_weight(const float* w, int N, int K, int npad, float* out, int64_t out_capacity) {
    if (!w || !out || N <= 0 || K <= 0 || npad <= 0 || npad % 32) return fail("bad pack arguments");
    const int NP = rup(N, npad), KP = rup(K, 32);
    if (out_capacity < (int64_t)NP * KP) return fail("pack output too small: need %lld floats", (long long)NP * KP);
    pack_weight_host(w, N, K, NP, KP, out);
    return 0;
}

int mellow_engine_create(const mellow_config_t* cfg, int device, mellow_engine_t** out) {
    if (!cfg || !out) return fail("null argument");
    if (cfg->abi_version != MELLOW_ABI_VERSION) return fail("ABI version mismatch: caller %d, library %d", cfg->abi_version, MELLOW_ABI_VERSION);
    if (cfg->hidden_size != 576 || cfg->head_dim != 64 || cfg->num_heads != 9 || cfg->num_kv_heads != 3 ||
        cfg->intermediate_size != 1536)
        return fail("unsupported decoder geometry (engine is built for SmolLM2-135M: hidden 576, 9q/3kv heads x 64, inter 1536)");
    if (cfg->prefix_len != 2 * 129 + 2 + cfg->text_len) return fail("prefix_len must be 2*129+2+text_len");
    if (cfg->vocab_size % 128) return fail("vocab_size must be a multiple of 128");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("no HIP device available: the Mellow engine has no CPU fallback");
    if (device < 0 || device >= n) return fail("device %d out of range (%d devices)", device, n);
    HIPCHK(hipSetDevice(device));
    mellow_engine* e = new mellow_engine();
    e->cfg = *cfg;
    e->device = device;
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (const char* pp = getenv("MELLOW_PREFILL_SPLIT")) e->prefill_parts = atoi(pp);
    HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));      // (streams: on first use)
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&e->ev_phase[i]));
    const char* ng = getenv("MELLOW_NO_GRAPH");
    if (ng && ng[0] == '1') e->use_graph = false;
    *out = e;
    // the default numeric mode (include/mellow_hip.h): fp32-accurate GEMMs on the bf16 matrix pipe -- the mode bench.py measures
    // and every engine-level parity test runs (next to the exact-fp32 MFMA mode); mellow_engine_set_precision overrides it
    return mellow_engine_set_precision(e, MELLOW_PRECISION_F32X3);
}

void mellow_engine_destroy(mellow_engine_t* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->step_exec) hipGraphExecDestroy(e->step_exec);
    if (e->step_exec8) hipGraphExecDestroy(e->step_exec8);
    for (void* p : e->allocs) hipFree(p);      // a fork's list holds only what it allocated itself (resample banks): the weights are its parent's
    mellow_engine::Buf* bufs[] = {&e->wavcat, &e->wpad, &e->power, &e->logmel, &e->X0, &e->X1, &e->T, &e->QKV, &e->H, &e->ats,
                                  &e->fpx, &e->fpxavg, &e->latv, &e->emb33, &e->e1, &e->gbuf, &e->sbuf, &e->proj33,
                                  &e->lm_x, &e->lm_xn, &e->lm_q, &e->lm_o, &e->lm_h, &e->lm_xn3, &e->lm_o3, &e->lm_h3, &e->kcache, &e->vcache, &e->dec,
                                  &e->dlogits, &e->cand, &e->out_tok};
    for (auto* b : bufs)
        if (b->p) hipFree(b->p);
    for (auto& kv : e->taps)
        if (kv.second.p) hipFree(kv.second.p);
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    for (int i = 0; i < 4; ++i)
        if (e->ev_phase[i]) hipEventDestroy(e->ev_phase[i]);
    if (e->d_tokens) hipFree(e->d_tokens);
    if (e->h_progress) hipHostFree(e->h_progress);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    for (int i = 0; i < 3; ++i) {
        if (e->ev_join[i]) hipEventDestroy(e->ev_join[i]);
        if (e->stream2[i]) hipStreamDestroy(e->stream2[i]);
    }
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

int mellow_engine_load_tensor(mellow_engine_t* e, const char* key, const void* data, const int64_t* shape, int ndim,
                              int dtype) {
    if (!e || !key || !data) return fail("null argument");
    if (e->finalized) return fail("engine already finalized");
    std::string k = key;
    if (k.rfind("module.", 0) == 0) k = k.substr(7);  // DataParallel prefix (reference wrapper.py:78-82)
    if (is_ignored_key(k)) return 0;
    bool known = (k == "mellow.rope_cos" || k == "mellow.rope_sin");
    if (!known) {
        for (const auto& r : build_required(&e->cfg))
            if (r == k) { known = true; break; }
    }
    if (!known) return fail("unexpected key in state_dict: %s", key);
    HostTensor t;
    t.dtype = dtype;
    for (int i = 0; i < ndim; ++i) t.shape.push_back(shape[i]);
    const size_t esz = dtype == MELLOW_I64 ? 8 : 4;
    t.data.resize((size_t)t.numel() * esz);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(t.data.data(), data, t.data.size(), hipMemcpyDefault));
    e->host[k] = std::move(t);
    return 0;
}

}  // extern "C"

// ---- finalize helpers --------------------------------------------------------------------------------------------------
static const HostTensor* get(mellow_engine* e, const std::string& k) {
    auto it = e->host.find(k);
    return it == e->host.end() ? nullptr : &it->second;
}
static int expect_shape(const HostTensor* t, const std::string& k, std::initializer_list<int64_t> shp) {
    if (!t) return fail("missing key in state_dict: %s", k.c_str());
    if (t->dtype != MELLOW_F32) return fail("%s: expected float32", k.c_str());
    if (t->shape.size() != shp.size())
        return fail("size mismatch for %s: rank %d in the checkpoint, %d expected", k.c_str(), (int)t->shape.size(), (int)shp.size());
    size_t i = 0;
    for (auto d : shp) {
        if (t->shape[i] != d)
            return fail("size mismatch for %s: dimension %d is %lld in the checkpoint, %lld expected", k.c_str(), (int)i,
                        (long long)t->shape[i], (long long)d);
        ++i;
    }
    return 0;
}
static int up_vec(mellow_engine* e, const std::string& k, int64_t n, float** out, int64_t pad_to = 0) {
    const HostTensor* t = get(e, k);
    CHK(expect_shape(t, k, {n}));
    return upload(e, out, t->f(), (size_t)n, (size_t)pad_to);
}
// pack host row-major [N][K] (optionally two sources for pairs) into P-layout on device
static int make_packed(mellow_engine* e, const float* w0, const float* w1, int N, int K, Packed* out) {
    Packed p;
    p.N = N;
    p.K = K;
    p.KP = rup(K, 32);
    p.Nw = w1 ? 64 * ((N + 31) / 32) : N;
    p.NP = rup(p.Nw, 128);
    float *d0 = nullptr, *d1 = nullptr;
    HIPCHK(hipMalloc(&d0, (size_t)N * K * sizeof(float)));
    HIPCHK(hipMemcpy(d0, w0, (size_t)N * K * sizeof(float), hipMemcpyHostToDevice));
    if (w1) {
        HIPCHK(hipMalloc(&d1, (size_t)N * K * sizeof(float)));
        HIPCHK(hipMemcpy(d1, w1, (size_t)N * K * sizeof(float), hipMemcpyHostToDevice));
    }
    CHK(dev_alloc(e, &p.p, (size_t)p.NP * p.KP));
    if (w1) launch_pack_weight_pairs(d0, d1, N, K, K, p.p, p.NP, p.KP, e->stream);
    else launch_pack_weight(d0, N, K, K, p.p, p.NP, p.KP, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipFree(d0));
    if (d1) HIPCHK(hipFree(d1));
    if (e->f32x3_terms && p.KP % 16 == 0 && !e->decode_only_weight) {
        float* pb = nullptr;
        CHK(dev_alloc(e, &pb, ((size_t)p.NP * p.KP * 6 + 3) / 4));
        launch_pack_bf16x3(p.p, p.NP, p.KP, pb, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(e->stream));
        e->bf_w[p.p] = pb;
    }
    if (e->fp8 && p.KP % 64 == 0 && !e->decode_only_weight) {
        float *w8f = nullptr, *sc = nullptr;
        CHK(dev_alloc(e, &w8f, ((size_t)p.NP * p.KP + 3) / 4));
        CHK(dev_alloc(e, &sc, (size_t)p.NP));
        launch_pack_fp8(p.p, p.NP, p.KP, reinterpret_cast<uint8_t*>(w8f), sc, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(e->stream));
        e->fp8_w[p.p] = {reinterpret_cast<uint8_t*>(w8f), sc};
    }
    *out = p;
    return 0;
}
// e4m3 copy of a packed decode weight (tiles x slots float4 slots, `rows` packed rows per tile)
static int make_dec_fp8(mellow_engine* e, const float* Wp, int tiles, int slots, int rows, float** out8, float** scale) {
    CHK(dev_alloc(e, out8, (size_t)tiles * slots));
    CHK(dev_alloc(e, scale, (size_t)tiles * rows));
    launch_pack_dec_fp8(Wp, tiles, slots, rows, *out8, *scale, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
static int make_packed16(mellow_engine* e, const float* w, int N, int K, float** out) {
    if (N % 16 || K % 16) return fail("P16 packing needs N and K multiples of 16");
    float* d0 = nullptr;
    HIPCHK(hipMalloc(&d0, (size_t)N * K * sizeof(float)));
    HIPCHK(hipMemcpy(d0, w, (size_t)N * K * sizeof(float), hipMemcpyHostToDevice));
    CHK(dev_alloc(e, out, (size_t)N * K));
    launch_pack_weight16(d0, N, K, *out, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipFree(d0));
    return 0;
}
static int pack_key(mellow_engine* e, const std::string& k, int N, int K, Packed* out) {
    const HostTensor* t = get(e, k);
    CHK(expect_shape(t, k, {N, K}));
    return make_packed(e, t->f(), nullptr, N, K, out);
}

static int alloc_state_words(mellow_engine* e);
extern "C" int mellow_engine_finalize(mellow_engine_t* e) {
    if (!e) return fail("null engine");
    if (e->finalized) return 0;
    HIPCHK(hipSetDevice(e->device));
    for (const auto& k : build_required(&e->cfg))
        if (!get(e, k)) return fail("missing key in state_dict: %s", k.c_str());
    const std::string E = ENC;
    // ---- front-end: DFT (re/im pairs), mel (transposed), bn0 as alpha/beta ----
    {
        const std::string kr = E + "spectrogram_extractor.stft.conv_real.weight", ki = E + "spectrogram_extractor.stft.conv_imag.weight";
        CHK(expect_shape(get(e, kr), kr, {kNfreq, 1, kNfft}));
        CHK(expect_shape(get(e, ki), ki, {kNfreq, 1, kNfft}));
        CHK(make_packed(e, get(e, kr)->f(), get(e, ki)->f(), kNfreq, kNfft, &e->dft));
        // MELLOW_STFT_FFT=0: the DFT GEMM on the split kernel; MELLOW_X3_STFT=0: the whole front-end on the exact fp32 kernel
        static const bool no_fft = (getenv("MELLOW_STFT_FFT") && getenv("MELLOW_STFT_FFT")[0] == '0') ||
                                   (getenv("MELLOW_X3_STFT") && getenv("MELLOW_X3_STFT")[0] == '0');
        if (e->f32x3_terms && !no_fft && kNfft == 1024) {
            // the reference builds these weights as window[n] * cos / -sin(2 pi k n / N) (torchlibrosa STFT, frozen parameters);
            // a checkpoint that holds anything else keeps the GEMM.  Row k = 0 of the real part IS the window.
            const float *wr = get(e, kr)->f(), *wi = get(e, ki)->f();
            double wmax = 0.0, dev = 0.0;
            for (int n = 0; n < kNfft; ++n) wmax = std::max(wmax, (double)fabsf(wr[n]));
            for (int k = 0; k < kNfreq; ++k)
                for (int n = 0; n < kNfft; ++n) {
                    const double a = 2.0 * M_PI * (double)((int64_t)k * n % kNfft) / kNfft, w0 = wr[n];
                    dev = std::max(dev, fabs((double)wr[(size_t)k * kNfft + n] - w0 * cos(a)));
                    dev = std::max(dev, fabs(fabs((double)wi[(size_t)k * kNfft + n]) - fabs(w0 * sin(a))));
                }
            if (wmax > 0.0 && dev <= 1e-6 * wmax) {
                std::vector<float> t1((size_t)16 * 64 * 2), t2((size_t)4 * 16 * 2);
                for (int k1 = 0; k1 < 16; ++k1)
                    for (int b = 0; b < 64; ++b) {
                        const double a = -2.0 * M_PI * (double)(b * k1) / 1024.0;
                        t1[((size_t)k1 * 64 + b) * 2] = (float)cos(a); t1[((size_t)k1 * 64 + b) * 2 + 1] = (float)sin(a);
                    }
                for (int d = 0; d < 4; ++d)
                    for (int q = 0; q < 16; ++q) {
                        const double a = -2.0 * M_PI * (double)(d * q) / 64.0;
                        t2[((size_t)d * 16 + q) * 2] = (float)cos(a); t2[((size_t)d * 16 + q) * 2 + 1] = (float)sin(a);
                    }
                CHK(upload(e, &e->fft_win, wr, kNfft));
                CHK(upload(e, &e->fft_tw1, t1.data(), t1.size()));
                CHK(upload(e, &e->fft_tw2, t2.data(), t2.size()));
            }
        }
        const std::string km = E + "logmel_extractor.melW";
        CHK(expect_shape(get(e, km), km, {kNfreq, kMel}));
        std::vector<float> mt((size_t)kMel * kNfreq);
        const float* mw = get(e, km)->f();
        for (int k = 0; k < kNfreq; ++k)
            for (int n = 0; n < kMel; ++n) mt[(size_t)n * kNfreq + k] = mw[(size_t)k * kMel + n];
        CHK(make_packed(e, mt.data(), nullptr, kMel, kNfreq, &e->mel));
        const HostTensor *w = get(e, E + "bn0.weight"), *b = get(e, E + "bn0.bias"), *rm = get(e, E + "bn0.running_mean"),
                         *rv = get(e, E + "bn0.running_var");
        for (const char* nm : {"bn0.weight", "bn0.bias", "bn0.running_mean", "bn0.running_var"})
            CHK(expect_shape(get(e, E + nm), E + nm, {kMel}));
        std::vector<float> al(kMel), be(kMel);
        for (int i = 0; i < kMel; ++i) {
            // eval BatchNorm: y = x*alpha + beta with alpha = w/sqrt(var+eps), beta = b - mean*alpha (fp32)
            const float invstd = 1.0f / sqrtf(rv->f()[i] + 1e-5f);
            al[i] = invstd * w->f()[i];
            be[i] = b->f()[i] - rm->f()[i] * al[i];
        }
        CHK(upload(e, &e->bn_alpha, al.data(), kMel));
        CHK(upload(e, &e->bn_beta, be.data(), kMel));
    }
    // ---- patch embed ----
    {
        const std::string k = E + "patch_embed.proj.weight";
        CHK(expect_shape(get(e, k), k, {96, 1, 4, 4}));
        CHK(upload(e, &e->pe_w, get(e, k)->f(), 96 * 16));
        CHK(up_vec(e, E + "patch_embed.proj.bias", 96, &e->pe_b));
        CHK(up_vec(e, E + "patch_embed.norm.weight", 96, &e->pe_nw));
        CHK(up_vec(e, E + "patch_embed.norm.bias", 96, &e->pe_nb));
    }
    // ---- Swin stages ----
    for (int s = 0; s < 4; ++s) {
        const int C = 96 << s, nH = kHeads[s], R = 64 >> s, nW = (R / kWin) * (R / kWin);
        for (int sh = 0; sh < 2; ++sh) {
            if (R <= kWin) continue;  // single window: identity order
            std::vector<int32_t> m((size_t)R * R);
            window_map_host(R, sh ? kWin / 2 : 0, m.data());
            float* d = nullptr;
            CHK(upload(e, &d, reinterpret_cast<const float*>(m.data()), m.size()));
            e->win_map[s][sh] = reinterpret_cast<int32_t*>(d);
        }
        for (int b = 0; b < kDepths[s]; ++b) {
            const std::string p = E + "layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
            SwinBlockW w{};
            CHK(up_vec(e, p + "norm1.weight", C, &w.n1w));
            CHK(up_vec(e, p + "norm1.bias", C, &w.n1b));
            CHK(up_vec(e, p + "norm2.weight", C, &w.n2w));
            CHK(up_vec(e, p + "norm2.bias", C, &w.n2b));
            CHK(pack_key(e, p + "attn.qkv.weight", 3 * C, C, &w.qkv));
            CHK(pack_key(e, p + "attn.proj.weight", C, C, &w.proj));
            CHK(pack_key(e, p + "mlp.fc1.weight", 4 * C, C, &w.fc1));
            CHK(pack_key(e, p + "mlp.fc2.weight", C, 4 * C, &w.fc2));
            CHK(up_vec(e, p + "attn.qkv.bias", 3 * C, &w.qkv_b, w.qkv.NP));
            CHK(up_vec(e, p + "attn.proj.bias", C, &w.proj_b, w.proj.NP));
            CHK(up_vec(e, p + "mlp.fc1.bias", 4 * C, &w.fc1_b, w.fc1.NP));
            CHK(up_vec(e, p + "mlp.fc2.bias", C, &w.fc2_b, w.fc2.NP));
            // expanded relative position bias: bias[h][i][j] = table[index[i][j]][h] (htsat.py:314-316)
            const HostTensor *tb = get(e, p + "attn.relative_position_bias_table"), *ix = get(e, p + "attn.relative_position_index");
            CHK(expect_shape(tb, p + "attn.relative_position_bias_table", {225, nH}));
            if (ix->numel() != 64 * 64) return fail("size mismatch for %sattn.relative_position_index", p.c_str());
            std::vector<float> be((size_t)nH * 4096);
            for (int i = 0; i < 4096; ++i) {
                int64_t id;
                if (ix->dtype == MELLOW_I64) id = reinterpret_cast<const int64_t*>(ix->data.data())[i];
                else id = reinterpret_cast<const int32_t*>(ix->data.data())[i];
                if (id < 0 || id >= 225) return fail("relative_position_index out of range");
                for (int h = 0; h < nH; ++h) be[(size_t)h * 4096 + i] = tb->f()[id * nH + h];
            }
            CHK(upload(e, &w.bias_exp, be.data(), be.size()));
            w.mask = nullptr;
            if ((b % 2 == 1) && R > kWin) {
                const std::string km = p + "attn_mask";
                CHK(expect_shape(get(e, km), km, {nW, 64, 64}));
                CHK(upload(e, &w.mask, get(e, km)->f(), (size_t)nW * 4096));
            }
            e->blocks[s].push_back(w);
        }
        if (s < 3) {
            const std::string p = E + "layers." + std::to_string(s) + ".downsample.";
            CHK(up_vec(e, p + "norm.weight", 4 * C, &e->merge[s].nw));
            CHK(up_vec(e, p + "norm.bias", 4 * C, &e->merge[s].nb));
            CHK(pack_key(e, p + "reduction.weight", 2 * C, 4 * C, &e->merge[s].red));
        }
    }
    // ---- tail ----
    CHK(up_vec(e, E + "norm.weight", kEncOut, &e->fn_w));
    CHK(up_vec(e, E + "norm.bias", kEncOut, &e->fn_b));
    {
        const std::string k = E + "tscam_conv.weight";
        CHK(expect_shape(get(e, k), k, {kClasses, kEncOut, 2, 3}));
        // conv weight [o][ch][cf][dt] -> GEMM weight [o][(cf*3+dt)*768 + ch]
        std::vector<float> wt((size_t)kClasses * 4608);
        const float* src = get(e, k)->f();
        for (int o = 0; o < kClasses; ++o)
            for (int ch = 0; ch < kEncOut; ++ch)
                for (int cf = 0; cf < 2; ++cf)
                    for (int dt = 0; dt < 3; ++dt)
                        wt[(size_t)o * 4608 + (cf * 3 + dt) * 768 + ch] = src[(((size_t)o * kEncOut + ch) * 2 + cf) * 3 + dt];
        CHK(make_packed(e, wt.data(), nullptr, kClasses, 4608, &e->tscam));
        CHK(up_vec(e, E + "tscam_conv.bias", kClasses, &e->tscam_b, e->tscam.NP));
    }
    CHK(pack_key(e, std::string(C2L) + "weight", kEncOut, kClasses, &e->c2l));
    CHK(up_vec(e, std::string(C2L) + "bias", kEncOut, &e->c2l_b, e->c2l.NP));
    CHK(pack_key(e, std::string(PRJ) + "linear1.weight", kProj, kEncOut, &e->lin1));
    CHK(pack_key(e, std::string(PRJ) + "linear2.weight", kProj, kProj, &e->lin2));
    CHK(up_vec(e, std::string(PRJ) + "layer_norm.weight", kProj, &e->pln_w));
    CHK(up_vec(e, std::string(PRJ) + "layer_norm.bias", kProj, &e->pln_b));
    {
        std::vector<int32_t> m(32);
        for (int i = 0; i < 32; ++i) m[i] = i + 1;
        float* d = nullptr;
        CHK(upload(e, &d, reinterpret_cast<const float*>(m.data()), 32));
        e->emb_row_map = reinterpret_cast<int32_t*>(d);
    }
    // ---- LM ----
    const std::string L = LMK;
    const int H = e->cfg.hidden_size, V = e->cfg.vocab_size, I = e->cfg.intermediate_size;
    {
        const std::string k = L + "model.embed_tokens.weight";
        CHK(expect_shape(get(e, k), k, {V, H}));
        CHK(upload(e, &e->embed, get(e, k)->f(), (size_t)V * H));
        e->decode_only_weight = true;                 // the lm_head runs in the decode kernels only (last position)
        CHK(make_packed(e, get(e, k)->f(), nullptr, V, H, &e->lm_head));
        e->decode_only_weight = false;
        if (e->fp8_decode) CHK(make_dec_fp8(e, e->lm_head.p, e->lm_head.NP / 32, (e->lm_head.KP / 8) * 64, 32, &e->head8, &e->head_sc));
    }
    // scratch for the load-time weight composition of dec_qkv2_kernel (fp32 decode weights only)
    float *cmpF = nullptr, *cmpD = nullptr, *cmpQ = nullptr, *cmpCat = nullptr;
    const bool no_fuse = getenv("MELLOW_DECODE_FUSE") && getenv("MELLOW_DECODE_FUSE")[0] == '0';   // keep the 5-launch layer (read per engine)
    const bool fuse = !no_fuse && H == 576 && I == 1536;
    if (fuse) {
        HIPCHK(hipMalloc(&cmpF, (size_t)960 * 576 * 4));
        HIPCHK(hipMalloc(&cmpD, (size_t)576 * 1536 * 4));
        HIPCHK(hipMalloc(&cmpQ, (size_t)960 * 1536 * 4));
        HIPCHK(hipMalloc(&cmpCat, (size_t)1024 * 2112 * 4));     // fp32: [W' | Q] row-major; fp8 mode: Q alone in P-layout (1024 x 1536)
    }
    for (int l = 0; l < e->cfg.num_layers; ++l) {
        const std::string p = L + "model.layers." + std::to_string(l) + ".";
        LMLayerW w{};
        const HostTensor *q = get(e, p + "self_attn.q_proj.weight"), *k = get(e, p + "self_attn.k_proj.weight"),
                         *v = get(e, p + "self_attn.v_proj.weight");
        CHK(expect_shape(q, p + "self_attn.q_proj.weight", {576, H}));
        CHK(expect_shape(k, p + "self_attn.k_proj.weight", {192, H}));
        CHK(expect_shape(v, p + "self_attn.v_proj.weight", {192, H}));
        std::vector<float> qkv((size_t)960 * H);
        memcpy(qkv.data(), q->f(), (size_t)576 * H * 4);
        memcpy(qkv.data() + (size_t)576 * H, k->f(), (size_t)192 * H * 4);
        memcpy(qkv.data() + (size_t)768 * H, v->f(), (size_t)192 * H * 4);
        CHK(make_packed(e, qkv.data(), nullptr, 960, H, &w.qkv));
        CHK(pack_key(e, p + "self_attn.o_proj.weight", H, 576, &w.o));
        const HostTensor *g = get(e, p + "mlp.gate_proj.weight"), *u = get(e, p + "mlp.up_proj.weight");
        CHK(expect_shape(g, p + "mlp.gate_proj.weight", {I, H}));
        CHK(expect_shape(u, p + "mlp.up_proj.weight", {I, H}));
        CHK(make_packed(e, g->f(), u->f(), I, H, &w.gateup));
        CHK(pack_key(e, p + "mlp.down_proj.weight", H, I, &w.down));
        CHK(up_vec(e, p + "input_layernorm.weight", H, &w.in_ln));
        CHK(up_vec(e, p + "post_attention_layernorm.weight", H, &w.post_ln));
        {   // decode copies: fold the norm weights into the columns; 16-row tiles for the complete-output o_proj
            const HostTensor *l1 = get(e, p + "input_layernorm.weight"), *l2 = get(e, p + "post_attention_layernorm.weight");
            std::vector<float> f(qkv);
            for (int n = 0; n < 960; ++n)
                for (int kk = 0; kk < H; ++kk) f[(size_t)n * H + kk] = qkv[(size_t)n * H + kk] * l1->f()[kk];
            e->decode_only_weight = true;
            CHK(make_packed(e, f.data(), nullptr, 960, H, &w.qkv_f));
            e->decode_only_weight = false;
            if (fuse && l > 0) {
                // Q = W'_l . Wd_{l-1} in fp64, rounded once; then [W'_l | Q] re-tiled into P-layout
                const std::string kd = L + "model.layers." + std::to_string(l - 1) + ".mlp.down_proj.weight";
                CHK(expect_shape(get(e, kd), kd, {H, I}));
                HIPCHK(hipMemcpyAsync(cmpF, f.data(), (size_t)960 * 576 * 4, hipMemcpyHostToDevice, e->stream));
                HIPCHK(hipMemcpyAsync(cmpD, get(e, kd)->f(), (size_t)576 * 1536 * 4, hipMemcpyHostToDevice, e->stream));
                launch_compose_f64(cmpF, cmpD, cmpQ, 960, 1536, 576, e->stream);
                if (e->fp8_decode) {
                    // e4m3 decode weights: the composed part is quantised on its own (its rows have their own magnitude); the
                    // W' part and the down weight of the launch are the unfused layer's e4m3 copies (qkv8, dn8)
                    launch_pack_weight(cmpQ, 960, 1536, 1536, cmpCat, 1024, 1536, e->stream);
                    HIPCHK(hipGetLastError());
                    CHK(make_dec_fp8(e, cmpCat, 32, (1536 / 8) * 64, 32, &w.q2h8, &w.q2h_sc));
                } else {
                    HIPCHK(hipMemcpy2DAsync(cmpCat, (size_t)2112 * 4, cmpF, (size_t)576 * 4, (size_t)576 * 4, 960, hipMemcpyDeviceToDevice, e->stream));
                    HIPCHK(hipMemcpy2DAsync(cmpCat + 576, (size_t)2112 * 4, cmpQ, (size_t)1536 * 4, (size_t)1536 * 4, 960, hipMemcpyDeviceToDevice, e->stream));
                    CHK(dev_alloc(e, &w.qkv2, (size_t)1024 * 2112));
                    launch_pack_weight(cmpCat, 960, 2112, 2112, w.qkv2, 1024, 2112, e->stream);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipStreamSynchronize(e->stream));
                }
            }
            std::vector<float> gf((size_t)I * H), uf((size_t)I * H);
            for (int n = 0; n < I; ++n)
                for (int kk = 0; kk < H; ++kk) {
                    gf[(size_t)n * H + kk] = g->f()[(size_t)n * H + kk] * l2->f()[kk];
                    uf[(size_t)n * H + kk] = u->f()[(size_t)n * H + kk] * l2->f()[kk];
                }
            {
                // 16-row tile t = gate[8t..8t+7] then up[8t..8t+7]: one workgroup of the decode gate/up kernel owns both
                // halves of 8 hidden units and applies the SwiGLU in its epilogue
                std::vector<float> il((size_t)2 * I * H);
                for (int t = 0; t < I / 8; ++t) {
                    memcpy(il.data() + (size_t)(2 * t) * 8 * H, gf.data() + (size_t)t * 8 * H, (size_t)8 * H * 4);
                    memcpy(il.data() + (size_t)(2 * t + 1) * 8 * H, uf.data() + (size_t)t * 8 * H, (size_t)8 * H * 4);
                }
                CHK(make_packed16(e, il.data(), 2 * I, H, &w.gu16));
            }
            CHK(make_packed16(e, get(e, p + "self_attn.o_proj.weight")->f(), H, 576, &w.o16));
        }
        if (e->fp8_decode) {
            CHK(make_dec_fp8(e, w.qkv_f.p, w.qkv_f.NP / 32, (w.qkv_f.KP / 8) * 64, 32, &w.qkv8, &w.qkv_sc));
            CHK(make_dec_fp8(e, w.o16, H / 16, (576 / 16) * 64, 16, &w.o8, &w.o_sc));
            CHK(make_dec_fp8(e, w.gu16, 2 * I / 16, (H / 16) * 64, 16, &w.gu8, &w.gu_sc));
            CHK(make_dec_fp8(e, w.down.p, w.down.NP / 32, (w.down.KP / 8) * 64, 32, &w.dn8, &w.dn_sc));
        }
        e->layers.push_back(w);
    }
    if (cmpF) { HIPCHK(hipFree(cmpF)); HIPCHK(hipFree(cmpD)); HIPCHK(hipFree(cmpQ)); HIPCHK(hipFree(cmpCat)); }
    CHK(up_vec(e, L + "model.norm.weight", H, &e->final_norm));
    // ---- RoPE tables [max_pos][32]: supplied by the host wrapper (computed the HF way with torch) or built here ----
    {
        const int P = e->cfg.max_positions;
        const HostTensor *tc = get(e, "mellow.rope_cos"), *ts = get(e, "mellow.rope_sin");
        std::vector<float> c((size_t)P * 32), s((size_t)P * 32);
        if (tc && ts && tc->numel() == (int64_t)P * 32 && ts->numel() == (int64_t)P * 32) {
            memcpy(c.data(), tc->f(), c.size() * 4);
            memcpy(s.data(), ts->f(), s.size() * 4);
        } else {
            rope_tables_host(e->cfg.rope_theta, 64, P, c.data(), s.data());
        }
        CHK(upload(e, &e->rope_cos, c.data(), c.size()));
        CHK(upload(e, &e->rope_sin, s.data(), s.size()));
    }
    CHK(alloc_state_words(e));
    e->host.clear();
    e->finalized = true;
    return 0;
}

// per-context device words of the generation loop (position, stop bookkeeping, block liveness) + the mapped progress word
static int alloc_state_words(mellow_engine* e) {
    HIPCHK(hipMalloc(&e->d_tokens, 4096 * sizeof(int32_t)));
    HIPCHK(hipMemset(e->d_tokens, 0, 4096 * sizeof(int32_t)));
    e->d_step = e->d_tokens + 1024;
    e->d_pos = e->d_tokens + 1025;
    e->d_nseen = e->d_tokens + 1026;
    e->d_arrive = e->d_tokens + 1027;
    e->d_ticket = e->d_tokens + 1028;
    e->d_params = e->d_tokens + 1032;
    e->d_blk_left = e->d_tokens + 1040;
    e->d_blk_live = e->d_tokens + 1072;
    e->d_seen = e->d_tokens + 2048;
    e->d_row_of_slot = e->d_tokens + 3072;
    e->d_ncompact = e->d_tokens + 1029;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_progress), 64, hipHostMallocMapped | hipHostMallocCoherent));
    *e->h_progress = 0;
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_progress), e->h_progress, 0));
    return 0;
}

// ---- GEMM wrappers -------------------------------------------------------------------------------------------------------
static int run_gemm(mellow_engine* e, const GemmArgs& a) {
    // f32x3: every dense GEMM of encoder + prefill, the STFT (EPI_POWER, K = 1024, framed A operand) and the mel projection
    // included (-0.6 ms per pass).  Through the split kernel the power spectrum differs from the ORACLE's fp32 conv1d by 2.5e-6
    // of its maximum -- two fp32 summation orders of a 1024-term dot product, squared -- while against an fp64 STFT it is
    // closer than the oracle's own fp32 arithmetic (tests/test_gpu_parity.py::test_encoder_taps holds it to both).
    // MELLOW_X3_STFT=0 keeps the front-end on the exact fp32 kernel.
    static const bool x3_stft = !(getenv("MELLOW_X3_STFT") && getenv("MELLOW_X3_STFT")[0] == '0');
    if (e->f32x3_terms && a.K % 16 == 0 &&
        ((a.a_mode == A_PLAIN && (a.epi == EPI_LINEAR || a.epi == EPI_SWIGLU || a.epi == EPI_QKV_ROPE)) || (x3_stft && a.K >= 192))) {
        auto it = e->bf_w.find(a.Wp);
        if (it != e->bf_w.end()) {
            // fused kernel: A stays fp32 (global and LDS) and is split into its three bf16 terms in registers; the
            // pre-split kernel (launch_split_rows + launch_gemm_bf16x3) remains reachable through mellow_debug_gemm_f32
            GemmArgs g = a;
            g.W8 = reinterpret_cast<const uint8_t*>(it->second);
            ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
            ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 200;
            launch_gemm_bf16x3_fused(g, e->stream);
            return 0;
        }
    }
    if (e->fp8 && e->fp8_prefill && a.a_mode == A_PLAIN && (a.epi == EPI_LINEAR || a.epi == EPI_SWIGLU || a.epi == EPI_QKV_ROPE)) {
        auto it = e->fp8_w.find(a.Wp);
        if (it != e->fp8_w.end()) {
            // quantise the activation rows, then the fp8 MFMA GEMM (same epilogue); profiled as one launch of the family
            const int64_t lda8 = (a.K + 63) / 64 * 64;
            CHK(ensure(e, e->a8, ((size_t)a.M * lda8 + 3) / 4));
            CHK(ensure(e, e->a8_scale, (size_t)a.M));
            GemmArgs g = a;
            g.A8 = reinterpret_cast<const uint8_t*>(e->a8.p); g.lda8 = lda8; g.a_scale = e->a8_scale.p;
            g.W8 = it->second.w8; g.w_scale = it->second.scale;
            ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
            ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 100;
            launch_quant_rows(a.A, a.lda, a.M, a.K, reinterpret_cast<uint8_t*>(e->a8.p), lda8, e->a8_scale.p, e->stream);
            launch_gemm_fp8(g, e->stream);
            return 0;
        }
    }
    ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
    ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi;
    launch_gemm(a, e->stream);
    return 0;
}
// f32x3 mode, LM prefill: the activation arrives pre-split in APB order from its producer (a3) and both operands are staged
// by LDS-DMA (gemm_x3q_kernel); counted in the same profile family as every other dense GEMM
static int run_gemm_apb(mellow_engine* e, const GemmArgs& a, const void* a3, hipStream_t st = nullptr) {
    auto it = e->bf_w.find(a.Wp);
    if (it == e->bf_w.end()) return fail("internal: no bf16-split copy of this weight");
    GemmArgs g = a;
    g.A8 = reinterpret_cast<const uint8_t*>(a3);
    g.W8 = reinterpret_cast<const uint8_t*>(it->second);
    ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
    ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 300;
    launch_gemm_bf16x3_apb(g, st ? st : e->stream);
    return 0;
}
static GemmArgs lin(const float* A, int64_t lda, int M, const Packed& w, float* C, int64_t ldc, const float* bias) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.M = M; g.K = w.KP; g.Wp = w.p; g.Nw = w.Nw; g.N = rup(w.N, 4); g.C = C; g.ldc = ldc; g.bias = bias;
    return g;
}

// ---- encoder --------------------------------------------------------------------------------------------------------------
// wav dev [n][n_samples] -> proj33 [n][33][576] in e->proj33
static int run_encoder(mellow_engine* e, const float* wav, int n, int64_t n_samples, int want_logmel_only, int apply_bn,
                       float* logmel_out) {
    if (n <= 0) return fail("n_clips must be positive");
    if (n_samples % 4 || n_samples < kNfft) return fail("n_samples must be a multiple of 4 and >= 1024");
    hipStream_t s = e->stream;
    const int frames = (int)(n_samples / kHop) + 1;
    const int64_t plen = n_samples + kNfft;
    const int M = n * frames;
    CHK(ensure(e, e->wpad, (size_t)n * plen));
    CHK(ensure(e, e->power, (size_t)M * 544));
    CHK(ensure(e, e->logmel, (size_t)M * 64));
    {
        ProfScope ps(e, PF_MISC, 0, 2.0 * n * plen * 4);
        launch_reflect_pad(wav, n, n_samples, e->wpad.p, plen, kNfft / 2, s);
    }
    if (e->fft_win) {   // A1 as a real FFT (f32x3 mode, weights verified to be the windowed DFT basis): 5 N log2 N flops per frame
        ProfScope ps(e, PF_GEMM, 5.0 * kNfft * 10.0 * M, (double)M * (kNfft + 544) * 4);
        ps.r.M = M; ps.r.N = 544; ps.r.K = kNfft; ps.r.epi = 400;
        launch_stft_fft_power(e->wpad.p, frames, plen, kHop, M, e->fft_win, e->fft_tw1, e->fft_tw2, e->power.p, s);
    } else {   // A1: STFT power as DFT GEMM on the checkpoint's conv weights (htsat.py:864)
        GemmArgs g;
        g.A = e->wpad.p; g.a_mode = A_FRAMES; g.fpc = frames; g.clip_stride = plen; g.hop = kHop;
        g.M = M; g.K = kNfft; g.Wp = e->dft.p; g.Nw = e->dft.Nw; g.N = 544; g.C = e->power.p; g.ldc = 544; g.epi = EPI_POWER;
        CHK(run_gemm(e, g));
    }
    CHK(tap(e, "power", e->power.p, (int64_t)M * 544));
    {   // A2+A3: mel projection, 10*log10, bn0 (htsat.py:865-870)
        GemmArgs g;
        g.A = e->power.p; g.lda = 544; g.M = M; g.K = 544; g.Wp = e->mel.p; g.Nw = 64; g.N = 64;
        g.C = want_logmel_only ? logmel_out : e->logmel.p; g.ldc = 64; g.epi = EPI_LOGMEL;
        g.apply_bn = apply_bn; g.bn_alpha = e->bn_alpha; g.bn_beta = e->bn_beta;
        CHK(run_gemm(e, g));
    }
    if (want_logmel_only) return 0;
    CHK(tap(e, "logmel_bn", e->logmel.p, (int64_t)M * 64));

    // A4/A4': crops
    int n_crops = 1, crop_len = frames, crop_hop = 0;
    if (frames > 1024) {
        n_crops = 0;
        for (int p = 0; p < frames - kLongCrop - 1; p += kLongHop) ++n_crops;
        crop_len = kLongCrop;
        crop_hop = kLongHop;
    }
    const int nv = n * n_crops;
    const int64_t M0 = (int64_t)nv * 4096;
    CHK(ensure(e, e->X0, (size_t)M0 * 96));
    CHK(ensure(e, e->X1, (size_t)M0 * 96));
    CHK(ensure(e, e->T, (size_t)M0 * 96));
    CHK(ensure(e, e->QKV, (size_t)M0 * 288));
    CHK(ensure(e, e->H, (size_t)M0 * 384));
    {
        ProfScope ps(e, PF_MISC, 0, (double)M0 * 96 * 4);
        launch_fold_patch_embed(e->logmel.p, n, frames, n_crops, crop_hop, crop_len, e->pe_w, e->pe_b, e->pe_nw, e->pe_nb,
                                e->X0.p, s);
    }
    CHK(tap(e, "patch", e->X0.p, M0 * 96));
    float *x = e->X0.p, *x2 = e->X1.p, *t = e->T.p;
    for (int st = 0; st < 4; ++st) {
        const int C = 96 << st, R = 64 >> st, N = R * R, nH = kHeads[st];
        const int nW = R > kWin ? (R / kWin) * (R / kWin) : 1;
        const int M1 = nv * N;
        for (int b = 0; b < kDepths[st]; ++b) {
            const SwinBlockW& w = e->blocks[st][b];
            const bool shifted = (b % 2 == 1) && R > kWin;
            const int32_t* map = R > kWin ? e->win_map[st][shifted ? 1 : 0] : nullptr;
            { ProfScope ps(e, PF_NORM, 0, 2.0 * M1 * C * 4); launch_layernorm(x, t, M1, C, w.n1w, w.n1b, map, N, s); }
            CHK(run_gemm(e, lin(t, C, M1, w.qkv, e->QKV.p, 3 * C, w.qkv_b)));
            {
                ProfScope ps(e, PF_WINDOW_ATTN, 4.0 * 64 * 64 * 24 * (double)(M1 / 64) * nH, 4.0 * M1 * C * 4);
                launch_window_attention(e->QKV.p, t, M1, C, nH, w.bias_exp, shifted ? w.mask : nullptr, nW, s);
            }
            {
                GemmArgs g = lin(t, C, M1, w.proj, x, C, w.proj_b);
                g.resid = x; g.ldr = C; g.crow_map = map; g.rows_in = N; g.rows_out = N;
                CHK(run_gemm(e, g));
            }
            { ProfScope ps(e, PF_NORM, 0, 2.0 * M1 * C * 4); launch_layernorm(x, t, M1, C, w.n2w, w.n2b, nullptr, N, s); }
            {
                GemmArgs g = lin(t, C, M1, w.fc1, e->H.p, 4 * C, w.fc1_b);
                g.act = ACT_GELU;
                CHK(run_gemm(e, g));
            }
            {
                GemmArgs g = lin(e->H.p, 4 * C, M1, w.fc2, x, C, w.fc2_b);
                g.resid = x; g.ldr = C;
                CHK(run_gemm(e, g));
            }
        }
        if (st < 3) {
            { ProfScope ps(e, PF_NORM, 0, 2.0 * M1 * C * 4); launch_merge_layernorm(x, t, nv, R, C, e->merge[st].nw, e->merge[st].nb, s); }
            CHK(run_gemm(e, lin(t, 4 * C, M1 / 4, e->merge[st].red, x2, 2 * C, nullptr)));
            float* tmp = x; x = x2; x2 = tmp;
        }
        if (e->taps_on) {
            char nm[16];
            snprintf(nm, sizeof(nm), "stage%d", st);
            const int64_t cnt = st < 3 ? (int64_t)nv * (N / 4) * (2 * C) : (int64_t)nv * N * C;
            CHK(tap(e, nm, x, cnt));
        }
    }
    // ---- tail (htsat.py:742-796, 950-955; mellow.py:48-52) ----
    CHK(ensure(e, e->ats, (size_t)nv * 32 * 4608));
    CHK(ensure(e, e->fpx, (size_t)nv * 32 * 544));
    CHK(ensure(e, e->emb33, (size_t)n * 33 * 768));
    CHK(ensure(e, e->e1, (size_t)n * 33 * 576));
    CHK(ensure(e, e->gbuf, (size_t)n * 33 * 576));
    CHK(ensure(e, e->sbuf, (size_t)n * 33 * 576));
    CHK(ensure(e, e->proj33, (size_t)n * 33 * 576));
    { ProfScope ps(e, PF_NORM, 0, 2.0 * nv * 64 * 768 * 4); launch_layernorm(x, t, nv * 64, kEncOut, e->fn_w, e->fn_b, nullptr, 64, s); }
    float* latent_dst = e->emb33.p;
    int64_t latent_stride = 33 * 768;
    if (n_crops > 1) {
        CHK(ensure(e, e->latv, (size_t)nv * 768));
        CHK(ensure(e, e->fpxavg, (size_t)n * 32 * 544));
        latent_dst = e->latv.p;
        latent_stride = 768;
    }
    { ProfScope ps(e, PF_MISC, 0, 7.0 * nv * 64 * 768 * 4); launch_tail_latent_im2col(t, nv, latent_dst, latent_stride, e->ats.p, s); }
    {
        GemmArgs g = lin(e->ats.p, 4608, nv * 32, e->tscam, e->fpx.p, 544, e->tscam_b);
        g.N = 544; g.act = ACT_SIGMOID;
        CHK(run_gemm(e, g));
    }
    const float* fpx = e->fpx.p;
    if (n_crops > 1) {
        ProfScope ps(e, PF_MISC, 0, 0);
        launch_crop_average(e->fpx.p, n, n_crops, 32 * 544, 32 * 544, e->fpxavg.p, 32 * 544, s);
        launch_crop_average(e->latv.p, n, n_crops, 768, 768, e->emb33.p, 33 * 768, s);
        fpx = e->fpxavg.p;
    }
    CHK(tap(e, "fpx", fpx, (int64_t)n * 32 * 544));
    {   // c2l on the 32 distinct framewise rows -> embedding rows 1..32 (htsat.py:952-954)
        GemmArgs g = lin(fpx, 544, n * 32, e->c2l, e->emb33.p, 768, e->c2l_b);
        g.crow_map = e->emb_row_map; g.rows_in = 32; g.rows_out = 33;
        CHK(run_gemm(e, g));
    }
    CHK(tap(e, "emb33", e->emb33.p, (int64_t)n * 33 * 768));
    CHK(run_gemm(e, lin(e->emb33.p, 768, n * 33, e->lin1, e->e1.p, 576, nullptr)));
    { ProfScope ps(e, PF_MISC, 0, 0); launch_gelu(e->e1.p, e->gbuf.p, (int64_t)n * 33 * 576, s); }
    {
        GemmArgs g = lin(e->gbuf.p, 576, n * 33, e->lin2, e->sbuf.p, 576, nullptr);
        g.resid = e->e1.p; g.ldr = 576;
        CHK(run_gemm(e, g));
    }
    { ProfScope ps(e, PF_NORM, 0, 0); launch_layernorm(e->sbuf.p, e->proj33.p, n * 33, 576, e->pln_w, e->pln_b, nullptr, 33, s); }
    CHK(tap(e, "proj33", e->proj33.p, (int64_t)n * 33 * 576));
    CHK(tap(e, "latent", e->emb33.p, 768));  // first clip's latent row (row 0 of emb33)
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- LM ----------------------------------------------------------------------------------------------------------------------
static inline int rb_of(int B) { return (B + 31) / 32; }
static inline size_t kv_layer_floats(const mellow_engine* e) { return (size_t)e->kv_B * 3 * e->kv_Tmax * 64; }

static int ensure_lm(mellow_engine* e, int B, int T, int Tmax, int ctx_end = 0) {
    if (B > 1024) return fail("batch of %d exceeds the 1024 rows one pass takes (mellow_generate chunks larger batches itself; the decode state block is sized for 32 row blocks)", B);
    if (ctx_end <= 0 || ctx_end > Tmax) ctx_end = Tmax;      // last context length the call will reach (<= page capacity)
    if (Tmax > e->cfg.max_positions) return fail("prefix + max_len = %d exceeds max_positions %d", Tmax, e->cfg.max_positions);
    const size_t Mp = (size_t)B * T;
    CHK(ensure(e, e->lm_x, Mp * 576));
    CHK(ensure(e, e->lm_xn, Mp * 576));
    CHK(ensure(e, e->lm_q, Mp * 576));
    CHK(ensure(e, e->lm_o, Mp * 576));
    CHK(ensure(e, e->lm_h, Mp * 1536));
    if (e->f32x3_terms) {                       // 6 bytes per element, rows padded to whole 128-row panels
        const size_t Mq = (size_t)rup((int)Mp, 128) + 3 * 128;   // + three panels: every part of the split prefill starts on a panel boundary
        CHK(ensure(e, e->lm_xn3, Mq * 576 * 6 / 4));
        CHK(ensure(e, e->lm_o3, Mq * 576 * 6 / 4));
        CHK(ensure(e, e->lm_h3, Mq * 1536 * 6 / 4));
    }
    const int Bp = rb_of(B) * 32;
    if (e->kv_B != Bp || e->kv_Tmax != Tmax) {
        e->kv_B = Bp;
        e->kv_Tmax = Tmax;
        CHK(ensure(e, e->kcache, kv_layer_floats(e) * e->cfg.num_layers));
        CHK(ensure(e, e->vcache, kv_layer_floats(e) * e->cfg.num_layers));
        // the decode attention loads whole key groups before it knows the position and masks them afterwards
        // (weight 0 x value): never-written page slots must hold finite numbers
        HIPCHK(hipMemsetAsync(e->kcache.p, 0, kv_layer_floats(e) * e->cfg.num_layers * sizeof(float), e->stream));
        HIPCHK(hipMemsetAsync(e->vcache.p, 0, kv_layer_floats(e) * e->cfg.num_layers * sizeof(float), e->stream));
        if (e->step_exec) { hipGraphExecDestroy(e->step_exec); e->step_exec = nullptr; }
        if (e->step_exec8) { hipGraphExecDestroy(e->step_exec8); e->step_exec8 = nullptr; }
    }
    {
        // carve the decode-step buffers out of one arena (all sizes are multiples of 64 floats = 256 B)
        const size_t RB = (size_t)Bp / 32, V = (size_t)e->cfg.vocab_size;
        const size_t n_x = (size_t)Bp * 576;
        size_t off = 0;
        auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
        const size_t o_xmidF = take(n_x), o_xnewR = take(n_x), o_xnF = take(n_x);
        const size_t o_dslabF = take(DEC_KC_DOWN * n_x), o_ssq1 = take((size_t)Bp * DEC_KC_QKV), o_rope = take(64);
        const size_t o_pq = take((size_t)DEC_KC_QKV * Bp * 960);
        const size_t o_att = take((size_t)DEC_TS * n_x), o_aml = take((size_t)DEC_TS * 9 * Bp * 2);
        const size_t o_ssq = take((size_t)Bp * 40), o_gu = take(RB * 192 * 256), o_xmidF16 = take(n_x);
        const bool fresh = e->dec.cap < off;
        CHK(ensure(e, e->dec, off));
        CHK(ensure(e, e->dlogits, (size_t)Bp * V));
        CHK(ensure(e, e->cand, (size_t)2 * Bp * (V / 32)));
        if (fresh) {
            // padded batch rows are computed but never read back; start from finite values
            HIPCHK(hipMemsetAsync(e->dec.p, 0, off * sizeof(float), e->stream));
            if (e->step_exec) { hipGraphExecDestroy(e->step_exec); e->step_exec = nullptr; }
            if (e->step_exec8) { hipGraphExecDestroy(e->step_exec8); e->step_exec8 = nullptr; }
        }
        float* p = e->dec.p;
        DecArgs& a = e->da;
        a.rows = Bp; a.RB = (int)RB; a.Tmax = Tmax; a.eps = e->cfg.rms_norm_eps; a.d_pos = e->d_pos; a.inc_pos = 0; a.first = 0;
        a.a8 = e->fp8_decode_act ? 1 : 0;
        a.blk_live = nullptr;                               // mellow_generate turns the per-block early exit on per call
        a.row_of_slot = nullptr;
        // (and the logits store off: the taps mellow_lm_prefill / mellow_lm_decode_step read dlogits, generation does not)
        a.rope_cos = e->rope_cos; a.rope_sin = e->rope_sin;
        a.xmidF = p + o_xmidF; a.xnewR = p + o_xnewR; a.xnF = p + o_xnF;
        a.dslabF = p + o_dslabF; a.slabF_stride4 = (int64_t)(n_x / 4); a.ssq1 = p + o_ssq1; a.rope_cur = p + o_rope;
        {
            // key split of the decode attention: balanced at the END of the reserved context, rounded down to whole
            // passes of a workgroup when that costs at most 4 groups of imbalance
            const int ng_end = (ctx_end - 1 + 3) / 4, chunk = dec_attn_chunk_groups();
            int gs = (ng_end + DEC_TS - 1) / DEC_TS;
            if (gs > chunk && gs % chunk <= 4) gs -= gs % chunk;
            gs = gs < 1 ? 1 : gs;
            if (gs != a.gs) {     // the split is baked into captured launches
                if (e->step_exec) { hipGraphExecDestroy(e->step_exec); e->step_exec = nullptr; }
                if (e->step_exec8) { hipGraphExecDestroy(e->step_exec8); e->step_exec8 = nullptr; }
            }
            a.gs = gs;
        }
        a.pq = p + o_pq; a.attF16 = p + o_att; a.att_ml = p + o_aml; a.ssq = p + o_ssq; a.guF = p + o_gu; a.xmidF16 = p + o_xmidF16;
        a.logits = e->dlogits.p; a.cand_val = e->cand.p; a.cand_idx = reinterpret_cast<int32_t*>(e->cand.p + (size_t)Bp * (V / 32));
    }
    if (Bp > 1024) return fail("batch too large for the decode state block");
    return 0;
}

// The decode attention loads whole key groups before it knows the position and masks them by WEIGHT (exp(-inf) = 0): a
// slot beyond the context must therefore hold a finite value, or 0 x NaN poisons the row.  A fresh page is zeroed when it
// is allocated; a reused one may hold an earlier call's appended keys -- even NaN from a poisoned request -- so the V slots
// beyond the prefix, [T, Tmax), are cleared once per prefill (one coalesced fill kernel, 80 MB at B = 32 / max_len 64: ~20 us; K needs none: a NaN score of a masked key is replaced by -inf with a select).
static int clear_page_tails(mellow_engine* e, int T, int t_end) {
    const int Tmax = e->kv_Tmax;
    if (t_end > Tmax) t_end = Tmax;
    if (T >= t_end) return 0;
    launch_clear_page_slots(e->vcache.p, (int64_t)e->cfg.num_layers * e->kv_B * 3, Tmax, T, t_end, e->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// loop bookkeeping fused into the arg-max kernel (reference wrapper.py:232-249): only mellow_generate records
struct RecordArgs {
    bool embed_next = false;
};
static LoopArgs loop_args(mellow_engine* e) {
    LoopArgs lp;
    lp.out_tokens = reinterpret_cast<int32_t*>(e->out_tok.p);
    lp.params = e->d_params; lp.seen_stop = e->d_seen; lp.n_seen = e->d_nseen; lp.arrive = e->d_arrive; lp.ticket = e->d_ticket;
    lp.host_progress = e->d_progress; lp.T0 = e->cfg.prefix_len;
    if (e->da.blk_live) { lp.blk_left = e->d_blk_left; lp.blk_live = e->d_blk_live; lp.blk_snap = e->d_blk_live + 32; }
    if (e->da.row_of_slot) { lp.row_of_slot = e->d_row_of_slot; lp.n_compactions = e->d_ncompact; }
    return lp;
}

// final norm (+ pending down slabs) + lm_head with fused per-tile arg-max candidates -> dlogits, d_tokens
static int run_lm_head(mellow_engine* e, int B, int pending_kcd, const RecordArgs* rec) {
    const int NT = e->cfg.vocab_size / 32, Bp = e->da.rows;
    { ProfScope ps(e, PF_NORM, 0, (double)(pending_kcd + 2) * Bp * 576 * 4);
      launch_dec_final_norm(e->da, e->final_norm, pending_kcd, e->stream); }
    { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * e->cfg.vocab_size, 576.0 * e->cfg.vocab_size * 4);
      if (e->head8) launch_dec_lm_head(e->da, e->head8, e->lm_head.KP / 8, e->cfg.vocab_size, e->stream, e->head_sc);
      else launch_dec_lm_head(e->da, e->lm_head.p, e->lm_head.KP / 8, e->cfg.vocab_size, e->stream); }
    { ProfScope ps(e, PF_MISC, 0, 0);
      launch_dec_argmax(e->da, B, NT, e->d_tokens, e->embed, (rec && rec->embed_next) ? 1 : 0, rec ? loop_args(e) : LoopArgs(),
                        e->stream);
      if (rec && e->da.row_of_slot) launch_dec_compact(e->da, B, loop_args(e), e->stream); }
    return 0;
}

static int enqueue_decode_layer_range(mellow_engine* e, int B, int l_begin, int l_end, bool inc_pos);
static int run_prefill(mellow_engine* e, int B, int T, const RecordArgs* rec, bool all_positions = false) {
    hipStream_t s = e->stream;
    const int M = B * T, Tmax = e->kv_Tmax;
    const int NL = e->cfg.num_layers;
    float *x = e->lm_x.p, *xn = e->lm_xn.p;
    static const bool no_apb = getenv("MELLOW_X3_NO_APB") != nullptr;        // developer A/B: the register-staged x3p kernel
    const bool apb = e->f32x3_terms && !no_apb;
    // Split prefill (f32x3 mode): the batch is cut into independent parts (2 by default) that run the same launches on their own
    // streams, so the tails and the fill / drain of one part's kernels are covered by another's (every buffer is indexed by row
    // or by example, so a part is an offset; its pre-split operands get their own panel-aligned region).  Measured before it was
    // built with two forked contexts (tools/half_chain_probe.py).  MELLOW_PREFILL_SPLIT=n: n parts (1 = one chain, at most 4).
    int nh = (apb && !e->prof_on) ? e->prefill_parts : 1;
    nh = nh < 1 ? 1 : (nh > 4 ? 4 : nh);
    if (nh > B) nh = B;
    for (int h = 1; h < nh; ++h)
        if (!e->stream2[h - 1]) HIPCHK(hipStreamCreateWithFlags(&e->stream2[h - 1], hipStreamNonBlocking));
    int hb0[4], hB[4];
    size_t prow[4];                                              // first row of each part's panel range
    hipStream_t hs[4] = {s, e->stream2[0], e->stream2[1], e->stream2[2]};
    for (int h = 0, b0 = 0, r = 0; h < nh; ++h) {
        hb0[h] = b0; hB[h] = B / nh + (h < B % nh ? 1 : 0); prow[h] = (size_t)r;
        b0 += hB[h]; r += rup(hB[h] * T, 128);
    }
    const bool split = nh > 1;
    if (split) {
        HIPCHK(hipEventRecord(e->ev_fork, s));
        for (int h = 1; h < nh; ++h) HIPCHK(hipStreamWaitEvent(hs[h], e->ev_fork, 0));
    }
    for (int l = 0; l < NL; ++l) {
        const LMLayerW& w = e->layers[l];
        bool last = false;
        for (int h = 0; h < nh; ++h) {
            hipStream_t st = hs[h];
            const int64_t r0 = (int64_t)hb0[h] * T;
            const int Mh = hB[h] * T, Bh = hB[h];
            float* xh = x + r0 * 576;
            float* xnh = xn + r0 * 576;
            float* qh = e->lm_q.p + r0 * 576;
            float* oh = e->lm_o.p + r0 * 576;
            float* hh = e->lm_h.p + r0 * 1536;
            float* kc = e->kcache.p + kv_layer_floats(e) * l + (size_t)hb0[h] * 3 * Tmax * 64;
            float* vc = e->vcache.p + kv_layer_floats(e) * l + (size_t)hb0[h] * 3 * Tmax * 64;
            // pre-split operand regions of this half (6 bytes per element, whole 128-row panels)
            char* xn3 = apb ? reinterpret_cast<char*>(e->lm_xn3.p) + prow[h] * 576 * 6 : nullptr;
            char* o3 = apb ? reinterpret_cast<char*>(e->lm_o3.p) + prow[h] * 576 * 6 : nullptr;
            char* h3 = apb ? reinterpret_cast<char*>(e->lm_h3.p) + prow[h] * 1536 * 6 : nullptr;
            // f32x3 mode: every GEMM input of the layer is written by its producer already split into three bf16 pieces, in the
            // order the GEMM's LDS stage wants it (APB, common.h), and the GEMM stages both operands by LDS-DMA (x3q)
            if (apb) { ProfScope ps(e, PF_NORM, 0, 2.5 * Mh * 576 * 4); launch_rmsnorm_apb(xh, xn3, Mh, 576, w.in_ln, e->cfg.rms_norm_eps, st); }
            else { ProfScope ps(e, PF_NORM, 0, 2.0 * Mh * 576 * 4); launch_rmsnorm(xh, xnh, Mh, 576, w.in_ln, e->cfg.rms_norm_eps, st); }
            {
                GemmArgs g;
                g.A = xnh; g.lda = 576; g.M = Mh; g.K = 576; g.Wp = w.qkv.p; g.Nw = 960; g.N = 960; g.epi = EPI_QKV_ROPE;
                g.q_out = qh; g.k_cache = kc; g.v_cache = vc; g.rope_cos = e->rope_cos; g.rope_sin = e->rope_sin;
                g.T = T; g.Tmax = Tmax; g.q_heads = 9; g.kv_heads = 3;
                if (apb) CHK(run_gemm_apb(e, g, xn3, st)); else CHK(run_gemm(e, g));
            }
            // The LAST layer only has to produce the final prefix row (nothing consumes the other rows' attention / MLP
            // outputs; their K/V pages were just written above): it is finished below by the decode kernels on B rows.
            if (l == NL - 1 && !all_positions) { last = true; continue; }
            {
                // causal QK^T + PV: 4*64 flops per (query,key) pair per head
                ProfScope ps(e, PF_PREFILL_ATTN, 4.0 * 64 * 9 * (double)Bh * ((double)T * (T + 1) / 2), 0);
                static const bool attn_f32 = getenv("MELLOW_X3_ATTN") && getenv("MELLOW_X3_ATTN")[0] == '0';   // A/B: f32x3 mode on the fp32 kernel
                launch_prefill_attention(qh, kc, vc, oh, apb ? o3 : nullptr, Bh, T, Tmax, e->f32x3_terms != 0 && !attn_f32, st);
            }
            {
                GemmArgs g = lin(oh, 576, Mh, w.o, xh, 576, nullptr);
                g.resid = xh; g.ldr = 576;
                if (apb) CHK(run_gemm_apb(e, g, o3, st)); else CHK(run_gemm(e, g));
            }
            if (apb) { ProfScope ps(e, PF_NORM, 0, 2.5 * Mh * 576 * 4); launch_rmsnorm_apb(xh, xn3, Mh, 576, w.post_ln, e->cfg.rms_norm_eps, st); }
            else { ProfScope ps(e, PF_NORM, 0, 2.0 * Mh * 576 * 4); launch_rmsnorm(xh, xnh, Mh, 576, w.post_ln, e->cfg.rms_norm_eps, st); }
            {
                GemmArgs g;
                g.A = xnh; g.lda = 576; g.M = Mh; g.K = 576; g.Wp = w.gateup.p; g.Nw = 3072; g.N = 1536; g.C = hh; g.ldc = 1536;
                g.epi = EPI_SWIGLU;
                if (apb) { g.C3 = h3; CHK(run_gemm_apb(e, g, xn3, st)); } else CHK(run_gemm(e, g));
            }
            {
                GemmArgs g = lin(hh, 1536, Mh, w.down, xh, 576, nullptr);
                g.resid = xh; g.ldr = 576;
                if (apb) CHK(run_gemm_apb(e, g, h3, st)); else CHK(run_gemm(e, g));
            }
        }
        if (last) break;
    }
    for (int h = 1; h < nh; ++h) {
        HIPCHK(hipEventRecord(e->ev_join[h - 1], hs[h]));
        HIPCHK(hipStreamWaitEvent(s, e->ev_join[h - 1], 0));
    }
    if (all_positions) {        // x = the hidden states after all layers, every position (mellow_lm_forward_logits)
        HIPCHK(hipGetLastError());
        return 0;
    }
    // x now holds the input of the last layer.  Position word = index of the LAST prefix token: the decode kernels
    // treat it as "the new token" (keys 0..T-2 from the pages, key T-1 recomputed and re-appended), and the first
    // kernel of every later decode step advances it; the arg-max records its token at column (*d_pos - prefix_len + 1) = 0.
    { ProfScope ps(e, PF_MISC, 0, 0); launch_dec_load_rows(e->da, B, x, 576, nullptr, T, 0, s); }
    e->cur_B = B;
    e->cur_pos = T;
    e->h_pos_word = T - 1;
    HIPCHK(hipMemcpyAsync(e->d_pos, &e->h_pos_word, sizeof(int32_t), hipMemcpyHostToDevice, s));
    CHK(enqueue_decode_layer_range(e, B, NL - 1, NL, false));
    CHK(run_lm_head(e, B, DEC_KC_DOWN, rec));
    HIPCHK(hipGetLastError());
    return 0;
}

// the 30 decode layers + head at position *d_pos (enqueue only; capture-safe).  5 launches per layer (decode.hip):
//   qkv split-K | attention (RMS scale, RoPE, KV append, key-split flash decoding) | o_proj (merge + residual) |
//   gate/up | down split-K (RMS scale, SwiGLU); the down slabs are summed by the next layer's qkv/attention.
static int run_lm_head(mellow_engine* e, int B, int pending_kcd, const RecordArgs* rec);
static int enqueue_decode_layer_range(mellow_engine* e, int B, int l_begin, int l_end, bool inc_pos) {
    hipStream_t s = e->stream;
    const int Bp = e->da.rows;
    // developer knobs (wrong tokens, timing only): every layer on layer 0's weights / KV pages -- is a decode kernel's time the
    // cold fetch of its weights (538 MB per step cycle through the 256 MB Infinity Cache) or of its KV pages?
    // Compiled in only with -DMELLOW_DEVPROBE (tools/ab_build.sh): the release library has no switch that changes its answers.
#ifdef MELLOW_DEVPROBE
    static const bool same_w = getenv("MELLOW_DEV_SAME_WEIGHTS") != nullptr, same_kv = getenv("MELLOW_DEV_SAME_KV") != nullptr;
    // MELLOW_DEV_SKIP: bit mask of the per-layer launches left out (1 qkv, 2 attention, 4 o_proj, 8 gate/up, 16 down): what a
    // fusion that removes that launch could gain at most
    static const int skip = getenv("MELLOW_DEV_SKIP") ? atoi(getenv("MELLOW_DEV_SKIP")) : 0;
#else
    constexpr bool same_w = false, same_kv = false;
    constexpr int skip = 0;
#endif
    for (int l = l_begin; l < l_end; ++l) {
        const LMLayerW& w = e->layers[same_w ? 0 : l];
        float* kc = e->kcache.p + kv_layer_floats(e) * (same_kv ? 0 : l);
        float* vc = e->vcache.p + kv_layer_floats(e) * (same_kv ? 0 : l);
        const int kcd = l == l_begin ? 0 : DEC_KC_DOWN;   // the first layer of the range starts from a materialised x
        // fused_in: this layer's q/k/v slabs (and the down slabs of x_new) were written by the previous layer's dec_qkv2 launch
        const bool fused_in = l > l_begin && (w.qkv2 != nullptr || w.q2h8 != nullptr) && !same_w;
        DecArgs a = e->da;
        a.first = l == l_begin ? 1 : 0;                     // the first kernel of a step stages the RoPE row ...
        a.inc_pos = (inc_pos && l == l_begin) ? 1 : 0;     // ... and advances the position word
        if (!(skip & 1) && !fused_in)
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 960.0, 576.0 * 960.0 * 4);
          if (w.qkv8) launch_dec_qkv(a, w.qkv8, w.qkv_f.KP / 8, kcd, s, w.qkv_sc);
          else launch_dec_qkv(a, w.qkv_f.p, w.qkv_f.KP / 8, kcd, s); }
        if (!(skip & 2))
        { ProfScope ps(e, PF_DECODE_ATTN, 4.0 * 64 * 9 * (double)B * (e->cur_pos + 1), 2.0 * (double)B * 3 * 64 * 4 * (e->cur_pos + 1));
          launch_dec_attn(e->da, kc, vc, fused_in, s); }
        if (!(skip & 4))
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 576.0, 576.0 * 576.0 * 4);
          if (w.o8) launch_dec_oproj(e->da, w.o8, s, w.o_sc);
          else launch_dec_oproj(e->da, w.o16, s); }
        if (!(skip & 8))
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 3072.0, 576.0 * 3072.0 * 4);
          if (w.gu8) launch_dec_gateup(e->da, w.gu8, s, w.gu_sc);
          else launch_dec_gateup(e->da, w.gu16, s); }
        const LMLayerW* nx = (l + 1 < l_end && !same_w) ? &e->layers[l + 1] : nullptr;
        if (nx && (nx->qkv2 || nx->q2h8)) {
            // the down projection of this layer and the q/k/v projection of the next one as one launch (decode.hip, dec_qkv2_kernel)
            if (!(skip & 16))
            { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * (2112.0 * 960.0 + 1536.0 * 576.0), (2112.0 * 960.0 + 1536.0 * 576.0) * (nx->q2h8 ? 1 : 4));
              if (nx->q2h8) launch_dec_qkv2_w8(e->da, nx->qkv8, nx->qkv_sc, nx->q2h8, nx->q2h_sc, w.dn8, w.dn_sc, s);
              else launch_dec_qkv2(e->da, nx->qkv2, w.down.p, s); }
        } else if (!(skip & 16))
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 1536.0, 576.0 * 1536.0 * 4);
          if (w.dn8) launch_dec_down(e->da, w.dn8, w.down.KP / 8, s, w.dn_sc);
          else launch_dec_down(e->da, w.down.p, w.down.KP / 8, s); }
    }
    return 0;
}
static int enqueue_decode_layers(mellow_engine* e, int B, const RecordArgs* rec) {
    CHK(enqueue_decode_layer_range(e, B, 0, e->cfg.num_layers, true));
    CHK(run_lm_head(e, B, DEC_KC_DOWN, rec));
    return 0;
}

extern "C" {

int mellow_debug_enable_taps(mellow_engine_t* e, int on) {
    if (!e) return fail("null engine");
    e->taps_on = on != 0;
    return 0;
}
int mellow_debug_tap(mellow_engine_t* e, const char* name, float* out, int64_t capacity, int64_t* numel) {
    if (!e || !name) return fail("null argument");
    auto it = e->tap_numel.find(name);
    if (it == e->tap_numel.end()) return fail("no such tap recorded: %s", name);
    if (numel) *numel = it->second;
    if (out) {
        if (capacity < it->second) return fail("tap buffer too small");
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipMemcpyAsync(out, e->taps[name].p, it->second * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return 0;
}

int mellow_logmel(mellow_engine_t* e, const float* wav, int n_clips, int64_t n_samples, int apply_bn, float* out) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!wav || !out) return fail("null argument");
    HIPCHK(hipSetDevice(e->device));
    CHK(run_encoder(e, wav, n_clips, n_samples, 1, apply_bn, out));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int mellow_encode(mellow_engine_t* e, const float* wav, int n_clips, int64_t n_samples, float* out) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!wav || !out) return fail("null argument");
    HIPCHK(hipSetDevice(e->device));
    CHK(run_encoder(e, wav, n_clips, n_samples, 0, 1, nullptr));
    launch_downsample33(e->proj33.p, n_clips, out, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// audio1|audio2 are separate caller buffers: stage them into one [2B][n] batch so the encoder runs ONE pass
// of 2B clips (the reference runs two passes of B, mellow.py:105-106)
static int encode_pair_to_prefix(mellow_engine* e, const float* a1, const float* a2, int64_t n_samples, const int32_t* ids,
                                 int B, float* prefix_out) {
    mellow_engine::Buf& cat = e->wavcat;
    CHK(ensure(e, cat, (size_t)2 * B * n_samples));
    HIPCHK(hipMemcpyAsync(cat.p, a1, (size_t)B * n_samples * 4, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(cat.p + (size_t)B * n_samples, a2, (size_t)B * n_samples * 4, hipMemcpyDeviceToDevice, e->stream));
    CHK(run_encoder(e, cat.p, 2 * B, n_samples, 0, 1, nullptr));
    { ProfScope ps(e, PF_MISC, 0, 0);
      launch_prefix_assemble(e->proj33.p, e->embed, ids, B, e->cfg.text_len, e->cfg.sep_token_id, e->cfg.vocab_size, prefix_out,
                             e->d_progress + 1, e->stream); }
    HIPCHK(hipGetLastError());
    return 0;
}
// The prompt ids are range-checked on the device (prefix_assemble_kernel sets word 1 of the mapped progress block); the host
// reads it once the stream is synchronised and fails like the reference's embedding lookup (IndexError in the Python binding).
static void clear_bad_id(mellow_engine* e) { __atomic_store_n(e->h_progress + 1, 0ull, __ATOMIC_RELEASE); }
static int check_bad_id(mellow_engine* e) {
    const unsigned long long w = __atomic_load_n(e->h_progress + 1, __ATOMIC_ACQUIRE);
    if (!w) return 0;
    return fail("index out of range in self: prompt id %d of example %u is outside the vocabulary [0, %d)", (int)(unsigned)(w & 0xffffffffu),
                (unsigned)((w >> 32) & 0x7fffffffu), e->cfg.vocab_size);
}

int mellow_prefix(mellow_engine_t* e, const float* audio1, const float* audio2, int64_t n_samples, const int32_t* input_ids,
                  int B, float* out) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!audio1 || !audio2 || !input_ids || !out) return fail("null argument");
    if (B <= 0) return fail("B must be positive");
    HIPCHK(hipSetDevice(e->device));
    clear_bad_id(e);
    CHK(encode_pair_to_prefix(e, audio1, audio2, n_samples, input_ids, B, out));
    HIPCHK(hipStreamSynchronize(e->stream));
    return check_bad_id(e);
}

int mellow_lm_prefill(mellow_engine_t* e, const float* prefix, int B, int T, int reserve, float* logits) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!prefix || B <= 0 || T <= 0 || reserve < 0) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    CHK(ensure_lm(e, B, T, T + reserve + 1));
    HIPCHK(hipMemcpyAsync(e->lm_x.p, prefix, (size_t)B * T * 576 * 4, hipMemcpyDeviceToDevice, e->stream));
    CHK(clear_page_tails(e, T, e->kv_Tmax));
    CHK(run_prefill(e, B, T, nullptr));
    if (logits)
        HIPCHK(hipMemcpyAsync(logits, e->dlogits.p, (size_t)B * e->cfg.vocab_size * 4, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int mellow_lm_decode_step(mellow_engine_t* e, const int32_t* token_ids, float* logits) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!token_ids) return fail("null argument");
    if (e->cur_B <= 0) return fail("decode step without a prefill");
    if (e->cur_pos + 1 > e->kv_Tmax) return fail("KV pages exhausted (reserve too small)");
    HIPCHK(hipSetDevice(e->device));
    const int B = e->cur_B;
    launch_dec_load_rows(e->da, B, e->embed, 576, token_ids, 0, e->cfg.vocab_size, e->stream);
    CHK(enqueue_decode_layers(e, B, nullptr));   // its first kernel advances the device position word
    e->cur_pos += 1;
    if (logits)
        HIPCHK(hipMemcpyAsync(logits, e->dlogits.p, (size_t)B * e->cfg.vocab_size * 4, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// Numeric tap of the decode step's lm_head kernel (dec_fullk_kernel) on caller-supplied rows: logits[B][vocab] = x[B][hidden] .
// lm_head^T with the engine's own head weights -- the e4m3 copy when the engine holds one (fp8 mode), and then act_fp8 selects
// whether the activations are quantised in the kernel (fp8 matrix pipe) or stay fp32.  x and logits are device buffers.
int mellow_debug_dec_head(mellow_engine_t* e, const float* x, int B, int act_fp8, float* logits) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!x || !logits || B <= 0 || B > 1024) return fail("bad argument");
    if (e->cfg.hidden_size != 576) return fail("the decode kernels are built for hidden size 576");
    HIPCHK(hipSetDevice(e->device));
    CHK(ensure_lm(e, B, 1, 2));
    DecArgs a = e->da;
    a.blk_live = nullptr; a.row_of_slot = nullptr;
    a.a8 = (act_fp8 && e->head8) ? 1 : 0;
    a.xnF = a.xmidF;                                  // dec_load_rows writes the F32-layout operand there
    launch_dec_load_rows(a, B, x, 576, nullptr, 1, 0, e->stream);
    if (e->head8) launch_dec_lm_head(a, e->head8, e->lm_head.KP / 8, e->cfg.vocab_size, e->stream, e->head_sc);
    else launch_dec_lm_head(a, e->lm_head.p, e->lm_head.KP / 8, e->cfg.vocab_size, e->stream);
    HIPCHK(hipMemcpyAsync(logits, e->dlogits.p, (size_t)B * e->cfg.vocab_size * 4, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    e->cur_B = 0;                                     // the decode state of an earlier prefill is gone
    return 0;
}

// lm.model.embed_tokens(ids) (reference decoder.py:47,64-66; wrapper.py:237): rows of the embedding table
int mellow_embed_tokens(mellow_engine_t* e, const int32_t* token_ids, int n, float* out) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!token_ids || !out || n <= 0) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    launch_gather_rows(e->embed, 576, token_ids, n, e->cfg.vocab_size, out, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// The decoder's forward over a whole embedded sequence (reference decoder.py:57-90 `self.lm(inputs_embeds=embedding_cat)`,
// reached from Mellow.forward mellow.py:89-98 -- the training-time forward): logits of EVERY position t >= from_pos, not only
// the last one.  embeds dev [B][T][hidden]; logits dev [B][T - from_pos][vocab].  All 30 layers run on all positions (the
// generation path's last-layer shortcut does not apply), then the final RMSNorm and the tied lm_head as one GEMM on the
// exact fp32 kernel (in every precision mode: the head is not part of the split / fp8 GEMM set).
int mellow_lm_forward_logits(mellow_engine_t* e, const float* embeds, int B, int T, int from_pos, float* logits) {
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!embeds || !logits || B <= 0 || T <= 0 || from_pos < 0 || from_pos >= T) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    CHK(ensure_lm(e, B, T, T + 1));
    HIPCHK(hipMemcpyAsync(e->lm_x.p, embeds, (size_t)B * T * 576 * 4, hipMemcpyDeviceToDevice, s));
    CHK(run_prefill(e, B, T, nullptr, true));
    e->cur_B = 0;                                   // no decode state: a decode step needs a real prefill first
    const int n = T - from_pos;
    // final norm on the selected rows only: gather [B][n][576] out of [B][T][576] into lm_xn, then normalise in place
    launch_gather_span(e->lm_x.p, B, T, from_pos, n, e->lm_o.p, s);
    launch_rmsnorm(e->lm_o.p, e->lm_xn.p, B * n, 576, e->final_norm, e->cfg.rms_norm_eps, s);
    GemmArgs g = lin(e->lm_xn.p, 576, B * n, e->lm_head, logits, e->cfg.vocab_size, nullptr);
    launch_gemm(g, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// A0 on the device: torchaudio-style sinc_interp_hann resampling (lowpass_filter_width 6, rolloff 0.99), the polyphase bank
// built exactly like mellow_amd/audio.py::_sinc_resample_kernel (float64, then cast to float32)
int mellow_resample(mellow_engine_t* e, const float* wav, int n_clips, int64_t n_in, int orig_freq, int new_freq, float* out,
                    int64_t out_capacity, int64_t* n_out) {
    if (!e || !wav || n_clips <= 0 || n_in <= 0 || orig_freq <= 0 || new_freq <= 0) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    int a = orig_freq, b = new_freq;
    while (b) { const int t = a % b; a = b; b = t; }
    const int orig = orig_freq / a, nw = new_freq / a;
    const int64_t target = (int64_t)((nw * n_in + orig - 1) / orig);       // ceil(new * length / orig)
    if (n_out) *n_out = target;
    if (!out) return 0;
    if (out_capacity < target) return fail("resample output buffer too small");
    const double PI = 3.14159265358979323846, lpw = 6.0, rolloff = 0.99;
    const double base_freq = (orig < nw ? orig : nw) * rolloff;
    const int width = (int)std::ceil(lpw * orig / base_freq);
    const int klen = 2 * width + orig;
    float* dw = nullptr;
    auto it = e->resample_banks.find({orig, nw});
    if (it != e->resample_banks.end()) {
        dw = it->second;
    } else {    // built once per rate pair and kept (no allocation / host filter design on later calls)
        std::vector<float> wT((size_t)klen * nw);
        const double scale = base_freq / orig;
        for (int p = 0; p < nw; ++p)
            for (int k = 0; k < klen; ++k) {
                double t = (double)(-p) / nw + (double)(k - width) / orig;
                t *= base_freq;
                if (t < -lpw) t = -lpw;
                if (t > lpw) t = lpw;
                const double c = std::cos(t * PI / lpw / 2.0);
                const double window = c * c;
                t *= PI;
                const double kern = t == 0.0 ? 1.0 : std::sin(t) / t;
                wT[(size_t)k * nw + p] = (float)(kern * window * scale);
            }
        HIPCHK(hipMalloc(&dw, wT.size() * sizeof(float)));
        e->allocs.push_back(dw);
        HIPCHK(hipMemcpy(dw, wT.data(), wT.size() * sizeof(float), hipMemcpyHostToDevice));
        e->resample_banks[{orig, nw}] = dw;
    }
    // rows of `out` are `target` long: the kernel writes with stride n_out = target
    launch_resample(wav, n_clips, n_in, dw, orig, nw, klen, width, out, target, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int mellow_argmax(mellow_engine_t* e, const float* logits, int B, int32_t* tokens) {
    if (!e || !logits || !tokens || B <= 0) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    launch_argmax(logits, B, e->cfg.vocab_size, e->cfg.vocab_size, tokens, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

}  // extern "C"

extern "C" {

// Wait (without touching the stream) until the arg-max kernel has published ticket >= want; *nseen = rows stopped so far.
static int wait_ticket(mellow_engine* e, unsigned want, unsigned* nseen) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        const unsigned long long v = __atomic_load_n(e->h_progress, __ATOMIC_ACQUIRE);
        if ((unsigned)(v >> 32) >= want) {
            if (nseen) *nseen = (unsigned)(v & 0xffffffffu);
            return 0;
        }
        if ((spins & 0x3ff) == 0) {
            const hipError_t q = hipStreamQuery(e->stream);
            if (q == hipSuccess) {      // nothing left in flight: the ticket must be there now
                const unsigned long long v2 = __atomic_load_n(e->h_progress, __ATOMIC_ACQUIRE);
                if ((unsigned)(v2 >> 32) >= want) continue;
                return fail("decode progress word stalled at ticket %u (wanted %u) with an idle stream", (unsigned)(v2 >> 32), want);
            }
            if (q != hipErrorNotReady) return fail("stream error while waiting for a decode step: %s", hipGetErrorString(q));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                return fail("timed out waiting for decode ticket %u", want);
        }
        // spin politely: a pause per poll, and after ~50 us of spinning yield the core between polls (EnginePool runs one
        // such loop per context thread)
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("yield");
#endif
        if (spins > 4096) std::this_thread::yield();
    }
}

static int generate_pass(mellow_engine_t* e, const float* audio1, const float* audio2, int64_t n_samples,
                         const int32_t* input_ids, int B, int max_len, int stop_id,
                         int ignore_stop, int32_t* out_tokens, int32_t* out_len, int32_t* out_steps, float* first_token_ms);
// The reference's loop (wrapper.py:216-249) takes any number of examples.  One pass of the engine takes up to 1024 rows (32 row
// blocks of loop state), so a larger batch runs as consecutive passes of <= 1024 rows on the same pages: examples are
// independent, the token record of every pass lands at its rows of `out_tokens`, a pass that stopped before the longest one is
// padded with -1 (never computed), and the reference's stop rule -- the loop ends at the first step at which EVERY row has
// produced the stop id -- is the maximum over the passes (a row's own length never depends on other rows).
int mellow_generate(mellow_engine_t* e, const float* audio1, const float* audio2, int64_t n_samples,
                    const int32_t* input_ids, int B, int max_len, float top_p, float temperature, int stop_id,
                    int ignore_stop, int32_t* out_tokens, int32_t* out_len, int32_t* out_steps, float* first_token_ms) {
    (void)top_p;
    (void)temperature;  // the reference's top-p/temperature path never changes the arg-max (wrapper.py:219-232)
    if (!e || !e->finalized) return fail("engine not finalized");
    if (!audio1 || !audio2 || !input_ids || !out_tokens) return fail("null argument");
    if (B <= 0 || max_len <= 0) return fail("B and max_len must be positive");
    constexpr int kPassRows = 1024;
    if (B <= kPassRows)
        return generate_pass(e, audio1, audio2, n_samples, input_ids, B, max_len, stop_id, ignore_stop, out_tokens, out_len, out_steps, first_token_ms);
    int steps_all = 0, enq_all = 0, rep_all = 0;
    float ph[3] = {0.f, 0.f, 0.f};
    std::vector<int> pass_steps;
    for (int r0 = 0; r0 < B; r0 += kPassRows) {
        const int nb = B - r0 < kPassRows ? B - r0 : kPassRows;
        int st = 0;
        float ftm = 0.f;
        CHK(generate_pass(e, audio1 + (size_t)r0 * n_samples, audio2 + (size_t)r0 * n_samples, n_samples, input_ids + (size_t)r0 * e->cfg.text_len,
                          nb, max_len, stop_id, ignore_stop, out_tokens + (size_t)r0 * max_len, out_len ? out_len + r0 : nullptr, &st, &ftm));
        if (r0 == 0 && first_token_ms) *first_token_ms = ftm;      // the first answers of the call: entry -> first token of the first pass
        pass_steps.push_back(st);
        steps_all = st > steps_all ? st : steps_all;
        enq_all = e->last_steps_enqueued > enq_all ? e->last_steps_enqueued : enq_all;
        rep_all += e->last_compactions;
        for (int i = 0; i < 3; ++i) ph[i] += e->phase_ms[i];
    }
    // columns a pass never reached (it stopped before the longest pass): -1, like the rows of a block that stopped early
    for (size_t p = 0; p < pass_steps.size(); ++p) {
        const int r0 = (int)p * kPassRows, nb = B - r0 < kPassRows ? B - r0 : kPassRows;
        if (pass_steps[p] >= steps_all) continue;
        int32_t* dst = out_tokens + (size_t)r0 * max_len + pass_steps[p];
        const size_t w = (size_t)(steps_all - pass_steps[p]) * sizeof(int32_t);
        hipPointerAttribute_t at;
        const bool on_device = hipPointerGetAttributes(&at, out_tokens) == hipSuccess && at.type == hipMemoryTypeDevice;
        if (!on_device) (void)hipGetLastError();            // a plain host pointer is not an error here
        if (on_device) HIPCHK(hipMemset2D(dst, (size_t)max_len * sizeof(int32_t), 0xff, w, nb));
        else for (int r = 0; r < nb; ++r) memset(dst + (size_t)r * max_len, 0xff, w);
    }
    e->last_steps_enqueued = enq_all;
    e->last_compactions = rep_all;
    for (int i = 0; i < 3; ++i) e->phase_ms[i] = ph[i];
    if (out_steps) *out_steps = steps_all;
    return 0;
}

static int generate_pass(mellow_engine_t* e, const float* audio1, const float* audio2, int64_t n_samples,
                         const int32_t* input_ids, int B, int max_len, int stop_id,
                         int ignore_stop, int32_t* out_tokens, int32_t* out_len, int32_t* out_steps, float* first_token_ms) {
    const auto t_entry = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int T = e->cfg.prefix_len;
    // KV page geometry in buckets of 64 positions, so that nearby max_len values share pages, key split and graphs
    int Tmax = rup(T + max_len, 64);
    if (Tmax > e->cfg.max_positions) Tmax = T + max_len;
    CHK(ensure_lm(e, B, T, Tmax, T + max_len));
    const int Bp = e->da.rows;
    CHK(ensure(e, e->out_tok, (size_t)Bp * max_len));
    HIPCHK(hipEventRecord(e->ev_phase[0], s));
    // loop state (the prefill's arg-max already records token 0 and publishes ticket 1)
    __atomic_store_n(e->h_progress, 0ull, __ATOMIC_RELEASE);
    clear_bad_id(e);
    HIPCHK(hipMemsetAsync(e->d_nseen, 0, 3 * sizeof(int32_t), s));       // n_seen, arrive, ticket
    HIPCHK(hipMemsetAsync(e->d_seen, 0, 1024 * sizeof(int32_t), s));
    e->h_params[0] = max_len;
    e->h_params[1] = stop_id;
    HIPCHK(hipMemcpyAsync(e->d_params, e->h_params, 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // Per-row-block early exit (reference stop rule, more than one 32-row block): once every row of a block has produced the
    // stop id, the block's workgroups return at once in every later kernel (its rows' texts are already cut there).  Columns a
    // row never reached are -1 in the token record.
    e->da.logits = nullptr;             // generation needs the arg-max candidates only: no 6 MB logits store per step
    e->da.blk_live = (!ignore_stop && e->da.RB > 1) ? e->d_blk_live : nullptr;
    e->da.blk_snap = e->d_blk_live + 32;
#ifdef MELLOW_DEVPROBE
    static const bool dev_dead = getenv("MELLOW_DEV_DEAD_BLOCKS") != nullptr;    // developer probe: launch-chain floor of a step
#else
    constexpr bool dev_dead = false;
#endif
    static const bool no_migrate = getenv("MELLOW_NO_ROW_MIGRATION") != nullptr;   // developer A/B: block exit without repacking
    e->da.row_of_slot = nullptr;
    if (dev_dead) {
        e->da.blk_live = e->d_blk_live;
        HIPCHK(hipMemsetAsync(e->d_blk_left, 0, 96 * sizeof(int32_t), s));
    } else if (e->da.blk_live) {
        if (!no_migrate && B <= 1024) {
            std::vector<int32_t> ident(1024);
            for (int i = 0; i < 1024; ++i) ident[i] = i < B ? i : -1;
            e->h_ident = ident;       // kept alive until the copy has run
            HIPCHK(hipMemcpyAsync(e->d_row_of_slot, e->h_ident.data(), 1024 * sizeof(int32_t), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemsetAsync(e->d_ncompact, 0, sizeof(int32_t), s));
            e->da.row_of_slot = e->d_row_of_slot;
        }
        for (int rb = 0; rb < 32; ++rb) {
            const int left = B - 32 * rb;
            e->h_blk[rb] = left <= 0 ? 0 : (left > 32 ? 32 : left);
            e->h_blk[32 + rb] = left > 0 ? 1 : 0;
        }
        HIPCHK(hipMemcpyAsync(e->d_blk_left, e->h_blk, 64 * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(e->out_tok.p, 0xff, (size_t)Bp * max_len * sizeof(int32_t), s));
    }
    CHK(clear_page_tails(e, T, e->kv_Tmax));     // everything a key-group load can touch (whole chunks are loaded, then masked)
    CHK(encode_pair_to_prefix(e, audio1, audio2, n_samples, input_ids, B, e->lm_x.p));
    HIPCHK(hipEventRecord(e->ev_phase[1], s));
    RecordArgs rec;
    rec.embed_next = true;
    CHK(run_prefill(e, B, T, &rec));
    HIPCHK(hipEventRecord(e->ev_phase[2], s));

    // one decode step = 30 x (qkv | attention | o_proj | gate/up | down) + final norm + lm_head + arg-max/record/embed,
    // captured once per (B, page geometry, buffers) and replayed; max_len and the stop id are read from d_params
    const bool graph = e->use_graph && !e->prof_on && max_len > 1;
    if (graph && (!e->step_exec || e->step_exec_B != B || e->step_exec_Tmax != e->kv_Tmax || e->graph_out_tok != e->out_tok.p ||
                  e->graph_blk != e->da.blk_live || e->graph_rows != e->da.row_of_slot)) {
        if (e->step_exec) { hipGraphExecDestroy(e->step_exec); e->step_exec = nullptr; }
        if (e->step_exec8) { hipGraphExecDestroy(e->step_exec8); e->step_exec8 = nullptr; }
        hipGraph_t gr = nullptr;
        HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_decode_layers(e, B, &rec);
        hipError_t ce = hipStreamEndCapture(s, &gr);
        if (rc) return rc;
        if (ce != hipSuccess) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(ce));
        HIPCHK(hipGraphInstantiate(&e->step_exec, gr, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(gr));
        // eight consecutive steps as ONE graph: the step reads its position from the device word, so a replay of the
        // same kernel sequence IS the next step; one launch per 8 steps removes the host/CP hand-over between graphs
        HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc8 = 0;
        for (int k = 0; k < 8 && !rc8; ++k) rc8 = enqueue_decode_layers(e, B, &rec);
        hipError_t ce8 = hipStreamEndCapture(s, &gr);
        if (rc8) return rc8;
        if (ce8 != hipSuccess) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(ce8));
        HIPCHK(hipGraphInstantiate(&e->step_exec8, gr, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(gr));
        e->step_exec_B = B; e->step_exec_Tmax = e->kv_Tmax; e->graph_out_tok = e->out_tok.p; e->graph_blk = e->da.blk_live; e->graph_rows = e->da.row_of_slot;
    }
    int steps_done = 1;   // token 0 came from the prefill
    double first_ms = -1.0;
    auto note_first = [&]() {
        if (first_ms < 0) first_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
    };
    if (ignore_stop) {
        // fixed-length mode: nothing to decide on the host, everything is enqueued at once
        for (int i = 1; i < max_len;) {
            const bool eight = graph && i + 8 <= max_len;
            if (eight) HIPCHK(hipGraphLaunch(e->step_exec8, s));
            else if (graph) HIPCHK(hipGraphLaunch(e->step_exec, s));
            else CHK(enqueue_decode_layers(e, B, &rec));
            i += eight ? 8 : 1;
            e->cur_pos += eight ? 8 : 1;
            steps_done = i;
        }
        CHK(wait_ticket(e, 1, nullptr));
        note_first();
    } else {
        // reference stop rule (wrapper.py:247-249): the loop ends after the first step at which every row has produced the
        // stop id at least once.  The arg-max kernel publishes (step ticket, rows stopped) to a host-visible word, so the
        // host follows the rule one step behind the device without synchronising: step i+1 is enqueued while step i runs,
        // and at most ONE step is ever enqueued past the deciding one.
        for (int i = 1; i < max_len; ++i) {
            if (graph) HIPCHK(hipGraphLaunch(e->step_exec, s));
            else CHK(enqueue_decode_layers(e, B, &rec));
            e->cur_pos += 1;
            steps_done = i + 1;
            unsigned nseen = 0;
            CHK(wait_ticket(e, (unsigned)i, &nseen));      // ticket i = the arg-max of step index i-1 is complete
            note_first();
            if ((int)nseen >= B) break;
        }
        if (first_ms < 0) { CHK(wait_ticket(e, 1, nullptr)); note_first(); }
    }
    HIPCHK(hipEventRecord(e->ev_phase[3], s));
    HIPCHK(hipGetLastError());
    // host-side length bookkeeping (reference wrapper.py:247-254) on the engine-owned record
    std::vector<int32_t> toks((size_t)B * max_len);
    HIPCHK(hipMemcpyAsync(out_tokens, e->out_tok.p, toks.size() * sizeof(int32_t), hipMemcpyDefault, s));
    HIPCHK(hipMemcpyAsync(toks.data(), e->out_tok.p, toks.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    CHK(check_bad_id(e));        // a prompt id outside the vocabulary (flagged by prefix_assemble_kernel): the reference raises IndexError
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventElapsedTime(&e->phase_ms[i], e->ev_phase[i], e->ev_phase[i + 1]));
    if (first_token_ms) *first_token_ms = (float)first_ms;
    int ref_steps = steps_done;
    if (!ignore_stop) {
        // the reference stops after the first step at which every row has produced stop_id at least once
        std::vector<char> seen(B, 0);
        int nseen = 0;
        for (int st = 0; st < steps_done; ++st) {
            for (int b = 0; b < B; ++b)
                if (!seen[b] && toks[(size_t)b * max_len + st] == stop_id) { seen[b] = 1; ++nseen; }
            if (nseen == B) { ref_steps = st + 1; break; }
        }
    }
    e->last_steps_enqueued = steps_done;
    e->cur_B = 0;      // the decode state of a generate call (no logits store, early-exit words) is not a base for the step taps:
                       // mellow_lm_decode_step needs a mellow_lm_prefill of its own
    e->last_compactions = 0;
    if (e->da.row_of_slot) HIPCHK(hipMemcpy(&e->last_compactions, e->d_ncompact, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_steps) *out_steps = ref_steps;
    if (out_len)
        for (int b = 0; b < B; ++b) {
            int n = ref_steps;
            for (int st = 0; st < ref_steps; ++st)
                if (toks[(size_t)b * max_len + st] == stop_id) { n = st; break; }
            out_len[b] = n;
        }
    return 0;
}

int mellow_last_steps_enqueued(mellow_engine_t* e) { return e ? e->last_steps_enqueued : -1; }
int mellow_last_row_repacks(mellow_engine_t* e) { return e ? e->last_compactions : -1; }
int mellow_stft_is_fft(mellow_engine_t* e) { return e && e->fft_win ? 1 : 0; }

int mellow_prof_enable(mellow_engine_t* e, int on) {
    if (!e) return fail("null engine");
    e->prof_on = on != 0;
    return 0;
}
int mellow_prof_reset(mellow_engine_t* e) {
    if (!e) return fail("null engine");
    hipStreamSynchronize(e->stream);
    e->prof.clear();
    e->ev_used = 0;
    return 0;
}
int mellow_prof_get(mellow_engine_t* e, int i, int64_t* launches, double* ms, double* flops, double* bytes) {
    if (!e || i < 0 || i >= PF_COUNT) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    int64_t n = 0;
    double t = 0, f = 0, by = 0;
    for (const auto& r : e->prof)
        if (r.fam == i) {
            float m = 0.f;
            HIPCHK(hipEventElapsedTime(&m, r.a, r.b));
            t += m; f += r.flops; by += r.bytes; ++n;
        }
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (bytes) *bytes = by;
    return 0;
}
int mellow_last_phase_ms(mellow_engine_t* e, float* encode_ms, float* prefill_ms, float* decode_ms) {
    if (!e) return fail("null engine");
    if (encode_ms) *encode_ms = e->phase_ms[0];
    if (prefill_ms) *prefill_ms = e->phase_ms[1];
    if (decode_ms) *decode_ms = e->phase_ms[2];
    return 0;
}
// developer instrumentation (not part of the public header): time `iters` launches of one plain GEMM shape on
// synthetic device buffers (garbage-in; EPI_LINEAR, no bias) -> average milliseconds per launch
int mellow_dev_gemm_time(mellow_engine_t* e, int M, int N, int K, int iters, float* ms_out) {
    if (!e || M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4 || iters <= 0 || !ms_out) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    float *A = nullptr, *W = nullptr, *Cc = nullptr;
    const size_t NP = (size_t)rup(N, 128);
    HIPCHK(hipMalloc(&A, (size_t)M * K * 4));
    HIPCHK(hipMalloc(&W, NP * K * 4));
    HIPCHK(hipMalloc(&Cc, (size_t)M * N * 4));
    HIPCHK(hipMemsetAsync(A, 0x3c, (size_t)M * K * 4, e->stream));      // 0x3c3c3c3c = 0.0115 (finite, non-zero)
    HIPCHK(hipMemsetAsync(W, 0x3c, NP * K * 4, e->stream));
    GemmArgs g;
    g.A = A; g.lda = K; g.M = M; g.K = K; g.Wp = W; g.Nw = N; g.N = N; g.C = Cc; g.ldc = N;
    launch_gemm(g, e->stream);
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, e->stream));
    for (int i = 0; i < iters; ++i) launch_gemm(g, e->stream);
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / iters;
    hipEventDestroy(a); hipEventDestroy(b);
    hipFree(A); hipFree(W); hipFree(Cc);
    return 0;
}
// one fp32 GEMM C[M][N] = A[M][K] . W[N][K]^T on host data through the exact fp32 MFMA kernel (mode 0) or the bf16x3 split
// kernel with 6 / 9 partial products (mode 6 / 9): the accuracy tap of include/mellow_hip.h
int mellow_debug_gemm_f32(mellow_engine_t* e, int mode, const float* A, int M, int K, const float* W, int N, float* C_out,
                          int iters, float* ms2) {
    if (!e || !A || !W || M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4) return fail("bad argument");
    if (mode != 0 && mode != 6 && mode != 9 && mode != 16 && mode != 17) return fail("mode must be 0, 6, 9, 16 (fused 6-term) or 17 (pre-split A, LDS-DMA)");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int NP = rup(N, 128);
    float *dA = nullptr, *dW = nullptr, *dWp = nullptr, *dC = nullptr;
    void *dA3 = nullptr, *dPB = nullptr;
    HIPCHK(hipMalloc(&dA, (size_t)M * K * 4));
    HIPCHK(hipMalloc(&dW, (size_t)N * K * 4));
    HIPCHK(hipMalloc(&dWp, (size_t)NP * K * 4));
    HIPCHK(hipMalloc(&dC, (size_t)M * N * 4));
    HIPCHK(hipMalloc(&dA3, (size_t)rup(M, 128) * K * 6));
    HIPCHK(hipMalloc(&dPB, (size_t)NP * K * 6));
    HIPCHK(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dW, W, (size_t)N * K * 4, hipMemcpyHostToDevice));
    launch_pack_weight(dW, N, K, K, dWp, NP, K, s);
    launch_pack_bf16x3(dWp, NP, K, dPB, s);
    GemmArgs g;
    g.A = dA; g.lda = K; g.M = M; g.K = K; g.Wp = dWp; g.Nw = N; g.N = N; g.C = dC; g.ldc = N;
    g.A8 = reinterpret_cast<const uint8_t*>(dA3); g.lda8 = (int64_t)3 * (K >> 3); g.W8 = reinterpret_cast<const uint8_t*>(dPB);
    auto run = [&](bool pre, bool main) {
        if (mode == 0) { if (main) launch_gemm(g, s); }
        else if (mode == 16) { if (main) launch_gemm_bf16x3_fused(g, s); }
        else if (mode == 17) { if (pre) launch_split_rows_apb(dA, K, M, K, dA3, s); if (main) launch_gemm_bf16x3_apb(g, s); }
        else { if (pre) launch_split_rows(dA, K, M, K, dA3, s); if (main) launch_gemm_bf16x3(g, mode, s); }
    };
    run(true, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    if (C_out) HIPCHK(hipMemcpy(C_out, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    if (ms2 && iters > 0) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        float m0 = 0.f, m1 = 0.f;
        HIPCHK(hipEventRecord(a, s));
        for (int i = 0; i < iters; ++i) run(true, false);
        HIPCHK(hipEventRecord(b, s));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&m0, a, b));
        HIPCHK(hipEventRecord(a, s));
        for (int i = 0; i < iters; ++i) run(false, true);
        HIPCHK(hipEventRecord(b, s));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&m1, a, b));
        ms2[0] = m0 / iters;
        ms2[1] = m1 / iters;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    hipFree(dA); hipFree(dW); hipFree(dWp); hipFree(dC); hipFree(dA3); hipFree(dPB);
    return 0;
}
// one fp8 GEMM C[M][N] = A[M][K] . W[N][K]^T on host data (quantise rows, pack + quantise weight, fp8 MFMA GEMM,
// plain epilogue) and, optionally, its average time: the quantisation parity tap of include/mellow_hip.h
int mellow_debug_gemm_fp8(mellow_engine_t* e, const float* A, int M, int K, const float* W, int N, float* C_out, int iters,
                        float* ms_out) {
    if (!e || !A || !W || M <= 0 || N <= 0 || K <= 0 || K % 64 || N % 4) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int NP = rup(N, 128);
    float *dA = nullptr, *dW = nullptr, *dWp = nullptr, *dC = nullptr, *dsa = nullptr, *dsw = nullptr;
    uint8_t *dA8 = nullptr, *dW8 = nullptr;
    HIPCHK(hipMalloc(&dA, (size_t)M * K * 4));
    HIPCHK(hipMalloc(&dW, (size_t)N * K * 4));
    HIPCHK(hipMalloc(&dWp, (size_t)NP * K * 4));
    HIPCHK(hipMalloc(&dC, (size_t)M * N * 4));
    HIPCHK(hipMalloc(&dsa, (size_t)M * 4));
    HIPCHK(hipMalloc(&dsw, (size_t)NP * 4));
    HIPCHK(hipMalloc(&dA8, (size_t)M * K));
    HIPCHK(hipMalloc(&dW8, (size_t)NP * K));
    HIPCHK(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dW, W, (size_t)N * K * 4, hipMemcpyHostToDevice));
    launch_pack_weight(dW, N, K, K, dWp, NP, K, s);
    launch_pack_fp8(dWp, NP, K, dW8, dsw, s);
    GemmArgs g;
    g.A = dA; g.lda = K; g.M = M; g.K = K; g.Wp = dWp; g.Nw = N; g.N = N; g.C = dC; g.ldc = N;
    g.A8 = dA8; g.lda8 = K; g.a_scale = dsa; g.W8 = dW8; g.w_scale = dsw;
    launch_quant_rows(dA, K, M, K, dA8, K, dsa, s);
    launch_gemm_fp8(g, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    if (C_out) HIPCHK(hipMemcpy(C_out, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    if (ms_out && iters > 0) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        float ms_q = 0.f, ms_g = 0.f;
        HIPCHK(hipEventRecord(a, s));
        for (int i = 0; i < iters; ++i) launch_quant_rows(dA, K, M, K, dA8, K, dsa, s);
        HIPCHK(hipEventRecord(b, s));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms_q, a, b));
        HIPCHK(hipEventRecord(a, s));
        for (int i = 0; i < iters; ++i) launch_gemm_fp8(g, s);
        HIPCHK(hipEventRecord(b, s));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms_g, a, b));
        ms_out[0] = ms_q / iters;
        ms_out[1] = ms_g / iters;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    hipFree(dA); hipFree(dW); hipFree(dWp); hipFree(dC); hipFree(dsa); hipFree(dsw); hipFree(dA8); hipFree(dW8);
    return 0;
}
// developer instrumentation (not part of the public header): one CSV line per profiled launch
int mellow_dev_prof_dump(mellow_engine_t* e, const char* path) {
    if (!e || !path) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    FILE* f = fopen(path, "w");
    if (!f) return fail("cannot open %s", path);
    fprintf(f, "fam,M,N,K,epi,ms,flops\n");
    for (const auto& r : e->prof) {
        float m = 0.f;
        hipEventElapsedTime(&m, r.a, r.b);
        fprintf(f, "%d,%d,%d,%d,%d,%.6f,%.0f\n", r.fam, r.M, r.N, r.K, r.epi, m, r.flops);
    }
    fclose(f);
    return 0;
}
// developer instrumentation (not part of the public header): s_memtime stamps of workgroup 0 of the decode kernels
int mellow_dev_kdebug(mellow_engine_t* e, int on, uint64_t* host_out64) {
    if (!e) return fail("null engine");
    static uint64_t* buf = nullptr;
    HIPCHK(hipSetDevice(e->device));
    if (on) {
        if (!buf) HIPCHK(hipMalloc(&buf, 64 * sizeof(uint64_t)));
        HIPCHK(hipMemset(buf, 0, 64 * sizeof(uint64_t)));
        set_kernel_debug_buffer(buf);
        set_gemm_debug_buffer(buf);
    } else {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (buf && host_out64) HIPCHK(hipMemcpy(host_out64, buf, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        set_kernel_debug_buffer(nullptr);
        set_gemm_debug_buffer(nullptr);
    }
    return 0;
}
int mellow_engine_set_precision(mellow_engine_t* e, int mode) {
    if (!e) return fail("null engine");
    if (e->finalized) return fail("precision must be chosen before mellow_engine_finalize");
    if (mode != MELLOW_PRECISION_F32 && mode != MELLOW_PRECISION_FP8 && mode != MELLOW_PRECISION_F32X3)
        return fail("unknown precision mode %d", mode);
    e->fp8 = mode == MELLOW_PRECISION_FP8;
    // developer / test knobs of the fp8 mode (read here, per engine): which half of the path reads e4m3 weights
    const char *fd = getenv("MELLOW_FP8_DECODE"), *fp = getenv("MELLOW_FP8_PREFILL");
    e->fp8_decode = e->fp8 && !(fd && fd[0] == '0');
    const char* fa = getenv("MELLOW_FP8_DECODE_ACT");
    e->fp8_decode_act = e->fp8_decode && !(fa && fa[0] == '0');
    e->fp8_prefill = !(fp && fp[0] == '0');
    e->f32x3_terms = 0;
    if (mode == MELLOW_PRECISION_F32X3) {
        // six partial products (a2*b3, a3*b2, a3*b3 dropped: < 2^-23 |a*b| in total); measured error against an fp64
        // product is identical to the nine-term form and slightly below the fp32 MFMA kernel's (tools/f32x3_check.py).
        // Developer knob: MELLOW_F32X3_TERMS=9 keeps every partial product.
        const char* t = getenv("MELLOW_F32X3_TERMS");
        e->f32x3_terms = (t && atoi(t) == 9) ? 9 : 6;
    }
    return 0;
}
// A second execution context on the same device that SHARES the parent's weights (read-only after finalize): own HIP stream,
// own workspaces, KV pages, decode buffers, captured graphs and loop words.  Calls on the two handles may overlap from
// different host threads (mellow_amd/serve.py).  The parent must outlive its forks.
int mellow_engine_fork(mellow_engine_t* parent, mellow_engine_t** out) {
    if (!parent || !out) return fail("null argument");
    if (!parent->finalized) return fail("fork needs a finalized engine");
    HIPCHK(hipSetDevice(parent->device));
    HIPCHK(hipStreamSynchronize(parent->stream));
    mellow_engine* c = new mellow_engine();
    c->cfg = parent->cfg; c->device = parent->device; c->finalized = true; c->owns_weights = false; c->use_graph = parent->use_graph;
    c->fp8 = parent->fp8; c->fp8_decode = parent->fp8_decode; c->fp8_decode_act = parent->fp8_decode_act; c->fp8_prefill = parent->fp8_prefill; c->f32x3_terms = parent->f32x3_terms;
    // weight pointers (device memory owned by the parent)
    c->dft = parent->dft; c->mel = parent->mel; c->fft_win = parent->fft_win; c->fft_tw1 = parent->fft_tw1; c->fft_tw2 = parent->fft_tw2;
    c->bn_alpha = parent->bn_alpha; c->bn_beta = parent->bn_beta;
    c->pe_w = parent->pe_w; c->pe_b = parent->pe_b; c->pe_nw = parent->pe_nw; c->pe_nb = parent->pe_nb;
    for (int i = 0; i < 4; ++i) c->blocks[i] = parent->blocks[i];
    for (int i = 0; i < 3; ++i) c->merge[i] = parent->merge[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) c->win_map[i][j] = parent->win_map[i][j];
    c->fn_w = parent->fn_w; c->fn_b = parent->fn_b;
    c->tscam = parent->tscam; c->c2l = parent->c2l; c->lin1 = parent->lin1; c->lin2 = parent->lin2;
    c->tscam_b = parent->tscam_b; c->c2l_b = parent->c2l_b; c->pln_w = parent->pln_w; c->pln_b = parent->pln_b;
    c->emb_row_map = parent->emb_row_map;
    c->embed = parent->embed; c->lm_head = parent->lm_head; c->layers = parent->layers; c->final_norm = parent->final_norm;
    c->rope_cos = parent->rope_cos; c->rope_sin = parent->rope_sin;
    c->head8 = parent->head8; c->head_sc = parent->head_sc;
    c->bf_w = parent->bf_w; c->fp8_w = parent->fp8_w;
    c->resample_banks = parent->resample_banks;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // Contexts that pipeline whole batches do not split their prefill: their overlap comes from each other, and HIP maps streams
    // onto a handful of hardware queues -- with extra streams per context two contexts' main streams end up on ONE queue and
    // serialise (measured: `pipelined` 543 -> 450 responses/s).  The parent stops splitting from its first fork on.
    c->prefill_parts = 1;
    parent->prefill_parts = 1;
    for (int i = 0; i < 3; ++i)
        if (parent->stream2[i]) { hipStreamDestroy(parent->stream2[i]); parent->stream2[i] = nullptr; }
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&c->ev_phase[i]));
    CHK(alloc_state_words(c));
    *out = c;
    return 0;
}

int mellow_set_graph(mellow_engine_t* e, int on) {
    if (!e) return fail("null engine");
    e->use_graph = on != 0;
    return 0;
}

}  // extern "C"
