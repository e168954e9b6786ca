// Host side of libmellow_hip.so: the C ABI of include/mellow_hip.h.
// Owns weights (re-tiled into MFMA fragment order at load), KV pages, workspaces, the private HIP
// stream, the captured decode-step hipGraph and the event-based profiler.
//
// Reference seams replaced (soham97/mellow):
//   model construction + load_state_dict + .to(cuda)      wrapper.py:59-88      -> create/load/finalize
//   Mellow.generate_prefix_inference                      mellow.py:100-108     -> run_encoder + prefix
//   MellowWrapper._generate_batch                         wrapper.py:197-249    -> mellow_generate
#include "engine_internal.h"

static thread_local std::string g_err;
int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
// ---- small helpers ---------------------------------------------------------------------------------------------
int ensure(mellow_engine* e, mellow_engine::Buf& b, size_t floats) {
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
int dev_alloc(mellow_engine* e, float** out, size_t floats) {
    const size_t bytes = (floats * sizeof(float) + 4095) / 4096 * 4096;
    if (!e->arena) {
        const size_t want = (size_t)(e->arena_mb < 0 ? 0 : e->arena_mb) << 20;        // option "arena_mb" (default 3400)
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
int upload(mellow_engine* e, float** out, const float* src, size_t floats, size_t alloc_floats) {
    if (alloc_floats < floats) alloc_floats = floats;
    CHK(dev_alloc(e, out, alloc_floats));
    if (alloc_floats > floats) HIPCHK(hipMemsetAsync(*out, 0, alloc_floats * sizeof(float), e->stream));
    HIPCHK(hipMemcpyAsync(*out, src, floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
hipEvent_t next_event(mellow_engine* e) {
    if (e->ev_used == e->ev_pool.size()) {
        hipEvent_t ev;
        hipEventCreate(&ev);
        e->ev_pool.push_back(ev);
    }
    return e->ev_pool[e->ev_used++];
}
int tap(mellow_engine* e, const char* name, const float* src, int64_t n) {
    if (!e->taps_on) return 0;
    auto& b = e->taps[name];
    CHK(ensure(e, b, (size_t)n));
    HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    e->tap_numel[name] = n;
    return 0;
}

std::vector<std::string> build_required(const mellow_config_t* cfg) {
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
const std::vector<std::string>& default_required() {
    static std::vector<std::string> k = build_required(nullptr);
    return k;
}
bool is_ignored_key(const std::string& k) {
    std::string E = ENC;
    return k == E + "bn0.num_batches_tracked" || k == E + "head.weight" || k == E + "head.bias" ||
           k == std::string(LMK) + "lm_head.weight";  // tied to embed_tokens
}

// ---- host-only helpers -----------------------------------------------------------------------------------------------
void window_map_host(int R, int shift, int32_t* out) {
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
void pack_weight_host(const float* w, int N, int K, int NP, int KP, float* out) {
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
void rope_tables_host(float theta, int head_dim, int P, float* c, float* s) {
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
    HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));      // (streams: on first use)
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&e->ev_phase[i]));
    *out = e;
    // the default numeric mode (include/mellow_hip.h): fp32-accurate GEMMs on the bf16 matrix pipe -- the mode bench.py measures
    // and every engine-level parity test runs (next to the exact-fp32 MFMA mode); mellow_engine_set_precision overrides it
    return mellow_engine_set_precision(e, MELLOW_PRECISION_F32X3);
}

void mellow_engine_destroy(mellow_engine_t* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->parent && --e->parent->n_forks == 0) e->parent->prefill_parts = e->parent->prefill_parts_saved;    // the last fork is gone
    if (e->step_exec) hipGraphExecDestroy(e->step_exec);
    if (e->step_exec8) hipGraphExecDestroy(e->step_exec8);
    for (void* p : e->allocs) hipFree(p);      // a fork's list holds only what it allocated itself (resample banks): the weights are its parent's
    mellow_engine::Buf* bufs[] = {&e->wavcat, &e->wpad, &e->power, &e->logmel, &e->X0, &e->X1, &e->T, &e->QKV, &e->H, &e->ats,
                                  &e->fpx, &e->fpxavg, &e->latv, &e->emb33, &e->e1, &e->gbuf, &e->sbuf, &e->proj33,
                                  &e->lm_x, &e->lm_xn, &e->lm_q, &e->lm_o, &e->lm_h, &e->lm_xn3, &e->lm_o3, &e->lm_h3, &e->lm_ssq, &e->enc_a3, &e->enc_h3, &e->sk_ws, &e->kcache, &e->vcache, &e->kcache16, &e->vcache16, &e->dec,
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

// per-context device words of the generation loop (position, stop bookkeeping, block liveness) + the mapped progress word
int alloc_state_words(mellow_engine* e) {
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

// ---- the split prefill's side streams: created once, and MEASURED to run beside the main stream -----------------------------
// One thread stamps the device-wide 100 MHz clock, spins for `ticks`, stamps again.
__global__ void stream_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    out[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    out[1] = wall_clock64();
}
// true when a kernel enqueued on `b` right after one on `a` STARTS before the first has ended (different hardware queues)
static int streams_overlap(mellow_engine* e, hipStream_t a, hipStream_t b, bool* yes) {
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc(&d, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), a));
    HIPCHK(hipStreamSynchronize(a));
    HIPCHK(hipStreamSynchronize(b));
    hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(1), 0, a, d, 20000ull);          // 200 us
    hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(1), 0, b, d + 2, 2000ull);       // 20 us
    HIPCHK(hipStreamSynchronize(a));
    HIPCHK(hipStreamSynchronize(b));
    unsigned long long h[4];
    HIPCHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    *yes = h[2] < h[1];
    return 0;
}
int ensure_prefill_streams(mellow_engine* e) {
    // A failed probe is not final: another process (or a profiler) may have kept the queues busy during the 200 us window.  The
    // engine stays on what it has for the next 16 calls, then probes again; streams that passed are kept.
    if (e->n_forks > 0 || e->parent) return 0;       // pipelined contexts do not split (mellow_engine_fork)
    if (e->streams_probed) {
        if (++e->probe_backoff < 16) { e->prefill_parts = e->prefill_parts_ok; return 0; }
        e->streams_probed = false;
        e->probe_backoff = 0;
        e->prefill_parts = e->prefill_parts_want;
    }
    int nh = e->prefill_parts < 1 ? 1 : (e->prefill_parts > 4 ? 4 : e->prefill_parts);
    for (int h = 1; h < nh; ++h) {
        if (e->stream2[h - 1]) continue;          // created and measured by an earlier call
        bool ok = false;
        for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
            hipStream_t st = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            int rc = streams_overlap(e, e->stream, st, &ok);
            for (int k = 1; k < h && ok && !rc; ++k) rc = streams_overlap(e, e->stream2[k - 1], st, &ok);
            if (rc) { hipStreamDestroy(st); return rc; }
            if (ok) e->stream2[h - 1] = st; else hipStreamDestroy(st);
        }
        if (!ok) {                // no further free hardware queue right now: run on the h streams that passed, and say so
            e->prefill_parts_want = e->prefill_parts;
            e->prefill_parts = e->prefill_parts_ok = h;
            e->streams_probed = true;            // = "probed and short of queues": tried again after 16 calls
            break;
        }
    }
    return 0;
}
// ---- explicit configuration --------------------------------------------------------------------------------------------------
// Every switch of the library is a named option set through mellow_engine_set_option (before mellow_engine_finalize) and reported
// by mellow_engine_describe: the library reads no environment variable, so a C-ABI consumer's environment cannot change its
// arithmetic.  "answers" marks the options that may change last bits or the numeric mode; the others select between forms that
// are bit-identical (asserted in tests/test_gpu_parity.py) or change scheduling only.
struct OptDesc { const char* key; const char* dflt; bool answers; const char* what; };
static const OptDesc kOptions[] = {
    {"prefill_split", "2", false, "f32x3: parts of the split LM prefill (1..4); bit-identical for every value"},
    {"prefill_fuse_norm", "1", true, "f32x3: norm-free LM prefill (GEMM epilogues emit split output + RMS partials); 0 = a normalisation launch per GEMM input"},
    {"decode_fuse", "1", true, "down projection of layer l fused with q/k/v of layer l + 1 through the load-time product W'Wd; 0 = five launches per layer"},
    {"decode_fuse_max_rb", "1", true, "largest number of 32-row blocks that runs the fused down + q/k/v launch on fp32 weights"},
    {"splitk", "8", true, "f32x3 encoder: largest split count of the split-K launches (0 = never split; a summation order)"},
    {"enc_apb", "0x1CC", false, "f32x3 encoder: Swin stages whose producers hand over pre-split (bit st, 4 + st, 8 + st); bit-identical"},
    {"graph", "1", false, "decode steps as hipGraph replays (0 = eager launches)"},
    {"fp8_decode", "1", true, "fp8 mode: e4m3 weights in the decode kernels"},
    {"fp8_prefill", "1", true, "fp8 mode: e4m3 GEMMs in encoder + LM prefill"},
    {"fp8_decode_act", "1", true, "fp8 mode: e4m3 activations in the decode GEMM kernels (fp8 matrix pipe)"},
    {"fp8_kv16", "1", true, "fp8 mode: bf16 shadow of the K/V pages for the decode step"},
    {"fp8_attn_bf16", "1", true, "fp8 mode: q / k / v stored once as bf16 (LM: bf16 K/V pages + bf16-once prefill attention; Swin: bf16 q/k/v rows into the exact window attention); 0 = fp32 q/k/v, the 3-way split attention"},
    {"decode_x3", "7", true, "f32x3: mask of decode GEMM launches on the bf16 pipe (1 lm_head, 2 gate/up, 4 fused down + q/k/v)"},
    {"decode_x3_min_rb", "2", true, "f32x3: fewest 32-row blocks at which the layer launches take their f32x3 forms"},
    {"x3_stft", "1", true, "f32x3: STFT / mel GEMMs on the split kernel (0 = exact fp32 kernel)"},
    {"stft_fft", "1", true, "f32x3: STFT as a 1024-point FFT when the checkpoint's conv weights are the windowed DFT basis"},
    {"x3_apb", "1", true, "f32x3 LM prefill: producers hand over pre-split, GEMMs staged by LDS-DMA (0 = register-staged kernel on fp32 activations)"},
    {"x3_attn", "1", true, "f32x3: prefill attention on exact bf16 splits (0 = exact fp32 MFMA kernel)"},
    {"x3w", "1", false, "f32x3: K = 96 GEMMs on the weight-stationary persistent kernel; bit-identical"},
    {"row_migration", "1", false, "reference-semantics loop, B > 32: rows still running are packed into the lowest 32-row blocks; tokens unchanged"},
    {"arena_mb", "3400", false, "size of the weight arena in MiB (before the first tensor is loaded)"},
};
static const OptDesc* find_opt(const char* key) {
    for (const auto& o : kOptions)
        if (!strcmp(o.key, key)) return &o;
    return nullptr;
}
static long opt_val(const mellow_engine* e, const char* key) {
    auto it = e->opts.find(key);
    return strtol(it != e->opts.end() ? it->second.c_str() : find_opt(key)->dflt, nullptr, 0);
}
// option table + precision mode -> resolved fields (the only place they are written)
int apply_options(mellow_engine* e) {
    const int mode = e->mode;
    e->prefill_parts = (int)opt_val(e, "prefill_split");
    e->prefill_fuse_norm = opt_val(e, "prefill_fuse_norm") != 0;
    e->decode_fuse = opt_val(e, "decode_fuse") != 0;
    e->dec_fuse_max_rb = (int)opt_val(e, "decode_fuse_max_rb");
    e->sk_max = (int)opt_val(e, "splitk");
    e->enc_apb_stages = (int)opt_val(e, "enc_apb");
    e->use_graph = opt_val(e, "graph") != 0;
    e->x3_stft = opt_val(e, "x3_stft") != 0;
    e->stft_fft = opt_val(e, "stft_fft") != 0 && e->x3_stft;
    e->x3_apb = opt_val(e, "x3_apb") != 0;
    e->x3_attn = opt_val(e, "x3_attn") != 0;
    e->x3w = opt_val(e, "x3w") != 0;
    e->row_migration = opt_val(e, "row_migration") != 0;
    e->arena_mb = (int)opt_val(e, "arena_mb");
    e->fp8 = mode == MELLOW_PRECISION_FP8;
    e->fp8_decode = e->fp8 && opt_val(e, "fp8_decode") != 0;
    e->fp8_decode_act = e->fp8_decode && opt_val(e, "fp8_decode_act") != 0;
    e->fp8_prefill = opt_val(e, "fp8_prefill") != 0;
    e->fp8_attn_bf16 = opt_val(e, "fp8_attn_bf16") != 0;
    e->kv16 = e->fp8_decode && opt_val(e, "fp8_kv16") != 0;      // fp8 mode: bf16 shadow pages for the decode step
    // f32x3: six partial products (a2*b3, a3*b2, a3*b3 dropped: < 2^-23 |a*b| in total); measured error against an fp64 product is
    // identical to the nine-term form and slightly below the fp32 MFMA kernel's (tools/f32x3_check.py)
    e->f32x3_terms = mode == MELLOW_PRECISION_F32X3 ? 6 : 0;
    // f32x3 mode: the decode step's GEMM launches split their operands in registers and run on the bf16 pipe as well (decode.hip)
    e->dec_x3 = mode == MELLOW_PRECISION_F32X3 ? ((int)opt_val(e, "decode_x3") & DEC_X3_ALL) : 0;
    e->dec_x3_min_rb = (int)opt_val(e, "decode_x3_min_rb") < 1 ? 1 : (int)opt_val(e, "decode_x3_min_rb");
    return 0;
}

extern "C" {

int mellow_engine_set_option(mellow_engine_t* e, const char* key, const char* value) {
    if (!e || !key || !value) return fail("null argument");
    if (e->finalized) return fail("options must be set before mellow_engine_finalize");
    const OptDesc* o = find_opt(key);
    if (!o) return fail("unknown option \"%s\" (mellow_engine_describe lists them)", key);
    char* end = nullptr;
    const long v = strtol(value, &end, 0);
    if (end == value || *end) return fail("option \"%s\": \"%s\" is not an integer", key, value);
    if (!strcmp(key, "prefill_split") && (v < 1 || v > 4)) return fail("option prefill_split must be 1..4");
    if (!strcmp(key, "arena_mb") && e->arena) return fail("option arena_mb must be set before the first tensor is loaded");
    e->opts[key] = value;
    return apply_options(e);
}

// JSON object: ABI, precision mode, every option with its resolved value, its default and whether it can change the answers,
// and what the engine has found out about itself (prefill parts in use, STFT form).  Returns the number of bytes the text needs
// (including the terminator); it is written only when capacity suffices.
int64_t mellow_engine_describe(mellow_engine_t* e, char* buf, int64_t capacity) {
    if (!e) return -1;
    std::string s = "{\"abi\": [" + std::to_string(MELLOW_ABI_VERSION) + ", " + std::to_string(MELLOW_ABI_MINOR) + "], \"precision\": \"";
    s += e->mode == MELLOW_PRECISION_F32 ? "f32" : e->mode == MELLOW_PRECISION_FP8 ? "fp8" : "f32x3";
    s += "\", \"reads_environment\": false, \"finalized\": ";
    s += e->finalized ? "true" : "false";
    s += ", \"stft_is_fft\": ";
    s += e->fft_win ? "true" : "false";
    s += ", \"non_default\": [";
    bool first = true;
    for (const auto& o : kOptions)
        if (opt_val(e, o.key) != strtol(o.dflt, nullptr, 0)) { s += std::string(first ? "" : ", ") + "\"" + o.key + "\""; first = false; }
    s += "], \"options\": {";
    first = true;
    for (const auto& o : kOptions) {
        s += std::string(first ? "" : ", ") + "\"" + o.key + "\": {\"value\": " + std::to_string(opt_val(e, o.key)) + ", \"default\": " +
             std::to_string(strtol(o.dflt, nullptr, 0)) + ", \"changes_answers\": " + (o.answers ? "true" : "false") + "}";
        first = false;
    }
    s += "}}";
    if (buf && capacity > (int64_t)s.size()) memcpy(buf, s.c_str(), s.size() + 1);
    return (int64_t)s.size() + 1;
}

int mellow_last_steps_enqueued(mellow_engine_t* e) { return e ? e->last_steps_enqueued : -1; }

int mellow_last_row_repacks(mellow_engine_t* e) { return e ? e->last_compactions : -1; }

int mellow_stft_is_fft(mellow_engine_t* e) { return e && e->fft_win ? 1 : 0; }

int mellow_abi_minor(void) { return MELLOW_ABI_MINOR; }
int mellow_prefill_parts(mellow_engine_t* e) {
    if (!e) return -1;
    if (!e->f32x3_terms && !(e->fp8 && e->fp8_prefill)) return 1;             // only the f32x3 / fp8 prefills (producers hand over in operand format) split
    if (hipSetDevice(e->device) != hipSuccess) return -1;
    if (ensure_prefill_streams(e) != 0) return -1;
    return e->prefill_parts < 1 ? 1 : (e->prefill_parts > 4 ? 4 : e->prefill_parts);
}

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

int mellow_engine_set_precision(mellow_engine_t* e, int mode) {
    if (!e) return fail("null engine");
    if (e->finalized) return fail("precision must be chosen before mellow_engine_finalize");
    if (mode != MELLOW_PRECISION_F32 && mode != MELLOW_PRECISION_FP8 && mode != MELLOW_PRECISION_F32X3)
        return fail("unknown precision mode %d", mode);
    e->mode = mode;
    return apply_options(e);
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
    c->opts = parent->opts; c->mode = parent->mode; c->x3_stft = parent->x3_stft; c->stft_fft = parent->stft_fft; c->x3_apb = parent->x3_apb; c->x3_attn = parent->x3_attn; c->x3w = parent->x3w;
    c->row_migration = parent->row_migration; c->decode_fuse = parent->decode_fuse; c->arena_mb = parent->arena_mb;
    c->prefill_fuse_norm = parent->prefill_fuse_norm; c->dec_fuse_max_rb = parent->dec_fuse_max_rb; c->enc_apb_stages = parent->enc_apb_stages; c->sk_max = parent->sk_max;
    c->fp8 = parent->fp8; c->fp8_decode = parent->fp8_decode; c->fp8_decode_act = parent->fp8_decode_act; c->fp8_prefill = parent->fp8_prefill; c->fp8_attn_bf16 = parent->fp8_attn_bf16; c->kv16 = parent->kv16; c->f32x3_terms = parent->f32x3_terms; c->dec_x3 = parent->dec_x3; c->dec_x3_min_rb = parent->dec_x3_min_rb;
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
    // every failure below releases what the child already owns (mellow_engine_destroy copes with a half-built context)
#define FORK_HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { mellow_engine_destroy(c); return fail("HIP error %s at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)
    FORK_HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // Contexts that pipeline whole batches do not split their prefill: their overlap comes from each other, and HIP maps streams
    // onto a handful of hardware queues -- with extra streams per context two contexts' main streams end up on ONE queue and
    // serialise (measured: `pipelined` 543 -> 450 responses/s).  While it has forks the parent does not split either (its extra
    // streams stay allocated and idle); it splits again once its last fork is destroyed.  A fork must not be created or destroyed
    // while a call is running on the PARENT (include/mellow_hip.h): this is the one place a fork touches its parent.
    c->prefill_parts = 1;
    FORK_HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) FORK_HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) FORK_HIPCHK(hipEventCreate(&c->ev_phase[i]));
#undef FORK_HIPCHK
    if (alloc_state_words(c)) { mellow_engine_destroy(c); return 1; }
    c->parent = parent;
    if (parent->n_forks++ == 0) { parent->prefill_parts_saved = parent->prefill_parts; parent->prefill_parts = 1; }
    *out = c;
    return 0;
}

int mellow_set_graph(mellow_engine_t* e, int on) {
    if (!e) return fail("null engine");
    e->use_graph = on != 0;
    return 0;
}

}  // extern "C"
