// GEMM dispatch and the audio path: front-end (A1-A3), HTSAT encoder (A4-A10), c2l + projection (A11-A13); taps.
#include "engine_internal.h"

// ---- GEMM wrappers -------------------------------------------------------------------------------------------------------
// split-K workspace of the f32x3 kernels (kernels.h: sk_*), handed to launches of the encoder chain only
static void with_splitk(mellow_engine* e, GemmArgs& g) {
    if (!e->sk_enable || e->sk_max < 2 || !e->sk_ws.p) return;
    g.sk_ws = e->sk_ws.p;
    g.sk_tiles = 256;
    g.sk_max = e->sk_max;
}
int run_gemm(mellow_engine* e, const GemmArgs& a) {
    // f32x3: every dense GEMM of encoder + prefill, the STFT (EPI_POWER, K = 1024, framed A operand) and the mel projection
    // included (-0.6 ms per pass).  Through the split kernel the power spectrum differs from the ORACLE's fp32 conv1d by 2.5e-6
    // of its maximum -- two fp32 summation orders of a 1024-term dot product, squared -- while against an fp64 STFT it is
    // closer than the oracle's own fp32 arithmetic (tests/test_gpu_parity.py::test_encoder_taps holds it to both).
    // Option "x3_stft" = 0 keeps the front-end on the exact fp32 kernel.
    const bool x3_stft = e->x3_stft;
    if (e->f32x3_terms && a.K % 16 == 0 &&
        ((a.a_mode == A_PLAIN && (a.epi == EPI_LINEAR || a.epi == EPI_SWIGLU || a.epi == EPI_QKV_ROPE)) || (x3_stft && a.K >= 192))) {
        auto it = e->bf_w.find(a.Wp);
        if (it != e->bf_w.end()) {
            // fused kernel: A stays fp32 (global and LDS) and is split into its three bf16 terms in registers
            GemmArgs g = a;
            g.W8 = reinterpret_cast<const uint8_t*>(it->second);
            with_splitk(e, g);
            ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
            ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 200;
            launch_gemm_bf16x3_fused(g, e->stream);
            return 0;
        }
    }
    if (e->fp8 && e->fp8_prefill && a.a_mode == A_PLAIN && (a.epi == EPI_LINEAR || a.epi == EPI_SWIGLU || a.epi == EPI_QKV_ROPE)) {
        auto it = e->fp8_w.find(a.Wp);
        if (it != e->fp8_w.end()) {
            // no producer handed this input over quantised: the standalone fp32 -> AMX pass, then the MX GEMM (same epilogue);
            // profiled as one launch of the family
            const int64_t lda8 = (a.K + 63) / 64 * 64, Mp = rup(a.M, 128);
            CHK(ensure(e, e->a8, ((size_t)Mp * lda8 + 3) / 4));
            CHK(ensure(e, e->a8_scale, (size_t)Mp * ((lda8 / 64 + 3) / 4) * 2));      // 2 scale words per row and four k64 steps
            GemmArgs g = a;
            g.A8 = reinterpret_cast<const uint8_t*>(e->a8.p); g.lda8 = lda8; g.a_sc = reinterpret_cast<const uint32_t*>(e->a8_scale.p);
            g.W8 = it->second.w8; g.w_scale = it->second.scale;
            ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
            ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 100;
            launch_quant_mx8(a.A, a.lda, a.M, a.K, e->a8.p, e->a8_scale.p, e->stream);
            launch_gemm_fp8(g, e->stream);
            return 0;
        }
    }
    ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
    ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi;
    launch_gemm(a, e->stream);
    return 0;
}
// The activation arrives from its PRODUCER already in the GEMM's operand format and LDS order, and both operands are staged by
// LDS-DMA; counted in the same profile family as every other dense GEMM.
//   f32x3 mode: a3 = APB image (three bf16 pieces, common.h) -> gemm_x3q_kernel
//   fp8 mode:   a3 = AMX image (MXFP8) + a3_scales (its scale bytes) -> gemm_mx8_kernel
int run_gemm_apb(mellow_engine* e, const GemmArgs& a, const void* a3, hipStream_t st, const void* a3_scales) {
    GemmArgs g = a;
    g.A8 = reinterpret_cast<const uint8_t*>(a3);
    if (a3_scales) {
        auto it = e->fp8_w.find(a.Wp);
        if (it == e->fp8_w.end()) return fail("internal: no e4m3 copy of this weight");
        g.a_sc = reinterpret_cast<const uint32_t*>(a3_scales); g.lda8 = rup(a.K, 64);
        g.W8 = it->second.w8; g.w_scale = it->second.scale;
        ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
        ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 500;
        launch_gemm_fp8(g, st ? st : e->stream);
        return 0;
    }
    auto it = e->bf_w.find(a.Wp);
    if (it == e->bf_w.end()) return fail("internal: no bf16-split copy of this weight");
    g.W8 = reinterpret_cast<const uint8_t*>(it->second);
    if (!st || st == e->stream) with_splitk(e, g);
    g.no_x3w = !e->x3w;
    ProfScope ps(e, PF_GEMM, gemm_flops(a), 0.0);
    ps.r.M = a.M; ps.r.N = a.Nw; ps.r.K = a.K; ps.r.epi = a.epi + 300;
    launch_gemm_bf16x3_apb(g, st ? st : e->stream);
    return 0;
}
GemmArgs lin(const float* A, int64_t lda, int M, const Packed& w, float* C, int64_t ldc, const float* bias) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.M = M; g.K = w.KP; g.Wp = w.p; g.Nw = w.Nw; g.N = rup(w.N, 4); g.C = C; g.ldc = ldc; g.bias = bias;
    return g;
}

// ---- encoder --------------------------------------------------------------------------------------------------------------
// wav dev [n][n_samples] -> proj33 [n][33][576] in e->proj33
int run_encoder(mellow_engine* e, const float* wav, int n, int64_t n_samples, int want_logmel_only, int apply_bn,
                       float* logmel_out) {
    if (n <= 0) return fail("n_clips must be positive");
    if (n_samples % 4 || n_samples < kNfft) return fail("n_samples must be a multiple of 4 and >= 1024");
    hipStream_t s = e->stream;
    struct SkScope {            // the encoder chain is one stream: its under-filled GEMM launches may use the split-K workspace
        mellow_engine* e;
        ~SkScope() { e->sk_enable = false; }
    } sk_scope{e};
    if (e->f32x3_terms && e->sk_max >= 2) {
        CHK(ensure(e, e->sk_ws, (size_t)512 * 16384));
        e->sk_enable = e->sk_ws.p != nullptr;
    }
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
            // f32x3 mode, stages in enc_apb_stages: the LayerNorms and the GELU epilogue of fc1 write their output pre-split in APB
            // order and qkv / fc1 / fc2 run on the LDS-DMA kernel (gemm_x3q_kernel); H never exists as fp32
            // bit 8 + st: only the LayerNorms hand over pre-split (qkv and fc1 on the APB kernels, fc1 still writes fp32): stage 0,
            // whose K = 96 GEMMs then run on the weight-stationary persistent kernel (gemm_x3w_kernel)
            // fp8 mode (`amx`), every stage: the same hand-over with AMX images (MXFP8, common.h) -- the LayerNorms and the GELU
            // epilogue of fc1 emit e4m3 + block scales in the consumer's LDS order, qkv / fc1 / fc2 run on gemm_mx8_kernel straight
            // from them (K = 96 zero-padded to two k64 steps); the window attention still writes fp32 (head_dim 24 does not tile 32-
            // column blocks) and the proj GEMM takes the standalone quantiser
            const bool amx = e->fp8 && e->fp8_prefill && e->x3_apb && e->fp8_w.count(w.qkv.p) && e->fp8_w.count(w.fc1.p) && e->fp8_w.count(w.fc2.p);
            const bool have_pb = e->f32x3_terms && e->bf_w.count(w.qkv.p) && e->bf_w.count(w.fc1.p) && e->bf_w.count(w.fc2.p);
            const bool apb_h = amx || (have_pb && ((e->enc_apb_stages >> st) & 1));
            const bool apb = apb_h || (have_pb && ((e->enc_apb_stages >> (8 + st)) & 1));
            const size_t M1p = (size_t)rup(M1, 128);
            char *a3s = nullptr, *h3s = nullptr;          // fp8 mode: scale bytes behind the image data, in the same buffers
            if (apb) {
                CHK(ensure(e, e->enc_a3, (M1p * C * 6 + 3) / 4));
                if (apb_h) CHK(ensure(e, e->enc_h3, (M1p * 4 * C * 6 + 3) / 4));
                if (amx) {
                    a3s = reinterpret_cast<char*>(e->enc_a3.p) + M1p * rup(C, 64);
                    h3s = reinterpret_cast<char*>(e->enc_h3.p) + M1p * rup(4 * C, 64);
                }
                { ProfScope ps(e, PF_NORM, 0, 2.5 * M1 * C * 4); launch_layernorm_apb(x, e->enc_a3.p, M1, C, w.n1w, w.n1b, map, N, s, a3s); }
                {
                    GemmArgs g = lin(nullptr, C, M1, w.qkv, e->QKV.p, 3 * C, w.qkv_b);
                    g.c16 = (amx && e->fp8_attn_bf16) ? 1 : 0;      // fp8 mode: q / k / v of the block as bf16 rows (the window attention widens them)
                    CHK(run_gemm_apb(e, g, e->enc_a3.p, s, a3s));
                }
            } else {
                { ProfScope ps(e, PF_NORM, 0, 2.0 * M1 * C * 4); launch_layernorm(x, t, M1, C, w.n1w, w.n1b, map, N, s); }
                CHK(run_gemm(e, lin(t, C, M1, w.qkv, e->QKV.p, 3 * C, w.qkv_b)));
            }
            // bits 4..7 of enc_apb_stages: the window attention hands its output over pre-split too (proj on the x3q kernel)
            const bool apb_proj = apb && !amx && ((e->enc_apb_stages >> (4 + st)) & 1) && e->bf_w.count(w.proj.p);
            {
                ProfScope ps(e, PF_WINDOW_ATTN, 4.0 * 64 * 64 * 24 * (double)(M1 / 64) * nH, 4.0 * M1 * C * 4);
                launch_window_attention(e->QKV.p, t, M1, C, nH, w.bias_exp, shifted ? w.mask : nullptr, nW, s, apb_proj ? e->enc_a3.p : nullptr, amx && e->fp8_attn_bf16);
            }
            {
                GemmArgs g = lin(t, C, M1, w.proj, x, C, w.proj_b);
                g.resid = x; g.ldr = C; g.crow_map = map; g.rows_in = N; g.rows_out = N;
                if (apb_proj) CHK(run_gemm_apb(e, g, e->enc_a3.p, s));
                else CHK(run_gemm(e, g));
            }
            if (apb) {
                { ProfScope ps(e, PF_NORM, 0, 2.5 * M1 * C * 4); launch_layernorm_apb(x, e->enc_a3.p, M1, C, w.n2w, w.n2b, nullptr, N, s, a3s); }
                {
                    GemmArgs g = lin(nullptr, C, M1, w.fc1, apb_h ? nullptr : e->H.p, 4 * C, w.fc1_b);
                    g.act = ACT_GELU; g.C3 = apb_h ? e->enc_h3.p : nullptr;
                    if (amx) { g.c3_fmt = 1; g.C3s = reinterpret_cast<uint8_t*>(h3s); g.c3_kt64 = rup(4 * C, 64) / 64; }
                    CHK(run_gemm_apb(e, g, e->enc_a3.p, s, a3s));
                }
                {
                    GemmArgs g = lin(apb_h ? nullptr : e->H.p, 4 * C, M1, w.fc2, x, C, w.fc2_b);
                    g.resid = x; g.ldr = C;
                    if (apb_h) CHK(run_gemm_apb(e, g, e->enc_h3.p, s, h3s));
                    else CHK(run_gemm(e, g));
                }
            } else {
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
