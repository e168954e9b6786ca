// mellow_engine_finalize: every reference checkpoint key (reference wrapper.py:74-82, load_state_dict) -> the device layouts the
// kernels read (engine_internal.h has the map of the host-side files).
#include "engine_internal.h"

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
    if (e->fp8 && e->fp8_prefill && !e->decode_only_weight) {       // WMX image: K zero-padded to whole k64 steps (K = 96 -> 128)
        float *w8f = nullptr, *sc = nullptr;
        CHK(dev_alloc(e, &w8f, ((size_t)p.NP * rup(p.KP, 64) + 3) / 4));
        CHK(dev_alloc(e, &sc, (size_t)p.NP));
        launch_pack_fp8(p.p, p.NP, p.KP, reinterpret_cast<uint8_t*>(w8f), sc, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(e->stream));
        e->fp8_w[p.p] = {reinterpret_cast<uint8_t*>(w8f), sc};
    }
    *out = p;
    return 0;
}
// e4m3 (WMX) copy of an already packed weight: the operand of gemm_mx8_kernel (fp8 mode)
static int make_w8(mellow_engine* e, const Packed& p) {
    if (!e->fp8 || !e->fp8_prefill || e->fp8_w.count(p.p)) return 0;
    float *w8f = nullptr, *sc = nullptr;
    CHK(dev_alloc(e, &w8f, ((size_t)p.NP * rup(p.KP, 64) + 3) / 4));
    CHK(dev_alloc(e, &sc, (size_t)p.NP));
    launch_pack_fp8(p.p, p.NP, p.KP, reinterpret_cast<uint8_t*>(w8f), sc, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    e->fp8_w[p.p] = {reinterpret_cast<uint8_t*>(w8f), sc};
    return 0;
}
// bf16-split (PB) copy of an already packed weight: the operand of the x3q GEMM (f32x3 mode)
static int make_pb(mellow_engine* e, const Packed& p) {
    if (!e->f32x3_terms || p.KP % 16 != 0 || e->bf_w.count(p.p)) return 0;
    float* pb = nullptr;
    CHK(dev_alloc(e, &pb, ((size_t)p.NP * p.KP * 6 + 3) / 4));
    launch_pack_bf16x3(p.p, p.NP, p.KP, pb, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    e->bf_w[p.p] = pb;
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
static int make_packed16(mellow_engine* e, const float* w, int N, int K, float** out, bool natural = false) {
    if (N % 16 || K % (natural ? 32 : 16)) return fail("P16 packing needs N and K multiples of 16 (32 for the P16N order)");
    float* d0 = nullptr;
    HIPCHK(hipMalloc(&d0, (size_t)N * K * sizeof(float)));
    HIPCHK(hipMemcpy(d0, w, (size_t)N * K * sizeof(float), hipMemcpyHostToDevice));
    CHK(dev_alloc(e, out, (size_t)N * K));
    if (natural) launch_pack_weight16n(d0, N, K, *out, e->stream);
    else launch_pack_weight16(d0, N, K, *out, e->stream);
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
        const bool no_fft = !e->stft_fft;            // options "stft_fft" / "x3_stft"
        if ((e->f32x3_terms || (e->fp8 && e->fp8_prefill)) && !no_fft && kNfft == 1024) {       // (fp8 mode: the front-end stays fp32 arithmetic; the FFT is that)
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
    const bool no_fuse = !e->decode_fuse;   // option "decode_fuse" = 0: keep the 5-launch layer
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
            if (e->prefill_fuse_norm) { CHK(make_pb(e, w.qkv_f)); CHK(make_w8(e, w.qkv_f)); }        // f32x3 / fp8 prefill without norm launches (run_prefill)
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
            if (e->prefill_fuse_norm && (e->f32x3_terms || (e->fp8 && e->fp8_prefill))) {   // the folded gate/up in the prefill's pair layout (+ its bf16 split / e4m3 copy)
                e->decode_only_weight = true;
                CHK(make_packed(e, gf.data(), uf.data(), I, H, &w.gateup_f));
                e->decode_only_weight = false;
                CHK(make_pb(e, w.gateup_f));
                CHK(make_w8(e, w.gateup_f));
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
                if (e->f32x3_terms && fuse && !e->fp8_decode) CHK(make_packed16(e, il.data(), 2 * I, H, &w.gu16n, true));    // f32x3 layer kernels (row blocks >= dec_x3_min_rb)
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
