// The decoder LM: KV pages and decode workspaces, prefill (A15), the KV-cached decode step, prefix assembly (A14), the lm taps and
// mellow_generate (A16: reference wrapper.py:197-256).
#include "engine_internal.h"


int ensure_lm(mellow_engine* e, int B, int T, int Tmax, int ctx_end) {
    if (B > 1024) return fail("batch of %d exceeds the 1024 rows one pass takes (mellow_generate chunks larger batches itself; the decode state block is sized for 32 row blocks)", B);
    if (ctx_end <= 0 || ctx_end > Tmax) ctx_end = Tmax;      // last context length the call will reach (<= page capacity)
    if (Tmax > e->cfg.max_positions) return fail("prefix + max_len = %d exceeds max_positions %d", Tmax, e->cfg.max_positions);
    const size_t Mp = (size_t)B * T;
    CHK(ensure(e, e->lm_x, Mp * 576));
    CHK(ensure(e, e->lm_xn, Mp * 576));
    CHK(ensure(e, e->lm_q, Mp * 576));
    CHK(ensure(e, e->lm_o, Mp * 576));
    CHK(ensure(e, e->lm_h, Mp * 1536));
    if (e->f32x3_terms || (e->fp8 && e->fp8_prefill)) {      // 6 bytes per element (the fp8 mode's AMX images + scale bytes need 1.05), rows padded to whole 128-row panels
        const size_t Mq = (size_t)rup((int)Mp, 128) + 3 * 128;   // + three panels: every part of the split prefill starts on a panel boundary
        CHK(ensure(e, e->lm_xn3, Mq * 576 * 6 / 4));
        CHK(ensure(e, e->lm_o3, Mq * 576 * 6 / 4));
        CHK(ensure(e, e->lm_h3, Mq * 1536 * 6 / 4));
        CHK(ensure(e, e->lm_ssq, Mq * 2 * 16));        // two statistics per row (after o_proj / after down), 9 partial sums each
    }
    // fp8 mode at its defaults: K and V exist as bf16 pages ONLY -- written by the q/k/v epilogue (rounded once), read by the bf16-once
    // prefill attention and by the decode attention; the fp32 pages and the conversion pass of round 5 are not touched
    e->kv16_direct = e->kv16 && e->fp8 && e->fp8_prefill && e->fp8_attn_bf16 && e->x3_apb && e->x3_attn && !e->layers.empty() &&
                     e->fp8_w.count(e->layers[0].qkv.p) != 0;
    dec_prepare_lds_attributes();            // (remembered per device: a no-op after the first call)
    const int Bp = rb_of(B) * 32;
    if (e->kv_B != Bp || e->kv_Tmax != Tmax) {
        e->kv_B = Bp;
        e->kv_Tmax = Tmax;
        CHK(ensure(e, e->kcache, kv_layer_floats(e) * e->cfg.num_layers));
        CHK(ensure(e, e->vcache, kv_layer_floats(e) * e->cfg.num_layers));
        if (e->kv16) {
            CHK(ensure(e, e->kcache16, kv_layer_floats(e) * e->cfg.num_layers / 2));
            CHK(ensure(e, e->vcache16, kv_layer_floats(e) * e->cfg.num_layers / 2));
        }
        // the decode attention loads whole key groups before it knows the position and masks them afterwards
        // (weight 0 x value): never-written page slots must hold finite numbers
        HIPCHK(hipMemsetAsync(e->kcache.p, 0, kv_layer_floats(e) * e->cfg.num_layers * sizeof(float), e->stream));
        HIPCHK(hipMemsetAsync(e->vcache.p, 0, kv_layer_floats(e) * e->cfg.num_layers * sizeof(float), e->stream));
        if (e->kv16) {
            HIPCHK(hipMemsetAsync(e->kcache16.p, 0, kv_layer_floats(e) * e->cfg.num_layers * sizeof(float) / 2, e->stream));
            HIPCHK(hipMemsetAsync(e->vcache16.p, 0, kv_layer_floats(e) * e->cfg.num_layers * sizeof(float) / 2, e->stream));
        }
        if (e->step_exec) { hipGraphExecDestroy(e->step_exec); e->step_exec = nullptr; }
        if (e->step_exec8) { hipGraphExecDestroy(e->step_exec8); e->step_exec8 = nullptr; }
    }
    {
        // carve the decode-step buffers out of one arena (all sizes are multiples of 64 floats = 256 B)
        const size_t RB = (size_t)Bp / 32, V = (size_t)e->cfg.vocab_size;
        const size_t n_x = (size_t)Bp * 576;
        size_t off = 0;
        auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
        const size_t o_xmidF = take(n_x), o_xnewR = take(n_x), o_xnF = take(n_x * 3 / 2);      // (xnF: fp32, or its 6-byte pre-split image)
        const size_t o_dslabF = take(DEC_KC_DOWN * n_x), o_ssq1 = take((size_t)Bp * DEC_KC_QKV), o_rope = take(64);
        const size_t o_pq = take((size_t)DEC_KC_QKV * Bp * 960);
        const size_t o_att = take((size_t)DEC_TS * n_x), o_aml = take((size_t)DEC_TS * 9 * Bp * 2);
        const size_t o_ssq = take((size_t)Bp * 40), o_gu = take(RB * 192 * 256), o_xmidF16 = take(n_x);
        // f32x3 layer kernels: 6-byte pre-split images of x_mid (two fragment orders) and of h
        const bool x3l = (e->dec_x3 & DEC_X3_GATEUP) && (e->dec_x3 & DEC_X3_QKV) && (int)RB >= e->dec_x3_min_rb && !e->fp8_decode &&
                         e->layers.size() > 1 && e->layers[1].qkv2 != nullptr && e->layers[1].gu16n != nullptr;
        const size_t o_x3a = take(x3l ? n_x * 3 / 2 : 0), o_x3b = take(x3l ? n_x * 3 / 2 : 0), o_h3 = take(x3l ? RB * 192 * 256 * 3 / 2 : 0);
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
        a.kv16 = e->kv16 ? 1 : 0;
        a.x3 = e->dec_x3;
        a.blk_live = nullptr;                               // mellow_generate turns the per-block early exit on per call
        a.row_of_slot = nullptr;
        // (and the logits store off: the taps mellow_lm_prefill / mellow_lm_decode_step read dlogits, generation does not)
        a.rope_cos = e->rope_cos; a.rope_sin = e->rope_sin;
        a.xmidF = p + o_xmidF; a.xnewR = p + o_xnewR; a.xnF = p + o_xnF;
        a.xn3 = (a.x3 & DEC_X3_HEAD) && !e->head8 && dec_head3r_fits(e->cfg.vocab_size) ? (void*)(p + o_xnF) : nullptr;
        a.xmid3_32 = x3l ? (void*)(p + o_x3a) : nullptr; a.xmid3_16 = x3l ? (void*)(p + o_x3b) : nullptr; a.h3 = x3l ? (void*)(p + o_h3) : nullptr;
        a.dslabF = p + o_dslabF; a.slabF_stride4 = (int64_t)(n_x / 4); a.ssq1 = p + o_ssq1; a.rope_cur = p + o_rope;
        {
            // key split of the decode attention: balanced at the END of the reserved context, rounded down to whole
            // passes of a workgroup when that costs at most 4 groups of imbalance
            const int ng_end = (ctx_end - 1 + 3) / 4, chunk = dec_attn_chunk_groups(e->kv16);
            a.ts = dec_key_splits((int)RB, e->kv16 || e->mode != MELLOW_PRECISION_F32X3);
            int gs = (ng_end + a.ts - 1) / a.ts;
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
int clear_page_tails(mellow_engine* e, int T, int t_end) {
    const int Tmax = e->kv_Tmax;
    if (t_end > Tmax) t_end = Tmax;
    if (T >= t_end) return 0;
    if (e->kv16_direct) launch_clear_page_slots(e->vcache16.p, (int64_t)e->cfg.num_layers * e->kv_B * 3, Tmax, T, t_end, e->stream, true);
    else launch_clear_page_slots(e->vcache.p, (int64_t)e->cfg.num_layers * e->kv_B * 3, Tmax, T, t_end, e->stream);
    HIPCHK(hipGetLastError());
    return 0;
}


LoopArgs loop_args(mellow_engine* e) {
    LoopArgs lp;
    lp.out_tokens = reinterpret_cast<int32_t*>(e->out_tok.p);
    lp.params = e->d_params; lp.seen_stop = e->d_seen; lp.n_seen = e->d_nseen; lp.arrive = e->d_arrive; lp.ticket = e->d_ticket;
    lp.host_progress = e->d_progress; lp.T0 = e->cfg.prefix_len;
    if (e->da.blk_live) { lp.blk_left = e->d_blk_left; lp.blk_live = e->d_blk_live; lp.blk_snap = e->d_blk_live + 32; }
    if (e->da.row_of_slot) { lp.row_of_slot = e->d_row_of_slot; lp.n_compactions = e->d_ncompact; }
    return lp;
}

// final norm (+ pending down slabs) + lm_head with fused per-tile arg-max candidates -> dlogits, d_tokens
int run_lm_head(mellow_engine* e, int B, int pending_kcd, const RecordArgs* rec) {
    const int NT = e->cfg.vocab_size / 32, Bp = e->da.rows;
    auto dh = [&](int k) { DecArgs x = e->da; x.dbg_seq = e->dbg_seq0 >= 0 ? e->dbg_seq0 + 5 * e->cfg.num_layers + k : -1000; return x; };
    { ProfScope ps(e, PF_NORM, 0, (double)(pending_kcd + 2) * Bp * 576 * 4);
      launch_dec_final_norm(dh(0), e->final_norm, pending_kcd, e->stream); }
    { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * e->cfg.vocab_size, 576.0 * e->cfg.vocab_size * 4);
      if (e->head8) launch_dec_lm_head(dh(1), e->head8, e->lm_head.KP / 8, e->cfg.vocab_size, e->stream, e->head_sc);
      else launch_dec_lm_head(dh(1), e->lm_head.p, e->lm_head.KP / 8, e->cfg.vocab_size, e->stream); }
    { ProfScope ps(e, PF_MISC, 0, 0);
      launch_dec_argmax(dh(2), B, NT, e->d_tokens, e->embed, (rec && rec->embed_next) ? 1 : 0, rec ? loop_args(e) : LoopArgs(),
                        e->stream);
      if (rec && e->da.row_of_slot) launch_dec_compact(e->da, B, loop_args(e), e->stream); }
    return 0;
}

int run_prefill(mellow_engine* e, int B, int T, const RecordArgs* rec, bool all_positions) {
    hipStream_t s = e->stream;
    const int M = B * T, Tmax = e->kv_Tmax;
    const int NL = e->cfg.num_layers;
    float *x = e->lm_x.p, *xn = e->lm_xn.p;
    // fp8 mode: the same producer -> consumer hand-over with AMX images (MXFP8, common.h) instead of APB ones: `amx`; the code below
    // says `apb` for "GEMM inputs leave their producers in operand format"
    const bool amx = e->fp8 && e->fp8_prefill && e->x3_apb && e->fp8_w.count(e->layers[0].qkv.p) != 0;
    const bool apb = (e->f32x3_terms && e->x3_apb) || amx;        // option "x3_apb" = 0: the register-staged x3p kernel / the standalone quantiser (developer A/B)
    // Split prefill (f32x3 mode): the batch is cut into independent parts (2 by default) that run the same launches on their own
    // streams, so the tails and the fill / drain of one part's kernels are covered by another's (every buffer is indexed by row
    // or by example, so a part is an offset; its pre-split operands get their own panel-aligned region).  Measured before it was
    // built with two forked contexts (tools/half_chain_probe.py).  MELLOW_PREFILL_SPLIT=n: n parts (1 = one chain, at most 4).
    if (apb && !e->prof_on && (e->prefill_parts > 1 || e->streams_probed)) CHK(ensure_prefill_streams(e));      // (may find that they would serialise: fewer parts)
    int nh = (apb && !e->prof_on) ? e->prefill_parts : 1;
    nh = nh < 1 ? 1 : (nh > 4 ? 4 : nh);
    if (nh > B) nh = B;
    int hb0[4], hB[4];
    size_t prow[4];                                              // first row of each part's panel range
    hipStream_t hs[4] = {s, e->stream2[0], e->stream2[1], e->stream2[2]};
    for (int h = 0, b0 = 0, r = 0; h < nh; ++h) {
        hb0[h] = b0; hB[h] = B / nh + (h < B % nh ? 1 : 0); prow[h] = (size_t)r;
        b0 += hB[h]; r += rup(hB[h] * T, 128);
    }
    const bool split = nh > 1;
    // Once the side streams wait on ev_fork, EVERY exit from this function has to join them back into `s` (an early error return
    // inside the layer loop would otherwise leave launches of the side streams running against buffers the caller's next call
    // reuses or frees from `s`): the guard joins in its destructor unless the normal path already has.
    struct Join {
        mellow_engine* e; hipStream_t* hs; int nh; bool done;
        int run() {
            if (done) return 0;
            done = true;
            for (int h = 1; h < nh; ++h) {
                HIPCHK(hipEventRecord(e->ev_join[h - 1], hs[h]));
                HIPCHK(hipStreamWaitEvent(hs[0], e->ev_join[h - 1], 0));
            }
            return 0;
        }
        ~Join() { (void)run(); }
    } join{e, hs, nh, !split};
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
            const bool p16 = amx && e->kv16_direct;          // bf16 pages: the same element offsets, two bytes each
            float* kc = p16 ? e->kcache16.p + (kv_layer_floats(e) * l + (size_t)hb0[h] * 3 * Tmax * 64) / 2
                            : e->kcache.p + kv_layer_floats(e) * l + (size_t)hb0[h] * 3 * Tmax * 64;
            float* vc = p16 ? e->vcache16.p + (kv_layer_floats(e) * l + (size_t)hb0[h] * 3 * Tmax * 64) / 2
                            : e->vcache.p + kv_layer_floats(e) * l + (size_t)hb0[h] * 3 * Tmax * 64;
            // pre-split operand regions of this half (6 bytes per element, whole 128-row panels)
            char* xn3 = apb ? reinterpret_cast<char*>(e->lm_xn3.p) + prow[h] * 576 * 6 : nullptr;
            char* o3 = apb ? reinterpret_cast<char*>(e->lm_o3.p) + prow[h] * 576 * 6 : nullptr;
            char* h3 = apb ? reinterpret_cast<char*>(e->lm_h3.p) + prow[h] * 1536 * 6 : nullptr;
            // fp8 mode: the same regions hold [AMX data: rows x K bytes | scale bytes: rows x ceil(K / 256) x 8]; rows of this part
            char *xn3s = nullptr, *o3s = nullptr, *h3s = nullptr;
            if (amx) {
                xn3s = xn3 + (size_t)rup(Mh, 128) * 576; o3s = o3 + (size_t)rup(Mh, 128) * 576; h3s = h3 + (size_t)rup(Mh, 128) * 1536;
            }
            auto c3_amx = [&](GemmArgs& g, char* scales, int kt64) { if (amx) { g.c3_fmt = 1; g.C3s = reinterpret_cast<uint8_t*>(scales); g.c3_kt64 = kt64; } };
            // norm-free chaining (fz): the residual stream leaves the o_proj / down GEMMs already pre-split together with its
            // sum-of-squares partials (ssq_mid after o_proj, ssq_in after down); the GEMM that follows runs on the norm-folded
            // weight and applies the row statistic to its accumulators -- 59 of the 60 normalisation launches of a prefill disappear
            // (the first layer's input is the prefix, which has no producing GEMM: it keeps its launch)
            const bool fz = apb && e->prefill_fuse_norm && w.gateup_f.p != nullptr;
            float* ssq_in = fz ? e->lm_ssq.p + prow[h] * 9 : nullptr;                       // [row][9], rows of this part
            float* ssq_mid = fz ? e->lm_ssq.p + e->lm_ssq.cap / 2 + prow[h] * 9 : nullptr;    // second half of the buffer
            auto with_rs = [&](GemmArgs& g, const float* ssq) { g.rs_ssq = ssq; g.rs_parts = 9; g.rs_dim = 576.f; g.rs_eps = e->cfg.rms_norm_eps; };
            (void)ssq_mid;
            // f32x3 mode: every GEMM input of the layer is written by its producer already split into three bf16 pieces, in the
            // order the GEMM's LDS stage wants it (APB, common.h), and the GEMM stages both operands by LDS-DMA (x3q)
            const bool fz_in = fz && l > 0;       // this layer's input came out of the previous layer's down GEMM pre-split
            if (fz_in) {}
            else if (apb) { ProfScope ps(e, PF_NORM, 0, 2.5 * Mh * 576 * 4); launch_rmsnorm_apb(xh, xn3, Mh, 576, w.in_ln, e->cfg.rms_norm_eps, st, xn3s); }
            else { ProfScope ps(e, PF_NORM, 0, 2.0 * Mh * 576 * 4); launch_rmsnorm(xh, xnh, Mh, 576, w.in_ln, e->cfg.rms_norm_eps, st); }
            {
                GemmArgs g;
                g.A = xnh; g.lda = 576; g.M = Mh; g.K = 576; g.Wp = fz_in ? w.qkv_f.p : w.qkv.p; g.Nw = 960; g.N = 960; g.epi = EPI_QKV_ROPE;
                if (fz_in) with_rs(g, ssq_in);
                g.q_out = qh; g.k_cache = kc; g.v_cache = vc; g.rope_cos = e->rope_cos; g.rope_sin = e->rope_sin;
                g.T = T; g.Tmax = Tmax; g.q_heads = 9; g.kv_heads = 3; g.kv16 = p16 ? 1 : 0;
                if (apb) CHK(run_gemm_apb(e, g, xn3, st, xn3s)); else CHK(run_gemm(e, g));
            }
            // The LAST layer only has to produce the final prefix row (nothing consumes the other rows' attention / MLP
            // outputs; their K/V pages were just written above): it is finished below by the decode kernels on B rows.
            if (l == NL - 1 && !all_positions) { last = true; continue; }
            {
                // causal QK^T + PV: 4*64 flops per (query,key) pair per head
                ProfScope ps(e, PF_PREFILL_ATTN, 4.0 * 64 * 9 * (double)Bh * ((double)T * (T + 1) / 2), 0);
                const bool attn_f32 = !e->x3_attn;   // option "x3_attn" = 0: f32x3 mode on the fp32 kernel (A/B)
                launch_prefill_attention(qh, kc, vc, oh, apb ? o3 : nullptr, Bh, T, Tmax, (e->f32x3_terms != 0 || amx) && !attn_f32, st, o3s, amx && e->fp8_attn_bf16, p16);
            }
            {
                GemmArgs g = lin(oh, 576, Mh, w.o, xh, 576, nullptr);
                g.resid = xh; g.ldr = 576;
                if (fz) { g.C3 = xn3; g.ssq_out = ssq_mid; g.ssq_parts = 9; c3_amx(g, xn3s, 9); }
                if (apb) CHK(run_gemm_apb(e, g, o3, st, o3s)); else CHK(run_gemm(e, g));
            }
            if (fz) {}
            else if (apb) { ProfScope ps(e, PF_NORM, 0, 2.5 * Mh * 576 * 4); launch_rmsnorm_apb(xh, xn3, Mh, 576, w.post_ln, e->cfg.rms_norm_eps, st, xn3s); }
            else { ProfScope ps(e, PF_NORM, 0, 2.0 * Mh * 576 * 4); launch_rmsnorm(xh, xnh, Mh, 576, w.post_ln, e->cfg.rms_norm_eps, st); }
            {
                GemmArgs g;
                g.A = xnh; g.lda = 576; g.M = Mh; g.K = 576; g.Wp = fz ? w.gateup_f.p : w.gateup.p; g.Nw = 3072; g.N = 1536; g.C = hh; g.ldc = 1536;
                g.epi = EPI_SWIGLU;
                if (fz) with_rs(g, ssq_mid);
                if (apb) { g.C3 = h3; c3_amx(g, h3s, 24); CHK(run_gemm_apb(e, g, xn3, st, xn3s)); } else CHK(run_gemm(e, g));
            }
            {
                GemmArgs g = lin(hh, 1536, Mh, w.down, xh, 576, nullptr);
                g.resid = xh; g.ldr = 576;
                if (fz) { g.C3 = xn3; g.ssq_out = ssq_in; g.ssq_parts = 9; c3_amx(g, xn3s, 9); }
                if (apb) CHK(run_gemm_apb(e, g, h3, st, h3s)); else CHK(run_gemm(e, g));
            }
        }
        if (last) break;
    }
    CHK(join.run());
    if (e->kv16 && !e->kv16_direct && !all_positions) {
        // fp8 mode: the decode step streams a bf16 shadow of the pages (whole pages: the cleared tails travel with them)
        ProfScope ps(e, PF_MISC, 0, 3.0 * kv_layer_floats(e) * NL * 4);
        launch_kv_to_bf16(e->kcache.p, e->kcache16.p, (int64_t)(kv_layer_floats(e) * NL), s);
        launch_kv_to_bf16(e->vcache.p, e->vcache16.p, (int64_t)(kv_layer_floats(e) * NL), s);
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
int enqueue_decode_layer_range(mellow_engine* e, int B, int l_begin, int l_end, bool inc_pos) {
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
        // (KV16: the bf16 shadow pages; a layer's pages are half as many floats)
        float* kc = e->kv16 ? e->kcache16.p + kv_layer_floats(e) / 2 * (same_kv ? 0 : l) : e->kcache.p + kv_layer_floats(e) * (same_kv ? 0 : l);
        float* vc = e->kv16 ? e->vcache16.p + kv_layer_floats(e) / 2 * (same_kv ? 0 : l) : e->vcache.p + kv_layer_floats(e) * (same_kv ? 0 : l);
        const int kcd = l == l_begin ? 0 : DEC_KC_DOWN;   // the first layer of the range starts from a materialised x
        // fused_in: this layer's q/k/v slabs (and the down slabs of x_new) were written by the previous layer's dec_qkv2 launch
        // The fused launch pays at ONE row block (B <= 32: 46.8 ms per 63 steps against 49.9 with the five-launch layer) and loses
        // beyond (fp32 weights, same box: B = 64: 68.4 ms fused against 68.0, B = 96: 91.2 / 87.9, B = 128: 105.6 / 102.1,
        // B = 256: 190.3 / 175.7, B = 512: 346.6 / 319.2 -- its composed operand is 1.7x the bytes of the two matrices it
        // replaces, and a larger batch is bound by bytes, not by launches); the e4m3 form was measured ahead at four row blocks
        // (round 3) and stays fused.  MELLOW_DECODE_FUSE_MAX_RB: developer override.
        const bool x3l = e->da.xmid3_32 != nullptr;       // f32x3 forms of gate/up and of the fused down + q/k/v launch (ensure_lm decides)
        const bool fuse_rb = x3l || e->da.RB <= e->dec_fuse_max_rb;
        const bool fused_in = l > l_begin && ((w.qkv2 != nullptr && fuse_rb) || w.q2h8 != nullptr) && !same_w;
        DecArgs a = e->da;
        const int sq = e->dbg_seq0 >= 0 ? e->dbg_seq0 + 5 * (l - l_begin) : -1000;       // launch index inside the step (kdebug builds)
        auto da = [&](int k) { DecArgs x = e->da; x.dbg_seq = sq + k; return x; };
        a.dbg_seq = sq;
        a.first = l == l_begin ? 1 : 0;                     // the first kernel of a step stages the RoPE row ...
        a.inc_pos = (inc_pos && l == l_begin) ? 1 : 0;     // ... and advances the position word
        if (!(skip & 1) && !fused_in)
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 960.0, 576.0 * 960.0 * 4);
          if (w.qkv8) launch_dec_qkv(a, w.qkv8, w.qkv_f.KP / 8, kcd, s, w.qkv_sc);
          else launch_dec_qkv(a, w.qkv_f.p, w.qkv_f.KP / 8, kcd, s); }
        if (!(skip & 2))
        { ProfScope ps(e, PF_DECODE_ATTN, 4.0 * 64 * 9 * (double)B * (e->cur_pos + 1), 2.0 * (double)B * 3 * 64 * 4 * (e->cur_pos + 1));
          launch_dec_attn(da(1), kc, vc, fused_in, s); }
        if (!(skip & 4))
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 576.0, 576.0 * 576.0 * 4);
          if (w.o8) launch_dec_oproj(da(2), w.o8, s, w.o_sc);
          else launch_dec_oproj(da(2), w.o16, s); }
        if (!(skip & 8))
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 3072.0, 576.0 * 3072.0 * 4);
          if (w.gu8) launch_dec_gateup(da(3), w.gu8, s, w.gu_sc);
          else if (x3l) launch_dec_gateup3(da(3), w.gu16n, s);
          else launch_dec_gateup(da(3), w.gu16, s); }
        const LMLayerW* nx = (l + 1 < l_end && !same_w) ? &e->layers[l + 1] : nullptr;
        if (nx && ((nx->qkv2 && fuse_rb) || nx->q2h8)) {
            // the down projection of this layer and the q/k/v projection of the next one as one launch (decode.hip, dec_qkv2_kernel)
            if (!(skip & 16))
            { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * (2112.0 * 960.0 + 1536.0 * 576.0), (2112.0 * 960.0 + 1536.0 * 576.0) * (nx->q2h8 ? 1 : 4));
              if (nx->q2h8) launch_dec_qkv2_w8(da(4), nx->qkv8, nx->qkv_sc, nx->q2h8, nx->q2h_sc, w.dn8, w.dn_sc, s);
              else if (x3l) launch_dec_qkv2x3(da(4), nx->qkv2, w.down.p, s);
              else launch_dec_qkv2(da(4), nx->qkv2, w.down.p, s); }
        } else if (!(skip & 16))
        { ProfScope ps(e, PF_SKINNY, 2.0 * Bp * 576.0 * 1536.0, 576.0 * 1536.0 * 4);
          if (w.dn8) launch_dec_down(da(4), w.dn8, w.down.KP / 8, s, w.dn_sc);
          else launch_dec_down(da(4), w.down.p, w.down.KP / 8, s); }
    }
    return 0;
}
int enqueue_decode_layers(mellow_engine* e, int B, const RecordArgs* rec) {
    CHK(enqueue_decode_layer_range(e, B, 0, e->cfg.num_layers, true));
    CHK(run_lm_head(e, B, DEC_KC_DOWN, rec));
    return 0;
}

// audio1|audio2 are separate caller buffers: stage them into one [2B][n] batch so the encoder runs ONE pass
// of 2B clips (the reference runs two passes of B, mellow.py:105-106)
int encode_pair_to_prefix(mellow_engine* e, const float* a1, const float* a2, int64_t n_samples, const int32_t* ids,
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
void clear_bad_id(mellow_engine* e) { __atomic_store_n(e->h_progress + 1, 0ull, __ATOMIC_RELEASE); }

int check_bad_id(mellow_engine* e) {
    const unsigned long long w = __atomic_load_n(e->h_progress + 1, __ATOMIC_ACQUIRE);
    if (!w) return 0;
    return fail("index out of range in self: prompt id %d of example %u is outside the vocabulary [0, %d)", (int)(unsigned)(w & 0xffffffffu),
                (unsigned)((w >> 32) & 0x7fffffffu), e->cfg.vocab_size);
}

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

extern "C" {

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
    a.xn3 = nullptr;                                  // (the f32x3 kernel splits these rows itself)
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
    const bool no_migrate = !e->row_migration;   // option "row_migration" = 0: block exit without repacking (developer A/B)
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

}  // extern "C"
