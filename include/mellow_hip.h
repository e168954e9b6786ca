/*
 * mellow_hip.h — C ABI of libmellow_hip.so, the MI355X (gfx950) engine behind MellowWrapper.generate().
 *
 * This is the drop-in boundary for the reference's inference hot path.  The reference has no native
 * interface (it is pure Python on PyTorch ATen); the seams this library replaces are the Python calls
 * listed per function below (file:line in soham97/mellow).  Nothing in these signatures is a torch
 * type: plain pointers, sizes and scalars, so the same library binds from ctypes (what
 * mellow_amd/engine.py does), cffi, pybind11 or cgo.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; mellow_last_error() then returns a
 *     thread-local message.  The Python wrapper re-raises the reference's exception types
 *     (ValueError / AssertionError / RuntimeError) from it.
 *   - "dev" pointers are HIP device pointers on the engine's device (caller-owned; a torch tensor's
 *     data_ptr() is fine).  "host" pointers are ordinary host memory.
 *   - all tensors are dense, row-major, fp32 unless marked int32.
 *   - calls on one engine are serialised by the caller (the reference is single-threaded, blocking:
 *     wrapper.py:212 `torch.no_grad()`, no re-entrancy).  Work is issued on the engine's private HIP
 *     stream; every data-path call returns after that stream has drained unless stated otherwise.
 *   - the engine owns weights, KV pages and workspaces; the caller owns inputs and outputs.
 */
#ifndef MELLOW_HIP_H
#define MELLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared between these two pragmas are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* 2: out_tokens may hold -1 (never-computed steps), first_token_ms is host wall-clock time, a decode step needs a prefill of
 *    its own after mellow_generate / mellow_lm_forward_logits, the optional "mellow.rope_cos/sin" tensors, new symbols */
#define MELLOW_ABI_VERSION 2
/* Minor revisions (same struct layouts and symbol meanings: a version-2 caller keeps working; mellow_abi_minor() reports it):
 *  1: the DEFAULT numeric mode of mellow_engine_create is MELLOW_PRECISION_F32X3 (round 4).  A caller that never calls
 *     mellow_engine_set_precision gets fp32-accurate GEMMs whose LAST BITS depend on how many examples share the call
 *     (split-K for small launches; from round 5 the decode step of a batch of more than 32 rows runs other kernels than one of
 *     up to 32 rows; from round 6 its attention merges one key split instead of two from two row blocks on): greedy tokens are
 *     asserted equal across those forms on the reference's fixtures, logits agree to 1e-3.  A caller that needs
 *     batch-size-independent GEMMs selects MELLOW_PRECISION_F32: encoder and prefill are then bit-identical whatever the batch;
 *     the decode step still picks wave counts per launch by the number of 32-row blocks (another fp32 summation order: <= 5e-4
 *     on the logits, same arg-max: tests/test_gpu_parity.py test_exact_fp32_mode_across_batch_sizes).
 *  2: mellow_prefill_parts (round 5).
 *  3: mellow_engine_set_option / mellow_engine_describe (round 6): the library no longer reads ANY environment variable; every
 *     switch it has is a named option, and the resolved configuration can be printed.  mellow_debug_gemm_f32 modes 6 / 9 are gone
 *     (the pre-split debug kernel was removed; 16 / 17 are the engine's own f32x3 kernels). */
#define MELLOW_ABI_MINOR 3

typedef struct mellow_engine mellow_engine_t;

/* Decoder LM hyper-parameters (reference: HF config.json of SmolLM2-135M fetched at decoder.py:25;
 * kept as data in mellow_amd/config/lm_smollm2_135m.yaml) + encoder/prefix constants of v0.yaml. */
typedef struct mellow_config {
    int32_t abi_version;        /* must be MELLOW_ABI_VERSION */
    int32_t vocab_size;         /* 49152 */
    int32_t hidden_size;        /* 576  (must equal encoder d_proj, wrapper.py:62) */
    int32_t intermediate_size;  /* 1536 */
    int32_t num_layers;         /* 30 */
    int32_t num_heads;          /* 9 */
    int32_t num_kv_heads;       /* 3 */
    int32_t head_dim;           /* 64 */
    float   rms_norm_eps;       /* 1e-5 */
    float   rope_theta;         /* 100000 */
    int32_t max_positions;      /* rows of the RoPE table to build (>= 389 + max_len) */
    int32_t text_len;           /* 129 (v0.yaml text_tokenization_len) */
    int32_t prefix_len;         /* 389 (v0.yaml prefix_length) */
    int32_t sep_token_id;       /* 0 (decoder.py:49) */
} mellow_config_t;

enum { MELLOW_F32 = 0, MELLOW_I32 = 1, MELLOW_I64 = 2 };

/* ---- library ------------------------------------------------------------------------------------ */
int         mellow_abi_version(void);
const char* mellow_last_error(void);
/* number of HIP devices visible (0 when there is no GPU; never fails) */
int         mellow_device_count(void);

/* ---- lifetime: replaces MellowWrapper.get_model_and_tokenizer's model construction + load_state_dict
 *      + model.to(cuda) (reference wrapper.py:59-88) ------------------------------------------------ */
int  mellow_engine_create(const mellow_config_t* cfg, int device, mellow_engine_t** out);
void mellow_engine_destroy(mellow_engine_t* e);
/* A second execution context on the same device that SHARES a finalized engine's weights (no copy): its own HIP stream,
 * workspaces, KV pages and captured graphs.  Calls on `parent` and on the fork may overlap from different host threads (a
 * serving front-end pipelines independent batches this way: mellow_amd/serve.py); each handle is still serialised by its
 * caller.  Destroy every fork before its parent.  mellow_engine_fork / the fork's mellow_engine_destroy touch the PARENT (its
 * fork count; while it has forks the parent runs its LM prefill as one chain instead of two half-batches on two streams, and it
 * goes back to the split form when its last fork is destroyed): do not call them while a call is running on the parent.
 * On failure nothing is leaked.  The reference has no counterpart (one synchronous model object). */
int  mellow_engine_fork(mellow_engine_t* parent, mellow_engine_t** out);

/* Hand one checkpoint tensor to the engine under its reference state_dict key, e.g.
 * "audio_encoder.base.htsat.layers.0.blocks.1.attn.qkv.weight".  `data` may be a host or a device
 * pointer; the engine copies / re-tiles it into its own arena before returning.  Keys the inference
 * path never reads are accepted and ignored (returns 0).  Unknown keys fail.
 *
 * Two OPTIONAL tensors that are not checkpoint keys: "mellow.rope_cos" and "mellow.rope_sin", f32 [max_positions][head_dim/2],
 * the rotary tables cos / sin(position * inv_freq).  The reference never stores them: transformers' LlamaRotaryEmbedding
 * recomputes them with torch on every forward, so their last bit is whatever torch's vectorised cos/sin give on the host it
 * runs on.  A binding that wants the engine to use EXACTLY the numbers its own torch would produce computes them the HF way
 *  and loads them here before finalize.  Without them the engine builds the tables itself
 * (mellow_host_rope_tables: fp32 inv_freq and angle as in HF, cos/sin evaluated in double and rounded to fp32 -- within
 * 1 ulp of torch's, tests/test_abi_cpu.py); both paths are parity-tested for 300 decode steps. */
int  mellow_engine_load_tensor(mellow_engine_t* e, const char* key, const void* data,
                               const int64_t* shape, int ndim, int dtype);
/* Verifies that every tensor the hot path reads has been loaded (strict, like load_state_dict at
 * wrapper.py:76) and builds derived tables (expanded relative-position bias, window maps, RoPE). */
int  mellow_engine_finalize(mellow_engine_t* e);
/* number of state_dict keys the engine requires / name of the i-th one (for loader validation) */
int         mellow_engine_num_required(void);
const char* mellow_engine_required_key(int i);

/* ---- the hot path: replaces Mellow.generate_prefix_inference (mellow.py:100-108, called
 *      wrapper.py:285) + MellowWrapper._generate_batch (wrapper.py:197-249) -------------------------
 * audio1/audio2 : dev f32 [B][n_samples]  (what preprocess_audio returns, wrapper.py:170-179)
 * input_ids     : dev i32 [B][text_len]   (preprocess_text's input_ids, wrapper.py:181-195)
 * max_len       : entry_length of the loop (wrapper.py:200)
 * top_p, temperature : accepted for API parity; the reference's filter never removes the arg-max
 *                 (wrapper.py:220-232) so the result is greedy for every value
 * stop_id       : tokenizer.encode(stop_token)[0] (wrapper.py:208)
 * ignore_stop   : 0 = reference semantics (loop ends when every row has produced stop_id,
 *                 wrapper.py:247-249); 1 = always run max_len steps (fixed-work benchmark mode)
 * out_tokens    : dev i32 [B][max_len]; columns >= *out_steps are undefined.  In reference-semantics mode with more than
 *                 one 32-row block, a block whose rows have ALL produced stop_id stops being computed (per-block early
 *                 exit): its rows hold -1 in the columns after that step (every such row's text is already cut)
 * out_len       : host i32 [B], tokens before the row's first stop_id (the text cut of wrapper.py:254)
 * out_steps     : host, number of loop iterations the reference would have run
 * first_token_ms: host wall-clock milliseconds from call entry until the first token id of every row exists
 *                 (observed through the device's progress word, without synchronising the stream)
 * In reference-semantics mode the host follows the stop rule one step behind the device through a mapped progress
 * word the arg-max kernel publishes: no stream synchronisation inside the loop, at most one step is enqueued past
 * the deciding one (mellow_last_steps_enqueued reports how many were).
 */
int  mellow_generate(mellow_engine_t* e, const float* audio1, const float* audio2, int64_t n_samples,
                     const int32_t* input_ids, int B, int max_len, float top_p, float temperature,
                     int stop_id, int ignore_stop, int32_t* out_tokens, int32_t* out_len,
                     int32_t* out_steps, float* first_token_ms);

/* ---- parity taps (same kernels as mellow_generate, stage by stage) ------------------------------- */
/* A1-A3: htsat.py:864-870.  wav dev [n][n_samples] -> out dev [n][frames][64]; apply_bn=0 gives the
 * LogmelFilterBank output, 1 the post-bn0 tensor. */
int  mellow_logmel(mellow_engine_t* e, const float* wav, int n_clips, int64_t n_samples, int apply_bn,
                   float* out);
/* A1-A13: AudioEncoder.forward (mellow.py:64-68) + downsample (decoder.py:14-18):
 * wav dev [n][n_samples] -> out dev [n][129][576]. */
int  mellow_encode(mellow_engine_t* e, const float* wav, int n_clips, int64_t n_samples, float* out);
/* A1-A14: generate_prefix_inference -> out dev [B][prefix_len][hidden]. */
int  mellow_prefix(mellow_engine_t* e, const float* audio1, const float* audio2, int64_t n_samples,
                   const int32_t* input_ids, int B, float* out);
/* A15 prefill: lm(inputs_embeds=prefix).logits[:, -1, :] (wrapper.py:217-218) with the KV pages
 * written.  prefix dev [B][T][hidden]; reserve = max extra tokens that will follow; logits dev [B][vocab]
 * (may be NULL). */
int  mellow_lm_prefill(mellow_engine_t* e, const float* prefix, int B, int T, int reserve, float* logits);
/* A15 decode step: append embed_tokens(token_ids) (wrapper.py:237) at the next position and return the
 * new last-position logits.  token_ids dev i32 [B]; logits dev [B][vocab] (may be NULL).  Continues the state of the last
 * mellow_lm_prefill; a mellow_generate or mellow_lm_forward_logits call in between ends that state (error). */
int  mellow_lm_decode_step(mellow_engine_t* e, const int32_t* token_ids, float* logits);
/* lm.model.embed_tokens(ids) (decoder.py:47,64-66; wrapper.py:237): token_ids dev i32 [n] -> out dev [n][hidden]. */
int  mellow_embed_tokens(mellow_engine_t* e, const int32_t* token_ids, int n, float* out);
/* The decoder's forward over a whole embedded sequence, `self.lm(inputs_embeds=embedding_cat).logits` of the training-time
 * forward (Mellow.forward mellow.py:89-98 -> DecoderModel.forward decoder.py:57-90, which concatenates the prefix with the
 * embedded answer tokens): embeds dev [B][T][hidden] -> logits dev [B][T - from_pos][vocab], the rows of positions
 * t >= from_pos.  Inference arithmetic only (no loss, no gradient); leaves no decode state behind. */
int  mellow_lm_forward_logits(mellow_engine_t* e, const float* embeds, int B, int T, int from_pos, float* logits);
/* A0 (host harness of the reference, wrapper.py:146 `torchaudio.transforms.Resample(sr, 32000)`) on the device:
 * sinc-interpolation resampling with a Hann window, lowpass_filter_width 6, rolloff 0.99, gcd-reduced polyphase bank.
 * wav dev [n][n_in] -> out dev [n][*n_out], *n_out = ceil(new_freq * n_in / orig_freq); out == NULL only queries *n_out. */
int  mellow_resample(mellow_engine_t* e, const float* wav, int n_clips, int64_t n_in, int orig_freq, int new_freq,
                     float* out, int64_t out_capacity, int64_t* n_out);
/* arg-max with first-index ties (torch.argmax, wrapper.py:232): logits dev [B][vocab] -> tokens dev i32 [B] */
int  mellow_argmax(mellow_engine_t* e, const float* logits, int B, int32_t* tokens);

/* Copy a named internal activation of the LAST mellow_encode / mellow_prefix call to `out` (dev).
 * Taps are recorded only after mellow_debug_enable_taps(e, 1).  Names: "power", "logmel_bn", "patch",
 * "stage0".."stage3", "latent", "fpx", "emb33", "proj33".  *numel receives the element count. */
int  mellow_debug_enable_taps(mellow_engine_t* e, int on);
int  mellow_debug_tap(mellow_engine_t* e, const char* name, float* out, int64_t capacity, int64_t* numel);

/* Numeric taps on HOST data (work on any engine, finalised or not): C[M][N] = A[M][K] . W[N][K]^T through one GEMM kernel.
 * iters > 0 and ms2 != NULL: ms2[0] / ms2[1] receive the average milliseconds of the operand pre-pass / of the GEMM.
 *   mellow_debug_gemm_f32: mode 0 = the exact fp32 MFMA kernel (no pre-pass), 9 / 6 = the bf16x3 kernel on pre-split rows with
 *     all nine / the six largest partial products, 16 = the fused six-product kernel MELLOW_PRECISION_F32X3 runs (A split in
 *     registers, no pre-pass).  K % 32 == 0, N % 4 == 0.
 *   mellow_debug_gemm_fp8: the MELLOW_PRECISION_FP8 GEMM -- A quantised per row and W per row to OCP e4m3 (scale =
 *     amax / 448, round to nearest even), exact products, fp32 accumulation.  K % 64 == 0, N % 4 == 0. */
int  mellow_debug_gemm_f32(mellow_engine_t* e, int mode, const float* A, int M, int K, const float* W, int N, float* C,
                           int iters, float* ms2);
int  mellow_debug_gemm_fp8(mellow_engine_t* e, const float* A, int M, int K, const float* W, int N, float* C,
                           int iters, float* ms2);
/* The decode step's lm_head kernel on caller-supplied rows (dev): logits[B][vocab] = x[B][hidden] . lm_head^T with the engine's
 * own head weights (the e4m3 copy in MELLOW_PRECISION_FP8; act_fp8 != 0 then also quantises x inside the kernel: one scale per
 * batch row and 72-column slice, fp8 matrix pipe).  Invalidates the decode state of an earlier prefill. */
int  mellow_debug_dec_head(mellow_engine_t* e, const float* x, int B, int act_fp8, float* logits);

/* (The library also exports three developer instrumentation entry points that are NOT part of this ABI and may change without
 *  a version bump: mellow_dev_gemm_time, mellow_dev_prof_dump, mellow_dev_kdebug -- timers and stamps used by tools/, none of
 *  them changes a result.) */

/* ---- measurement ---------------------------------------------------------------------------------
 * Per-kernel-family accounting with HIP events on the engine's stream.  When enabled, every launch
 * of a profiled family is bracketed by an event pair and its algorithmic work is accumulated;
 * hipGraph replay is switched off while profiling (events cannot be interleaved into a replay).
 * families: see mellow_prof_family_name(). */
int         mellow_prof_enable(mellow_engine_t* e, int on);
int         mellow_prof_reset(mellow_engine_t* e);
int         mellow_prof_num_families(void);
const char* mellow_prof_family_name(int i);
/* launches, total milliseconds, algorithmic flops and algorithmic bytes accumulated for family i */
int         mellow_prof_get(mellow_engine_t* e, int i, int64_t* launches, double* ms, double* flops,
                            double* bytes);
/* phase wall times of the last mellow_generate call (HIP events): front-end+encoder+prefix, prefill,
 * decode loop; milliseconds */
int         mellow_last_phase_ms(mellow_engine_t* e, float* encode_ms, float* prefill_ms, float* decode_ms);
/* decode steps (counting the prefill's token) the last mellow_generate call enqueued: *out_steps, or *out_steps + 1 */
int         mellow_last_steps_enqueued(mellow_engine_t* e);
/* Diagnostic: how many times the last mellow_generate call repacked the still-running rows into fewer 32-row blocks
 * (reference-semantics mode, B > 32: a block's kernels return at once when all of ITS rows have produced the stop id; rows
 * migrate between blocks so that this happens as early as the number of running rows allows). */
int         mellow_last_row_repacks(mellow_engine_t* e);
/* 1 when this engine runs the STFT (A1, htsat.py:864) as a 1024-point FFT per frame instead of the DFT GEMM: f32x3 mode and a
 * checkpoint whose conv_real / conv_imag weights are window[n] * cos / sin(2 pi k n / 1024) to 1e-6 (checked at finalize). */
int         mellow_stft_is_fft(mellow_engine_t* e);
/* Number of independent parts the next f32x3 LM prefill of this engine runs as (2 by default: two half-batches on two HIP
 * streams, so that one part's kernel tails are covered by the other's).  The engine OWNS this configuration: the first call
 * (this function or the first prefill) creates the side stream(s) and MEASURES with a device-clock probe whether they really
 * run beside the main stream -- HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says
 * otherwise when the runtime starts), and two streams that share a queue serialise.  A side stream that does not overlap is
 * replaced (up to 8 attempts); if none does, the engine falls back to ONE chain and this function returns 1.  Results are
 * bit-identical for every value; only the speed differs (the reference has no counterpart: wrapper.py:87-88 is its device model). */
int         mellow_prefill_parts(mellow_engine_t* e);
/* MELLOW_ABI_MINOR of the library */
int         mellow_abi_minor(void);
/* 1 = replay the decode step from a captured hipGraph (default), 0 = eager launches */
int         mellow_set_graph(mellow_engine_t* e, int on);

/* ---- numeric mode of the dense GEMMs of the encoder (Swin linears) and of LM prefill; call before the first
 *      mellow_engine_load_tensor.  The reference has no counterpart (fp32 ATen matmuls throughout).
 *   MELLOW_PRECISION_F32: exact fp32 on v_mfma_f32_32x32x2_f32 everywhere (fmaf-chain accumulation, the closest arithmetic to the
 *     reference's ATen matmuls); the parity suite runs in this mode AND in the default one with the same tolerances.
 *   MELLOW_PRECISION_FP8: BASELINE config 5 -- OCP e4m3 weights (per-output-channel scale) and activations (per-row
 *     scale, quantised on the fly), fp32 accumulate on v_mfma_f32_32x32x16_fp8_fp8; the GEMM kernels of the decode step read
 *     e4m3 weights and quantise their activations in registers (one scale per batch row and wave k-slice).  Front-end (STFT,
 *     mel), K % 64 != 0 layers, the attentions and the norms stay fp32.  Not bit-exact: report token agreement.
 *   MELLOW_PRECISION_F32X3 (DEFAULT of mellow_engine_create, of the Python `Engine` / `MellowWrapper`, and the mode bench.py
 *     reports): fp32 GEMMs on the bf16 matrix pipe -- every fp32 operand is split EXACTLY into three bf16 terms, the six largest
 *     partial products (the rest is < 2^-23 |a*b|) are accumulated in fp32: fp32-accurate (error against fp64 measured <= the
 *     fp32 MFMA kernel's), not bit-identical to MELLOW_PRECISION_F32.  Every engine-level parity test (tolerances against the
 *     reference's fp32 outputs, exact greedy tokens) runs in this mode and in MELLOW_PRECISION_F32.  In this mode the last bits
 *     of an example's activations may depend on how many examples the call holds: encoder launches of few output tiles are split
 *     along K (a fixed summation order per launch shape; option "splitk" = 0 turns it off). */
#define MELLOW_PRECISION_F32 0
#define MELLOW_PRECISION_FP8 1
#define MELLOW_PRECISION_F32X3 2
int         mellow_engine_set_precision(mellow_engine_t* e, int mode);

/* ---- explicit configuration (round 6).  The library reads no environment variable: a consumer's environment cannot change its
 *      arithmetic.  Every switch is a named integer option, set after mellow_engine_create and before mellow_engine_finalize
 *      ("arena_mb": before the first mellow_engine_load_tensor); an unknown key or a malformed value is an error.  The defaults
 *      are the configuration bench.py measures and the parity suite runs; the reference has no counterpart (wrapper.py:35-49 has
 *      no hidden modes, and neither has a default engine).  Keys: prefill_split, prefill_fuse_norm, decode_fuse,
 *      decode_fuse_max_rb, splitk, enc_apb, graph, fp8_decode, fp8_prefill, fp8_decode_act, fp8_kv16, fp8_attn_bf16, decode_x3, decode_x3_min_rb,
 *      x3_stft, stft_fft, x3_apb, x3_attn, x3w, row_migration, arena_mb (mellow_amd/csrc/engine.cpp: kOptions says what each does
 *      and whether it can change the answers). */
int         mellow_engine_set_option(mellow_engine_t* e, const char* key, const char* value);
/* JSON text: {"abi": [major, minor], "precision": "f32x3" | "f32" | "fp8", "reads_environment": false, "finalized": ...,
 *  "stft_is_fft": ..., "non_default": [keys], "options": {key: {"value", "default", "changes_answers"}}}.  Returns the bytes the
 *  text needs including its terminator (-1 for a null engine); writes it only when `capacity` suffices (call with buf = NULL
 *  to size). */
int64_t     mellow_engine_describe(mellow_engine_t* e, char* buf, int64_t capacity);

/* ---- host-only helpers (callable without a GPU; used by CPU tests) -------------------------------- */
/* token permutation of a Swin block: out[m] = source token (h*R+w) feeding window-order row m, for
 * resolution R and cyclic shift `shift` (htsat.py:427-436).  out host i32 [R*R]. */
int  mellow_host_window_map(int R, int shift, int32_t* out);
/* packs a row-major [N][K] fp32 matrix into the engine's MFMA fragment order:
 * out host f32 [NP/32][KP/8][64][4] with NP = roundup(N,npad), KP = roundup(K,32), zero padded. */
int  mellow_host_pack_weight(const float* w, int N, int K, int npad, float* out, int64_t out_capacity);
/* the rotary tables the engine builds when "mellow.rope_cos/sin" are not loaded (transformers LlamaRotaryEmbedding:
 * inv_freq = 1 / theta^(2i/head_dim) and angle = position * inv_freq in fp32; cos / sin correctly rounded to fp32).
 * cos_out / sin_out host f32 [max_pos][head_dim/2]. */
int  mellow_host_rope_tables(float theta, int head_dim, int max_pos, float* cos_out, float* sin_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MELLOW_HIP_H */
