// Developer entry points (not used by the product path): GEMM timing and debug taps, profiler dump, kernel stamps.
#include "engine_internal.h"

extern "C" {

// developer instrumentation (not part of the public header): time `iters` launches of one plain GEMM shape on
// synthetic device buffers (garbage-in; EPI_LINEAR, no bias) -> average milliseconds per launch
__attribute__((visibility("default"))) int mellow_dev_gemm_time(mellow_engine_t* e, int M, int N, int K, int iters, float* ms_out) {
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

// one fp32 GEMM C[M][N] = A[M][K] . W[N][K]^T on host data through the exact fp32 MFMA kernel (mode 0) or the engine's f32x3
// kernels (mode 16: A split in registers; mode 17: A pre-split in APB order, both operands by LDS-DMA): the accuracy tap of
// include/mellow_hip.h
int mellow_debug_gemm_f32(mellow_engine_t* e, int mode, const float* A, int M, int K, const float* W, int N, float* C_out,
                          int iters, float* ms2) {
    if (!e || !A || !W || M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4) return fail("bad argument");
    if (mode != 0 && mode != 16 && mode != 17) return fail("mode must be 0 (fp32 MFMA), 16 (f32x3, A split in registers) or 17 (f32x3, pre-split A, LDS-DMA)");
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
        else { if (pre) launch_split_rows_apb(dA, K, M, K, dA3, s); if (main) launch_gemm_bf16x3_apb(g, s); }
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
    if (!e || !A || !W || M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int NP = rup(N, 128), K64 = rup(K, 64), Mp = rup(M, 128);
    float *dA = nullptr, *dW = nullptr, *dWp = nullptr, *dC = nullptr, *dsa = nullptr, *dsw = nullptr;
    uint8_t *dA8 = nullptr, *dW8 = nullptr;
    HIPCHK(hipMalloc(&dA, (size_t)M * K * 4));
    HIPCHK(hipMalloc(&dW, (size_t)N * K * 4));
    HIPCHK(hipMalloc(&dWp, (size_t)NP * K * 4));
    HIPCHK(hipMalloc(&dC, (size_t)M * N * 4));
    HIPCHK(hipMalloc(&dsa, (size_t)Mp * ((K64 / 64 + 3) / 4) * 8));       // 2 scale words per row and four k64 steps
    HIPCHK(hipMalloc(&dsw, (size_t)NP * 4));
    HIPCHK(hipMalloc(&dA8, (size_t)Mp * K64));
    HIPCHK(hipMalloc(&dW8, (size_t)NP * K64));
    HIPCHK(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dW, W, (size_t)N * K * 4, hipMemcpyHostToDevice));
    launch_pack_weight(dW, N, K, K, dWp, NP, K, s);
    launch_pack_fp8(dWp, NP, K, dW8, dsw, s);
    GemmArgs g;
    g.A = dA; g.lda = K; g.M = M; g.K = K; g.Wp = dWp; g.Nw = N; g.N = N; g.C = dC; g.ldc = N;
    g.A8 = dA8; g.lda8 = K64; g.a_sc = reinterpret_cast<const uint32_t*>(dsa); g.W8 = dW8; g.w_scale = dsw;
    launch_quant_mx8(dA, K, M, K, dA8, dsa, s);
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
        for (int i = 0; i < iters; ++i) launch_quant_mx8(dA, K, M, K, dA8, dsa, s);
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
__attribute__((visibility("default"))) int mellow_dev_prof_dump(mellow_engine_t* e, const char* path) {
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
__attribute__((visibility("default"))) int mellow_dev_kdebug(mellow_engine_t* e, int on, uint64_t* host_out64) {
    if (!e) return fail("null engine");
    static uint64_t* buf = nullptr;
    HIPCHK(hipSetDevice(e->device));
    if (on) {
        // [0, 64): phase stamps of workgroup 0; [64, 1024): (earliest start, latest end) of every launch of a decode step
        // (DecArgs::dbg_seq, 100 MHz clock), the starts initialised to the maximum
        if (!buf) HIPCHK(hipMalloc(&buf, 1024 * sizeof(uint64_t)));
        std::vector<uint64_t> init(1024, 0);
        for (int i = 64; i < 1024; i += 2) init[i] = ~0ull;
        HIPCHK(hipMemcpy(buf, init.data(), 1024 * sizeof(uint64_t), hipMemcpyHostToDevice));
        set_kernel_debug_buffer(buf);
        set_gemm_debug_buffer(buf);
        e->dbg_seq0 = 0;
    } else {
        HIPCHK(hipStreamSynchronize(e->stream));
        e->dbg_seq0 = -1;
        // (host_out64 receives the 64 phase stamps; mellow_dev_kdebug_spans the launch spans)
        if (buf && host_out64) HIPCHK(hipMemcpy(host_out64, buf, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (buf) HIPCHK(hipMemcpy(e->dbg_spans, buf + 64, 960 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        set_kernel_debug_buffer(nullptr);
        set_gemm_debug_buffer(nullptr);
    }
    return 0;
}

// the launch spans collected by the last mellow_dev_kdebug(on) .. (off) bracket: out[2 i] / out[2 i + 1] = earliest start / latest
// end (10 ns ticks) of launch i of the LAST decode step that ran inside the bracket; 480 pairs
__attribute__((visibility("default"))) int mellow_dev_kdebug_spans(mellow_engine_t* e, uint64_t* out960) {
    if (!e || !out960) return fail("null argument");
    memcpy(out960, e->dbg_spans, 960 * sizeof(uint64_t));
    return 0;
}

}  // extern "C"
