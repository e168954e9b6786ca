// Developer microbenchmark (GPU box), round 6: the LM prefill attention alone (B examples x 3 kv groups x 13 query tiles, T = 389
// keys, Tmax = 512) on random q / K / V pages: us per launch, and a checksum of the output (variants that must be bit-identical
// print the same one).  Build one binary per variant:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMELLOW_PAX_ABL=bits] -o tools/microbench/pab_<name>.bin tools/microbench/prefill_attn_bench.hip
#include "../../mellow_amd/csrc/prefill_attn.hip"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int T = 389, Tmax = 512;
    for (int B : {16, 32}) {
        const size_t nq = (size_t)B * T * 576, nkv = (size_t)B * 3 * Tmax * 64;
        std::vector<float> hq(nq), hk(nkv), hv(nkv);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& x : hq) x = rnd();
        for (auto& x : hk) x = rnd();
        for (auto& x : hv) x = rnd();
        float *q, *k, *v, *o; void* o3;
        CK(hipMalloc(&q, nq * 4)); CK(hipMalloc(&k, nkv * 4)); CK(hipMalloc(&v, nkv * 4)); CK(hipMalloc(&o, nq * 4)); CK(hipMalloc(&o3, nq * 6 + (1 << 20)));
        CK(hipMemcpy(q, hq.data(), nq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), nkv * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, hv.data(), nkv * 4, hipMemcpyHostToDevice));
        hipStream_t st; CK(hipStreamCreate(&st));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int x3 = 2; x3 >= (argc > 1 ? 0 : 1); --x3) {      // 2 = operands rounded once to bf16 (fp8 mode), 1 = exact 3-way split, 0 = fp32 MFMA
            for (int i = 0; i < 5; ++i) mellow::launch_prefill_attention(q, k, v, o, nullptr, B, T, Tmax, x3 != 0, st, nullptr, x3 == 2);
            CK(hipStreamSynchronize(st));
            float best = 1e9f, sum = 0.f;
            const int R = 40;
            for (int i = 0; i < R; ++i) {
                CK(hipEventRecord(a, st));
                mellow::launch_prefill_attention(q, k, v, o, nullptr, B, T, Tmax, x3 != 0, st, nullptr, x3 == 2);
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; best = ms < best ? ms : best;
            }
            std::vector<float> ho(nq);
            CK(hipMemcpy(ho.data(), o, nq * 4, hipMemcpyDeviceToHost));
            uint64_t cs = 0; double l1 = 0;
            for (size_t i = 0; i < nq; ++i) { uint32_t u; memcpy(&u, &ho[i], 4); cs = cs * 1099511628211ull + u; l1 += fabs(ho[i]); }
            printf("B=%d %s abl=%d: avg %.1f us, min %.1f us, checksum %016llx, mean|o| %.6f\n", B, x3 == 2 ? "bf16" : x3 ? "x3" : "f32", MELLOW_PAX_ABL, sum / R * 1e3, best * 1e3,
                   (unsigned long long)cs, l1 / nq);
        }
        hipFree(q); hipFree(k); hipFree(v); hipFree(o); hipFree(o3);
    }
    return 0;
}
