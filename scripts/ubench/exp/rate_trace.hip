// per-wave timing of the Breakout rasteriser after [nothing] and after [step] (exp build only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../../include/toybox_amd.h"
extern "C" int tbx_exp_trace(uint64_t* p);
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1;}}while(0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    tbx_engine* e = nullptr;
    if (tbx_create(TBX_GAME_BREAKOUT, n, 0, nullptr, 0, &e)) { printf("create failed\n"); return 1; }
    tbx_seed(e, -1, 1234); tbx_new_game(e, nullptr);
    hipStream_t s; CK(hipStreamCreate(&s));
    uint64_t t = 0;
    for (int i = 0; i < 600; i++) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s);
    const size_t waves = (size_t)n * 10;
    uint64_t* tr; CK(hipMalloc((void**)&tr, waves * 32)); CK(hipMemset(tr, 0, waves * 32));
    const bool trace_on = argc > 2 ? atoi(argv[2]) != 0 : true;
    if (trace_on && tbx_exp_trace(tr)) { printf("trace set failed\n"); return 1; }
    std::vector<uint64_t> h(waves * 4);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int round = 0; round < 2; round++)
        for (int p = 0; p < 2; p++) {
            for (int i = 0; i < 30; i++) { if (p) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); tbx_render_device(e, nullptr, 3, s); }
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 50; i++) { if (p) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); tbx_render_device(e, nullptr, 3, s); }
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            CK(hipMemcpy(h.data(), tr, waves * 32, hipMemcpyDeviceToHost));
            uint64_t w_min = ~0ull, w_end = 0; double sum_ld = 0, sum_pt = 0, sum_dur = 0; std::vector<uint64_t> ld(waves), dur(waves);
            double xs[8] = {0}; size_t xn[8] = {0};
            for (size_t w = 0; w < waves; w++) {
                const uint64_t w0 = h[4 * w], d = h[4 * w + 3] & 0xFFFFFFFFFFFFFFull; const int x = (int)(h[4 * w + 3] >> 56) & 7;
                w_min = std::min(w_min, w0); w_end = std::max(w_end, w0 + d);
                ld[w] = h[4 * w + 1]; dur[w] = d; sum_ld += (double)h[4 * w + 1]; sum_pt += (double)h[4 * w + 2]; sum_dur += (double)d; xs[x] += (double)d; xn[x]++;
            }
            std::sort(ld.begin(), ld.end()); std::sort(dur.begin(), dur.end());
            printf("round %d %-14s loop %.4f ms/iter | kernel span %.1f us | record load cycles mean %.0f p50 %llu p90 %llu p99 %llu max %llu | paint cycles mean %.0f | wave wall (10 ns ticks) mean %.0f p50 %llu p90 %llu p99 %llu\n",
                   round, p ? "step;render" : "render only", ms / 50, (w_end - w_min) * 0.01, sum_ld / waves, (unsigned long long)ld[waves / 2], (unsigned long long)ld[waves * 9 / 10],
                   (unsigned long long)ld[waves * 99 / 100], (unsigned long long)ld[waves - 1], sum_pt / waves, sum_dur / waves, (unsigned long long)dur[waves / 2],
                   (unsigned long long)dur[waves * 9 / 10], (unsigned long long)dur[waves * 99 / 100]);
            printf("   per-XCD mean wave wall:"); for (int x = 0; x < 8; x++) printf(" %d:%.0f(%zu)", x, xn[x] ? xs[x] / xn[x] : 0.0, xn[x]); printf("\n");
            // waves started / mean wall per 100-us bucket of start time
            const int NB = 16; double bs[NB] = {0}; size_t bn[NB] = {0}; double bl[NB] = {0};
            for (size_t w = 0; w < waves; w++) { int k = (int)((h[4 * w] - w_min) / 10000); if (k >= NB) k = NB - 1; bn[k]++; bs[k] += (double)(h[4 * w + 3] & 0xFFFFFFFFFFFFFFull); bl[k] += (double)h[4 * w + 1]; }
            printf("   by start time (100 us buckets): waves / mean wall ticks / mean load cycles\n     ");
            for (int k = 0; k < NB; k++) if (bn[k]) printf(" [%d] %zu/%.0f/%.0f", k, bn[k], bs[k] / bn[k], bl[k] / bn[k]);
            printf("\n");
        }
    tbx_destroy(e);
    return 0;
}
