// start time of every rasteriser wave after [nothing] and after [step] (exp2 build: one 8-byte store per wave, nothing else)
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
    uint64_t* tr; CK(hipMalloc((void**)&tr, waves * 8)); CK(hipMemset(tr, 0, waves * 8));
    const bool trace_on = argc > 2 ? atoi(argv[2]) != 0 : true;
    if (trace_on && tbx_exp_trace(tr)) { printf("trace set failed\n"); return 1; }
    std::vector<uint64_t> h(waves);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int round = 0; round < 2; round++)
        for (int p = 0; p < 2; p++) {
            for (int i = 0; i < 30; i++) { if (p) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); tbx_render_device(e, nullptr, 3, s); }
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 50; i++) { if (p) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); tbx_render_device(e, nullptr, 3, s); }
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("round %d %-14s loop %.4f ms/iter\n", round, p ? "step;render" : "render only", ms / 50);
            if (!trace_on) continue;
            CK(hipMemcpy(h.data(), tr, waves * 8, hipMemcpyDeviceToHost));
            std::vector<uint64_t> st(h); std::sort(st.begin(), st.end());
            const uint64_t w_min = st[0];
            printf("   last wave starts at %.1f us; starts per 50-us bucket:", (st[waves - 1] - w_min) * 0.01);
            size_t k = 0; for (uint64_t edge = 5000; k < waves; edge += 5000) { size_t c = 0; while (k < waves && st[k] - w_min < edge) { k++; c++; } printf(" %zu", c); }
            printf("\n   first 20 us, starts per 1-us bucket:");
            k = 0; for (uint64_t edge = 100; edge <= 2000; edge += 100) { size_t c = 0; while (k < waves && st[k] - w_min < edge) { k++; c++; } printf(" %zu", c); }
            // how far out of launch order do waves start?  mean |rank by start time - wave id| over the kernel
            printf("\n");
        }
    tbx_destroy(e);
    return 0;
}
