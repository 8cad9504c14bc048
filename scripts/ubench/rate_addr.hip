// Does the rasteriser's rate depend on WHERE the frames go?  One engine (Breakout, 65 536 envs, fixed records), several output
// buffers and offsets, render-only loops interleaved over them.  (diagnostic; uses the product library through its C-ABI)
// usage: rate_addr [envs] [game id] [0 render only | 1 step;render] [offset scan KiB | 0] [pre-roll steps] [TBX_OPT_PIPELINE] [TBX_OPT_RENDER_SPLIT]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../include/toybox_amd.h"
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1;}}while(0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const int game = argc > 2 ? atoi(argv[2]) : TBX_GAME_BREAKOUT;
    const int with_step = argc > 3 ? atoi(argv[3]) : 0;
    tbx_engine* e = nullptr;
    if (tbx_create(game, n, 0, nullptr, 0, &e)) { printf("create failed: %s\n", tbx_last_error(nullptr)); return 1; }
    tbx_seed(e, -1, 1234); tbx_new_game(e, nullptr);
    int w = 0, h = 0; tbx_frame_dims(game, &w, &h);
    const size_t frame_bytes = (size_t)n * w * h * 3;
    hipStream_t s; CK(hipStreamCreate(&s));
    uint64_t t = 0;
    const int preroll = argc > 5 ? atoi(argv[5]) : 600;
    if (argc > 6) tbx_set_option(e, TBX_OPT_PIPELINE, atoi(argv[6]));
    if (argc > 7) tbx_set_option(e, TBX_OPT_RENDER_SPLIT, atoi(argv[7]));       // waves per frame
    for (int i = 0; i < preroll; i++) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s);
    struct Target { const char* name; uint8_t* p; };
    std::vector<Target> targets;
    targets.push_back({"engine-owned buffer", nullptr});
    const size_t slack = 96u << 20;
    static char names[64][64];
    const int scan = argc > 4 ? atoi(argv[4]) : 0;      // > 0: one allocation, offsets k * scan KiB
    if (scan > 0) {
        uint8_t* p; CK(hipMalloc((void**)&p, frame_bytes + slack));
        uint8_t* base = (uint8_t*)(((uintptr_t)p + (32u << 20) - 1) & ~(uintptr_t)((32u << 20) - 1));   // 32 MiB aligned
        for (int k = 0; k < 33; k++) {
            snprintf(names[k], 64, "32 MiB-aligned + %d KiB", k * scan);
            targets.push_back({names[k], base + (size_t)k * scan * 1024});
        }
    } else
        for (int b = 0; b < 4; b++) {
            uint8_t* p; CK(hipMalloc((void**)&p, frame_bytes + slack));
            const size_t offs[4] = {0, 4096, 2u << 20, (2u << 20) + 128 * 37};
            for (int o = 0; o < (b == 0 ? 4 : 1); o++) {
                snprintf(names[b * 4 + o], 64, "hipMalloc #%d + %zu (%p)", b, offs[o], (void*)(p + offs[o]));
                targets.push_back({names[b * 4 + o], p + offs[o]});
            }
        }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int round = 0; round < 2; round++)
        for (auto& tg : targets) {
            for (int i = 0; i < 10; i++) { if (with_step) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); tbx_render_device(e, tg.p, 3, s); }
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 60; i++) { if (with_step) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); tbx_render_device(e, tg.p, 3, s); }
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("round %d  %-52s %.4f ms  %.0f GB/s\n", round, tg.name, ms / 60, frame_bytes / (ms / 60) / 1e6);
            fflush(stdout);
        }
    tbx_destroy(e);
    return 0;
}
