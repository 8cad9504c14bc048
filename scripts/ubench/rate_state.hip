// What, queued between two rasteriser launches, puts the loop into the slower of its two rates?  Uses the product library through
// its C-ABI (Breakout, 65 536 envs) and times 100-iteration loops of [X ; render] for several X.  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../include/toybox_amd.h"
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1;}}while(0)
__global__ void empty_kernel() {}
__global__ void touch_kernel(uint32_t* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1u; }
// one 64-byte record per thread, like the step kernel's render record
__global__ void recwrite_kernel(uint4* p, int n, uint32_t v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { p[4 * i] = make_uint4(v, i, 2, 3); p[4 * i + 1] = make_uint4(v, 1, 2, 3); p[4 * i + 2] = make_uint4(v, 1, 2, 3); p[4 * i + 3] = make_uint4(v, 1, 2, 3); } }
// a long dependent chain of binary64 arithmetic per thread, no memory traffic to speak of
__global__ void f64_kernel(double* p, int n, int iters) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; double x = 1.0 + i * 1e-9, y = 0.5; for (int k = 0; k < iters; k++) { x = x * 1.0000001 + y; y = y * 0.999999 + 1e-7; if (x > 1e6) x *= 1e-6; } p[i] = x + y; }
__global__ void i32_kernel(uint32_t* p, int n, int iters) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; uint32_t x = i * 2654435761u; for (int k = 0; k < iters; k++) { x = x * 1664525u + 1013904223u; if (x & 0x100) x ^= x >> 7; } p[i] = x; }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    tbx_engine* e = nullptr;
    const int game = argc > 2 ? atoi(argv[2]) : TBX_GAME_BREAKOUT;
    if (tbx_create(game, n, 0, nullptr, 0, &e)) { printf("create failed: %s\n", tbx_last_error(nullptr)); return 1; }
    tbx_seed(e, -1, 1234); tbx_new_game(e, nullptr);
    hipStream_t s; CK(hipStreamCreate(&s));
    uint64_t t = 0;
    for (int i = 0; i < 600; i++) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s);
    uint32_t* scratch; const size_t words = 13u << 18;     // 13 MB, about what the step kernel touches
    CK(hipMalloc((void**)&scratch, words * 16));
    hipEvent_t ev, a, b; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[] = {"render only", "step ; render", "empty<<<1,64>>> ; render", "empty<<<512,128>>> ; render", "memset 4 B ; render",
                           "event record ; render", "touch 13 MB ; render", "step ; empty<<<1,64>>> ; render", "render ; render ; step (2 frames per step)",
                           "64-B record per thread (4 MB) ; render", "f64 chain 2000 x 65536 threads ; render", "i32 chain 2000 x 65536 threads ; render",
                           "f64 chain 200 ; render", "step with wave-per-env kernel (+ record prep) ; render",
                           "step ; touch 13 MB ; render", "step ; 64-B record per thread (4 MB, elsewhere) ; render", "step ; touch 52 MB ; render"};
    for (int round = 0; round < 3; round++)
        for (int p = 0; p < 17; p++) {
            if (p == 13) tbx_set_option(e, TBX_OPT_STEP_FORM, 2); else if (p == 0 || p == 14) tbx_set_option(e, TBX_OPT_STEP_FORM, 0);
            const int K = 100;
            for (int w = 0; w < 2; w++) {
                if (w == 1) CK(hipEventRecord(a, s));
                for (int i = 0; i < (w ? K : 10); i++) {
                    switch (p) {
                    case 1: tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); break;
                    case 2: empty_kernel<<<1, 64, 0, s>>>(); break;
                    case 3: empty_kernel<<<512, 128, 0, s>>>(); break;
                    case 4: CK(hipMemsetAsync(scratch, 0, 4, s)); break;
                    case 5: CK(hipEventRecord(ev, s)); break;
                    case 6: touch_kernel<<<(unsigned)((words + 255) / 256), 256, 0, s>>>(scratch, words); break;
                    case 7: tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); empty_kernel<<<1, 64, 0, s>>>(); break;
                    case 8: tbx_render_device(e, nullptr, 3, s); if (i & 1) tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); break;
                    case 9: recwrite_kernel<<<(n + 127) / 128, 128, 0, s>>>((uint4*)scratch, n, (uint32_t)i); break;
                    case 10: f64_kernel<<<(n + 127) / 128, 128, 0, s>>>((double*)scratch, n, 2000); break;
                    case 11: i32_kernel<<<(n + 127) / 128, 128, 0, s>>>(scratch, n, 2000); break;
                    case 12: f64_kernel<<<(n + 127) / 128, 128, 0, s>>>((double*)scratch, n, 200); break;
                    case 13: tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); break;
                    case 14: tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); touch_kernel<<<(unsigned)((words + 255) / 256), 256, 0, s>>>(scratch, words); break;
                    case 15: tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); recwrite_kernel<<<(n + 127) / 128, 128, 0, s>>>((uint4*)scratch, n, (uint32_t)i); break;
                    case 16: tbx_step_synthetic(e, 1337, t++, 0, TBX_STEP_AUTO_RESET, s); touch_kernel<<<(unsigned)((4 * words + 255) / 256), 256, 0, s>>>(scratch, 4 * words); break;
                    default: break;
                    }
                    tbx_render_device(e, nullptr, 3, s);
                }
            }
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("round %d  %-44s %.4f ms per iteration\n", round, names[p], ms / K);
            fflush(stdout);
        }
    tbx_destroy(e);
    return 0;
}
