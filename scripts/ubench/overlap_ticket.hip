// Overlapping consecutive "fused" launches of ONE engine on two streams (round 6): what orders launch N+1
// behind the STEP HALF of launch N (a few blocks at the front of the grid) without waiting for its rasteriser half?
//   serial   one stream, launch after launch (today's tbx_render_step_synthetic loop)
//   event    two streams, launch N+1 waits for the completion EVENT of launch N (no overlap: the control)
//   waitk    two streams; a one-wave kernel in front of launch N+1 spins (bounded) until the step blocks of launch N have
//            bumped a device counter -- the launch itself never spins, so nothing can fill the chip with waiting blocks
//   waitv    two streams; hipStreamWaitValue64 on the same counter (command-processor wait; only if the device reports
//            hipDeviceAttributeCanUseStreamWaitValue)
// The launch is a stand-in for brk_render_step_kernel_w5: `step_blocks` blocks spin ~10 us, fence, bump the counter; every
// other block's four waves store 12 KiB each with 16-byte stores (five waves per SIMD like the rasteriser).  (diagnostic)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1;}}while(0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void fused_like(uint4* __restrict__ out, unsigned long long* arrive,
                                                                                                 int step_blocks, int spin_us, uint32_t tag)
{
    if ((int)blockIdx.x < step_blocks) {
        const uint64_t a = wall_clock64();
        while (wall_clock64() - a < (uint64_t)spin_us * 100) {}
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence(); __hip_atomic_fetch_add(arrive, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        return;
    }
    const int wave = ((int)blockIdx.x - step_blocks) * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint4* dst = out + (size_t)wave * 768 + lane;          // 12 KiB per wave
    const uint4 v = make_uint4(tag, wave, lane, 0);
#pragma unroll
    for (int i = 0; i < 12; i++) dst[i * 64] = v;
}

__global__ void wait_kernel(const unsigned long long* arrive, unsigned long long want, uint32_t* timeout_flag)
{
    if (threadIdx.x != 0) return;
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200000000ull) { atomicOr(timeout_flag, 1u); return; }   // 2 s: give up loudly, never hang
    }
}

int main(int argc, char** argv)
{
    const int envs = argc > 1 ? atoi(argv[1]) : 8192, iters = argc > 2 ? atoi(argv[2]) : 400;
    const int step_blocks = (envs + 255) / 256, waves = envs * 10, raster_blocks = (waves + 3) / 4;
    const size_t bytes = (size_t)raster_blocks * 4 * 12288;
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("envs %d: %d step blocks + %d rasteriser blocks, %.1f MB per launch; hipDeviceAttributeCanUseStreamWaitValue = %d\n", envs, step_blocks, raster_blocks, bytes / 1e6, can);
    uint4* buf[2]; CK(hipMalloc((void**)&buf[0], bytes)); CK(hipMalloc((void**)&buf[1], bytes));
    unsigned long long* arrive = nullptr;
    if (hipExtMallocWithFlags((void**)&arrive, 8, hipMallocSignalMemory) != hipSuccess) { printf("no signal memory; plain hipMalloc\n"); CK(hipMalloc((void**)&arrive, 8)); can = 0; }
    uint32_t* tflag; CK(hipMalloc((void**)&tflag, 4)); CK(hipMemset(tflag, 0, 4));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t lane[2]; for (int k = 0; k < 2; k++) CK(hipStreamCreateWithPriority(&lane[k], hipStreamNonBlocking, hi));
    hipEvent_t done[2]; for (int k = 0; k < 2; k++) CK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    const char* names[] = {"serial", "event", "waitk", "waitv"};
    for (int round = 0; round < 2; round++)
    for (int mode = 0; mode < 4; mode++) {
        if (mode == 3 && !can) { if (round == 0) printf("waitv   not supported on this device\n"); continue; }
        CK(hipMemset(arrive, 0, 8)); CK(hipDeviceSynchronize());
        unsigned long long seq = 0;
        auto run = [&](int n) -> int {
            for (int i = 0; i < n; i++) {
                const int p = (mode == 0) ? 0 : (int)(seq & 1);
                hipStream_t s = lane[p];
                if (seq > 0) {
                    if (mode == 1) CK(hipStreamWaitEvent(s, done[p ^ 1], 0));
                    if (mode == 2) hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, s, arrive, seq * (unsigned long long)step_blocks, tflag);
                    if (mode == 3) CK(hipStreamWaitValue64(s, arrive, seq * (unsigned long long)step_blocks, hipStreamWaitValueGte));
                }
                if (mode == 1) hipExtLaunchKernelGGL(fused_like, dim3(step_blocks + raster_blocks), dim3(256), 0, s, nullptr, done[p], 0, buf[seq & 1], arrive, step_blocks, 10, (uint32_t)seq);
                else hipLaunchKernelGGL(fused_like, dim3(step_blocks + raster_blocks), dim3(256), 0, s, buf[seq & 1], arrive, step_blocks, 10, (uint32_t)seq);
                seq++;
            }
            return 0;
        };
        if (run(40)) return 1;
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        if (run(iters)) return 1;
        CK(hipDeviceSynchronize());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
        uint32_t tf = 0; CK(hipMemcpy(&tf, tflag, 4, hipMemcpyDeviceToHost));
        unsigned long long got = 0; CK(hipMemcpy(&got, arrive, 8, hipMemcpyDeviceToHost));
        printf("%-7s %.4f ms per launch = %.3f TB/s (%.3f of 8); counter %llu / %llu%s\n", names[mode], ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0, got,
               seq * (unsigned long long)step_blocks, tf ? "  WAIT KERNEL TIMED OUT" : "");
    }
    return 0;
}
