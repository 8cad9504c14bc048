// The shader clock WHILE something else runs (round 6, r06_experiments item 5): a one-wave kernel on a stream of its own stamps
// s_memtime (shader cycles) against s_memrealtime (100 MHz) every `interval_us` until the host raises a flag in mapped memory, so
// the clock of each interval = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, "in-kernel clock").  The box's hwmon
// files do not move under load here, this does.  Built as a shared object and driven from scripts/box_probe.py.  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace {
constexpr int MAX_SAMPLES = 4096;
hipStream_t g_stream = nullptr;
uint64_t* g_samples = nullptr;            // device: [0] = count, then (memtime, realtime) pairs
volatile uint32_t* g_stop_host = nullptr; // mapped host memory
uint32_t* g_stop_dev = nullptr;
}

__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t* out, const uint32_t* stop, uint64_t interval_ticks, uint64_t max_ticks)
{
    if (threadIdx.x != 0) return;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t next = r0;
    int n = 0;
    for (;;) {
        const uint64_t r = __builtin_amdgcn_s_memrealtime();
        if (r < next) { __builtin_amdgcn_s_sleep(64); continue; }
        const uint64_t m = __builtin_amdgcn_s_memtime();
        const uint64_t r2 = __builtin_amdgcn_s_memrealtime();
        out[1 + 2 * n] = m; out[2 + 2 * n] = r2;
        n += 1;
        next = r + interval_ticks;
        const uint32_t s = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (s || n >= MAX_SAMPLES || r - r0 > max_ticks) break;
    }
    out[0] = (uint64_t)n;
}

extern "C" int clkp_start(int interval_us, int max_ms)
{
    if (!g_stream) {
        // the LOW priority level: the runtime multiplexes a process's streams onto four hardware queues per priority level, and a
        // fifth ordinary stream would share a queue with one of the measured loop's (this kernel then holds that stream up until it ends)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return 1;
        if (hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, lo) != hipSuccess) return 1;
        if (hipMalloc(&g_samples, sizeof(uint64_t) * (1 + 2 * MAX_SAMPLES)) != hipSuccess) return 2;
        if (hipHostMalloc((void**)&g_stop_host, 64, hipHostMallocMapped) != hipSuccess) return 3;
        if (hipHostGetDevicePointer((void**)&g_stop_dev, (void*)g_stop_host, 0) != hipSuccess) return 4;
    }
    *g_stop_host = 0u;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, g_stream, g_samples, g_stop_dev, (uint64_t)interval_us * 100ull, (uint64_t)max_ms * 100000ull);
    return hipGetLastError() == hipSuccess ? 0 : 5;
}

// raises the flag, waits for the kernel, writes the clock of each interval in MHz; returns the number of intervals (< 0: error)
extern "C" int clkp_stop(double* mhz, int max)
{
    if (!g_stream) return -1;
    *g_stop_host = 1u;
    if (hipStreamSynchronize(g_stream) != hipSuccess) return -2;
    static uint64_t host[1 + 2 * MAX_SAMPLES];
    if (hipMemcpy(host, g_samples, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    const int n = (int)host[0];
    int k = 0;
    for (int i = 1; i < n && k < max; i++) {
        const double dm = (double)(host[1 + 2 * i] - host[1 + 2 * (i - 1)]), dr = (double)(host[2 + 2 * i] - host[2 + 2 * (i - 1)]);
        mhz[k++] = dr > 0 && dm > 0 && dm < dr * 100.0 ? dm / dr * 100.0 : 0.0;      // (a stamp pair torn by a clock change reads as 0)
    }
    return k;
}
