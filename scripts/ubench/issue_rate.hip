// issue_rate.hip -- how many scalar-ALU and vector-ALU instructions a gfx950 compute unit issues per cycle, alone and side
// by side, at 1..8 waves per SIMD.  (Diagnostic; not part of the product.  The rasterisers execute about as many SALU as VALU
// instructions -- profiles/r04_pmc_sq_counters.txt -- and DESIGN.md section 6 prices their issue time with these rates.)
//
// Each wave runs REPS iterations of a block of independent instructions: S scalar adds on 8 SGPR chains and / or V vector adds
// on 8 VGPR chains, interleaved.  The launch is one block of (waves per SIMD x 4) waves per CU; rate = instructions / (cycles
// of the slowest wave, s_memtime at start and end).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 2000;

#define S8 "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
#define V8 "v_add_u32 %8, %8, 1\n v_add_u32 %9, %9, 1\n v_add_u32 %10, %10, 1\n v_add_u32 %11, %11, 1\n v_add_u32 %12, %12, 1\n v_add_u32 %13, %13, 1\n v_add_u32 %14, %14, 1\n v_add_u32 %15, %15, 1\n"
#define SV8 "s_add_u32 %0, %0, 1\n v_add_u32 %8, %8, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %9, %9, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %10, %10, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %11, %11, 1\n" \
            "s_add_u32 %4, %4, 1\n v_add_u32 %12, %12, 1\n s_add_u32 %5, %5, 1\n v_add_u32 %13, %13, 1\n s_add_u32 %6, %6, 1\n v_add_u32 %14, %14, 1\n s_add_u32 %7, %7, 1\n v_add_u32 %15, %15, 1\n"

template <int MODE>   // 0: 32 SALU per iteration, 1: 32 VALU, 2: 16 SALU + 16 VALU interleaved, 3: 32 SALU + 32 VALU in blocks of 8
__global__ void issue_kernel(unsigned long long* cycles, unsigned* sink)
{
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r++) {
        if (MODE == 0)
            asm volatile(S8 S8 S8 S8 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7),
                         "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "scc");
        else if (MODE == 1)
            asm volatile(V8 V8 V8 V8 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7),
                         "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "scc");
        else if (MODE == 2)
            asm volatile(SV8 SV8 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7),
                         "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "scc");
        else
            asm volatile(S8 V8 S8 V8 S8 V8 S8 V8 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7),
                         "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 0xFFFFFFFFu) *sink = 1;
}

template <int MODE>
static void run(const char* what, int salu, int valu, int waves_per_simd, unsigned long long* d_cycles, unsigned* d_sink)
{
    const int blocks = 256, wpb = 4 * waves_per_simd;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(issue_kernel<MODE>, dim3(blocks), dim3(64 * wpb), 0, 0, d_cycles, d_sink);      // warm-up
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(issue_kernel<MODE>, dim3(blocks), dim3(64 * wpb), 0, 0, d_cycles, d_sink);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(blocks * wpb);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof h[0], hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    // s_memtime counts at a fixed 100 MHz on this part: convert through the event time instead -- per-CU rates in
    // instructions per nanosecond, and per cycle at the clock the launch sustained if the caller knows it
    const double per_cu_s = (double)salu * REPS * wpb, per_cu_v = (double)valu * REPS * wpb;
    printf("%-34s waves/SIMD %d  %8.3f ms  SALU %6.3f  VALU %6.3f  per ns per CU   (memtime ticks median %llu, max %llu)\n", what, waves_per_simd, ms,
           per_cu_s / (ms * 1e6), per_cu_v / (ms * 1e6), h[h.size() / 2], h.back());
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
}

int main()
{
    unsigned long long* d_cycles; unsigned* d_sink;
    CHECK(hipMalloc(&d_cycles, 256 * 32 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, 4));
    for (int w : {1, 2, 4, 8}) {
        run<0>("32 SALU", 32, 0, w, d_cycles, d_sink);
        run<1>("32 VALU", 0, 32, w, d_cycles, d_sink);
        run<2>("16 SALU + 16 VALU interleaved", 16, 16, w, d_cycles, d_sink);
        run<3>("32 SALU + 32 VALU, blocks of 8", 32, 32, w, d_cycles, d_sink);
    }
    return 0;
}
