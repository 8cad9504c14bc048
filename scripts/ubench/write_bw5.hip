// Same-run comparison of wave-per-frame store loops (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int FRAME16 = 7200;
// persistent or not; UNROLL of the store loop; IDX64: 64-bit induction
template <bool PERSIST, int UNROLL>
__global__ __launch_bounds__(256) void wpf(uint4* out, int nframes) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), nw = PERSIST ? gridDim.x * 4 : (1 << 30);
    for (int f = w; f < nframes; f += nw) {
        uint4* fr = out + (size_t)f * FRAME16;
        uint4 v = make_uint4(f, 1, 2, 3);
#pragma unroll UNROLL
        for (int i = lane; i < FRAME16; i += 64) fr[i] = v;
    }
}
// stores issued in groups of G with a dummy dependent VALU chain between groups (emulates compute phases)
template <int G, int WORK>
__global__ __launch_bounds__(256) void wpf_work(uint4* out, int nframes) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint4* fr = out + (size_t)f * FRAME16;
    uint32_t a = f, b = lane;
    for (int i0 = 0; i0 < FRAME16; i0 += 64 * G) {
#pragma unroll
        for (int k = 0; k < WORK; k++) { a = a * 5u + b; b ^= a >> 3; }
        uint4 v = make_uint4(a, b, 2, 3);
#pragma unroll
        for (int g = 0; g < G; g++) { int i = i0 + g * 64 + lane; if (i < FRAME16) fr[i] = v; }
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int nf = 65536; const size_t bytes = (size_t)nf * FRAME16 * 16;
    uint4* p; CK(hipMalloc((void**)&p, bytes));
    auto rep = [&](const char* n, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", n, ms, bytes / ms / 1e6); };
    for (int round = 0; round < 2; round++) {
        rep("memset", timeit([&] { hipMemsetAsync(p, 1, bytes, 0); }, 10));
        rep("wpf persist g=2048 unroll1", timeit([&] { wpf<true, 1><<<2048, 256>>>(p, nf); }, 10));
        rep("wpf persist g=2048 unroll4", timeit([&] { wpf<true, 4><<<2048, 256>>>(p, nf); }, 10));
        rep("wpf persist g=2048 unroll8", timeit([&] { wpf<true, 8><<<2048, 256>>>(p, nf); }, 10));
        rep("wpf persist g=1024 unroll4", timeit([&] { wpf<true, 4><<<1024, 256>>>(p, nf); }, 10));
        rep("wpf persist g=512 unroll4", timeit([&] { wpf<true, 4><<<512, 256>>>(p, nf); }, 10));
        rep("wpf persist g=256 unroll4", timeit([&] { wpf<true, 4><<<256, 256>>>(p, nf); }, 10));
        rep("wpf oneshot g=16384 unroll1", timeit([&] { wpf<false, 1><<<16384, 256>>>(p, nf); }, 10));
        rep("wpf oneshot g=16384 unroll4", timeit([&] { wpf<false, 4><<<16384, 256>>>(p, nf); }, 10));
        rep("wpf_work G=6 work=16 (8-row units)", timeit([&] { wpf_work<6, 16><<<16384, 256>>>(p, nf); }, 10));
        rep("wpf_work G=6 work=64", timeit([&] { wpf_work<6, 64><<<16384, 256>>>(p, nf); }, 10));
        rep("wpf_work G=12 work=128", timeit([&] { wpf_work<12, 128><<<16384, 256>>>(p, nf); }, 10));
        rep("wpf_work G=3 work=32", timeit([&] { wpf_work<3, 32><<<16384, 256>>>(p, nf); }, 10));
    }
    hipFree(p); return 0;
}
