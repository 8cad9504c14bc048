// Write-bandwidth experiments, round 4: globally address-ordered work units (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__global__ void fill16(uint4* p, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(i, 1, 2, 3);
    for (; i < n16; i += stride) p[i] = v;
}
// persistent: wave w writes unit u = w + NW*i; unit = UNIT16 16-byte chunks, contiguous
template <int UNIT16>
__global__ __launch_bounds__(256) void units(uint4* out, size_t n16) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    const size_t nunits = n16 / UNIT16;
    for (size_t u = w; u < nunits; u += nw) {
        uint4* base = out + u * UNIT16;
        uint4 v = make_uint4((uint32_t)u, 1, 2, 3);
#pragma unroll 4
        for (int i = lane; i < UNIT16; i += 64) base[i] = v;
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int nf = 65536; const size_t bytes = (size_t)nf * 115200;
    uint8_t* p; CK(hipMalloc((void**)&p, bytes));
    auto rep = [&](const char* n, int g, float ms) { printf("%-26s grid=%5d %8.3f ms  %7.1f GB/s\n", n, g, ms, bytes / ms / 1e6); };
    rep("memset", 0, timeit([&] { hipMemsetAsync(p, 1, bytes, 0); }, 10));
    for (int g : {128, 256, 512, 768, 1024, 2048}) rep("fill16 bs256", g, timeit([&] { fill16<<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
    for (int g : {256, 512, 1024, 2048}) {
        rep("units 720B (1 row)", g, timeit([&] { units<45><<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
        rep("units 1KiB", g, timeit([&] { units<64><<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
        rep("units 2880B (4 rows)", g, timeit([&] { units<180><<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
        rep("units 5760B (8 rows)", g, timeit([&] { units<360><<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
        rep("units 23040B (32 rows)", g, timeit([&] { units<1440><<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
        rep("units 115200B (frame)", g, timeit([&] { units<7200><<<g, 256>>>((uint4*)p, bytes / 16); }, 10));
    }
    hipFree(p); return 0;
}
