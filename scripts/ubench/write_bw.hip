// Write-bandwidth ceilings for the rasteriser's store patterns on MI355X (diagnostic, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// A: flat fill, 16 B per lane, grid-stride, fully coalesced
__global__ void fill16(uint4* p, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(i, 1, 2, 3);
    for (; i < n16; i += stride) p[i] = v;
}
// B: wave-per-frame, rows of 720 B written as 60 lanes x 12 B (the r01a rasteriser pattern)
struct alignas(4) U3 { uint32_t a, b, c; };
__global__ __launch_bounds__(256) void rows12(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint8_t* fr = out + (size_t)f * 115200;
    U3 v{(uint32_t)f, 2u, 3u};
    if (lane < 60) for (int y = 0; y < 160; y++) *reinterpret_cast<U3*>(fr + y * 720 + lane * 12) = v;
}
// C: wave-per-frame, 1 KiB per store instruction (64 lanes x 16 B), 112.5 stores per frame
__global__ __launch_bounds__(256) void wave16(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
    uint4 v = make_uint4(f, 1, 2, 3);
    for (int i = lane; i < 7200; i += 64) fr[i] = v;
}
// D: block-per-4-frames but each wave writes 4 KiB chunks (4 x dwordx4 per lane back to back)
__global__ __launch_bounds__(256) void wave16x4(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
    uint4 v = make_uint4(f, 1, 2, 3);
    int i = lane;
    for (; i + 192 < 7200; i += 256) { fr[i] = v; fr[i + 64] = v; fr[i + 128] = v; fr[i + 192] = v; }
    for (; i < 7200; i += 64) fr[i] = v;
}
// E: nontemporal variant of C
__global__ __launch_bounds__(256) void wave16nt(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
    for (int i = lane; i < 7200; i += 64) {
        __builtin_nontemporal_store((uint32_t)f, &fr[i].x); __builtin_nontemporal_store(1u, &fr[i].y);
        __builtin_nontemporal_store(2u, &fr[i].z); __builtin_nontemporal_store(3u, &fr[i].w);
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
    auto rep = [&](const char* n, float ms) { printf("%-28s %8.3f ms  %7.1f GB/s\n", n, ms, bytes / ms / 1e6); };
    rep("memset", timeit([&] { hipMemsetAsync(p, 1, bytes, 0); }, 10));
    for (int g : {2048, 4096, 8192, 16384})
        { char nm[64]; snprintf(nm, 64, "fill16 grid=%d", g); rep(nm, timeit([&] { fill16<<<g, 256>>>((uint4*)p, bytes / 16); }, 10)); }
    rep("rows12 (60 lanes x 12B)", timeit([&] { rows12<<<nf / 4, 256>>>(p, nf); }, 10));
    rep("wave16 (1KiB/instr)", timeit([&] { wave16<<<nf / 4, 256>>>(p, nf); }, 10));
    rep("wave16x4", timeit([&] { wave16x4<<<nf / 4, 256>>>(p, nf); }, 10));
    rep("wave16 nontemporal", timeit([&] { wave16nt<<<nf / 4, 256>>>(p, nf); }, 10));
    hipFree(p); return 0;
}
