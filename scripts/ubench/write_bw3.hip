// Write-bandwidth experiments, round 3: persistent grids / limited streams in flight (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
struct alignas(4) U3 { uint32_t a, b, c; };

// persistent, wave-per-frame, 1 KiB per store
__global__ __launch_bounds__(256) void pw16(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63; int w = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int f = w; f < nframes; f += nw) {
        uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
        uint4 v = make_uint4(f, 1, 2, 3);
        for (int i = lane; i < 7200; i += 64) fr[i] = v;
    }
}
// persistent, wave-per-frame, rows 60 lanes x 12 B, rotated
__global__ __launch_bounds__(256) void pr12(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63; int w = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int f = w; f < nframes; f += nw) {
        uint8_t* fr = out + (size_t)f * 115200;
        U3 v{(uint32_t)f, 2u, 3u};
        int y0 = ((f * 2654435761u) >> 8) % 160;
        if (lane < 60) for (int k = 0; k < 160; k++) { int y = y0 + k; if (y >= 160) y -= 160; *reinterpret_cast<U3*>(fr + y * 720 + lane * 12) = v; }
    }
}
// persistent, block-per-frame (4 waves interleave 1 KiB chunks of one frame)
__global__ __launch_bounds__(256) void pb16(uint8_t* out, int nframes) {
    for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
        uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
        uint4 v = make_uint4(f, 1, 2, 3);
        for (int i = threadIdx.x; i < 7200; i += 256) fr[i] = v;
    }
}
// persistent, block-per-frame with 1024 threads
__global__ __launch_bounds__(1024) void pb16k(uint8_t* out, int nframes) {
    for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
        uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
        uint4 v = make_uint4(f, 1, 2, 3);
        for (int i = threadIdx.x; i < 7200; i += 1024) fr[i] = v;
    }
}
// non-persistent wave-per-frame but consecutive frames in the SAME wave slot order: block b handles frames 4b..4b+3 (baseline)
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
    for (int g : {256, 512, 1024, 2048, 4096}) {
        rep("pw16 wave/frame", g, timeit([&] { pw16<<<g, 256>>>(p, nf); }, 10));
        rep("pr12 wave/frame rows rot", g, timeit([&] { pr12<<<g, 256>>>(p, nf); }, 10));
        rep("pb16 block/frame", g, timeit([&] { pb16<<<g, 256>>>(p, nf); }, 10));
    }
    for (int g : {256, 512}) rep("pb16k block1024/frame", g, timeit([&] { pb16k<<<g, 1024>>>(p, nf); }, 10));
    hipFree(p); return 0;
}
