// Write-bandwidth experiments, round 2 (diagnostic, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int UNROLL>
__global__ void fill16u(uint4* p, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(i, 1, 2, 3);
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) p[i + u * stride] = v;
    }
    for (; i < n16; i += stride) p[i] = v;
}
// wave-per-frame, 1 KiB per store, optional rotated start and padded frame stride
template <bool ROT>
__global__ __launch_bounds__(256) void wave16(uint8_t* out, int nframes, size_t fstride) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * fstride);
    uint4 v = make_uint4(f, 1, 2, 3);
    const int total = 7200;          // 16-B chunks per frame
    int start = ROT ? ((f * 2654435761u) >> 8) % 112 * 64 : 0;   // rotate by whole KiB chunks
    for (int k = 0; k < 113; k++) {
        int i = start + k * 64 + lane;
        if (i >= 7232) i -= 7232;    // wrap over 113 chunks
        if (i < total) fr[i] = v;
    }
}
// block-per-frame: 4 waves split one frame into interleaved 1 KiB chunks
__global__ __launch_bounds__(256) void block16(uint8_t* out, int nframes) {
    int f = blockIdx.x;
    uint4* fr = reinterpret_cast<uint4*>(out + (size_t)f * 115200);
    uint4 v = make_uint4(f, 1, 2, 3);
    for (int i = threadIdx.x; i < 7200; i += 256) fr[i] = v;
}
// rows of 720 B: 60 lanes x 12 B, with rotated row start
struct alignas(4) U3 { uint32_t a, b, c; };
template <bool ROT>
__global__ __launch_bounds__(256) void rows12(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint8_t* fr = out + (size_t)f * 115200;
    U3 v{(uint32_t)f, 2u, 3u};
    int y0 = ROT ? ((f * 2654435761u) >> 8) % 160 : 0;
    if (lane < 60) for (int k = 0; k < 160; k++) { int y = y0 + k; if (y >= 160) y -= 160; *reinterpret_cast<U3*>(fr + y * 720 + lane * 12) = v; }
}
// rows of 720 B written as 45 lanes x 16 B
template <bool ROT>
__global__ __launch_bounds__(256) void rows16(uint8_t* out, int nframes) {
    int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint8_t* fr = out + (size_t)f * 115200;
    uint4 v = make_uint4(f, 1, 2, 3);
    int y0 = ROT ? ((f * 2654435761u) >> 8) % 160 : 0;
    if (lane < 45) for (int k = 0; k < 160; k++) { int y = y0 + k; if (y >= 160) y -= 160; *reinterpret_cast<uint4*>(fr + y * 720 + lane * 16) = v; }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int nf = 65536; const size_t bytes = (size_t)nf * 115200;
    uint8_t* p; CK(hipMalloc((void**)&p, (size_t)nf * (115200 + 4096)));
    auto rep = [&](const char* n, float ms) { printf("%-34s %8.3f ms  %7.1f GB/s\n", n, ms, bytes / ms / 1e6); };
    rep("memset", timeit([&] { hipMemsetAsync(p, 1, bytes, 0); }, 10));
    for (int bs : {256, 1024}) for (int g : {1024, 2048, 4096, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "fill16u1 bs=%d grid=%d", bs, g); rep(nm, timeit([&] { fill16u<1><<<g, bs>>>((uint4*)p, bytes / 16); }, 10));
        snprintf(nm, 64, "fill16u4 bs=%d grid=%d", bs, g); rep(nm, timeit([&] { fill16u<4><<<g, bs>>>((uint4*)p, bytes / 16); }, 10));
    }
    rep("wave16", timeit([&] { wave16<false><<<nf / 4, 256>>>(p, nf, 115200); }, 10));
    rep("wave16 rot", timeit([&] { wave16<true><<<nf / 4, 256>>>(p, nf, 115200); }, 10));
    rep("wave16 stride+256", timeit([&] { wave16<false><<<nf / 4, 256>>>(p, nf, 115200 + 256); }, 10));
    rep("wave16 stride+1024", timeit([&] { wave16<false><<<nf / 4, 256>>>(p, nf, 115200 + 1024); }, 10));
    rep("wave16 rot stride+4096", timeit([&] { wave16<true><<<nf / 4, 256>>>(p, nf, 115200 + 4096); }, 10));
    rep("block16 (block per frame)", timeit([&] { block16<<<nf, 256>>>(p, nf); }, 10));
    rep("rows12", timeit([&] { rows12<false><<<nf / 4, 256>>>(p, nf); }, 10));
    rep("rows12 rot", timeit([&] { rows12<true><<<nf / 4, 256>>>(p, nf); }, 10));
    rep("rows16 (45 lanes)", timeit([&] { rows16<false><<<nf / 4, 256>>>(p, nf); }, 10));
    rep("rows16 rot", timeit([&] { rows16<true><<<nf / 4, 256>>>(p, nf); }, 10));
    hipFree(p); return 0;
}
