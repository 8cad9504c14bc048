// LDS-staged scanline writer experiments (diagnostic, not product): which unit size / grid shape /
// flush shape gets a wave-per-frame rasteriser closest to the memset rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int W = 240, H = 160, C = 3, ROWB = W * C, FRAME = ROWB * H;

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// R rows per unit, PERSIST: waves stride over frames, ROT: 0 none, 1 linear (f*13), 2 hashed; WORK: dummy VALU per row
template <int R, bool PERSIST, int ROT, int WORK, int FLUSH>
__global__ __launch_bounds__(256) void staged(uint8_t* out, int nframes) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * R * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* lds = lds_all + wave * R * ROWB;
    const int nw = PERSIST ? gridDim.x * 4 : 1 << 30;
    constexpr int NU = H / R, CHUNKS = R * ROWB / 16;
    for (int f = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave); f < nframes; f += nw) {
        uint8_t* frame = out + (size_t)f * FRAME;
        int u0 = ROT == 0 ? 0 : ROT == 1 ? (int)(((uint32_t)f * 13u) % NU) : (int)(hash32(f) % NU);
        for (int k = 0; k < NU; k++) {
            int u = u0 + k; if (u >= NU) u -= NU;
#pragma unroll 1
            for (int r = 0; r < R; r++) {
                uint32_t a = f + r + u, b = lane, c = 7;
#pragma unroll
                for (int i = 0; i < WORK; i++) { a = a * 1664525u + b; b ^= a >> 3; c += b; }
                if (lane < 60) { uint32_t* p = reinterpret_cast<uint32_t*>(lds + r * ROWB + lane * 12); p[0] = a; p[1] = b; p[2] = c; }
            }
            const uint4* src = reinterpret_cast<const uint4*>(lds);
            uint4* dst = reinterpret_cast<uint4*>(frame + (size_t)u * R * ROWB);
            __builtin_amdgcn_wave_barrier();
            if (FLUSH == 0) {
#pragma unroll 4
                for (int i = lane; i < CHUNKS; i += 64) dst[i] = src[i];
            } else {   // read everything first, then store
                constexpr int NI = (CHUNKS + 63) / 64;
                uint4 v[NI];
#pragma unroll
                for (int j = 0; j < NI; j++) { int i = lane + 64 * j; if (i < CHUNKS) v[j] = src[i]; }
#pragma unroll
                for (int j = 0; j < NI; j++) { int i = lane + 64 * j; if (i < CHUNKS) dst[i] = v[j]; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (!PERSIST) break;
    }
}
// direct 12-byte row stores, no LDS
struct alignas(4) U3 { uint32_t a, b, c; };
template <int ROT, int WORK>
__global__ __launch_bounds__(256) void direct12(uint8_t* out, int nframes) {
    const int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    uint8_t* frame = out + (size_t)f * FRAME;
    int y0 = ROT == 0 ? 0 : (int)(hash32(f) % H);
    for (int k = 0; k < H; k++) {
        int y = y0 + k; if (y >= H) y -= H;
        uint32_t a = f + y, b = lane, c = 7;
#pragma unroll
        for (int i = 0; i < WORK; i++) { a = a * 1664525u + b; b ^= a >> 3; c += b; }
        if (lane < 60) *reinterpret_cast<U3*>(frame + y * ROWB + lane * 12) = U3{a, b, c};
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int nf = 65536; const size_t bytes = (size_t)nf * FRAME;
    uint8_t* p; CK(hipMalloc((void**)&p, bytes));
    auto rep = [&](const char* n, float ms) { printf("%-52s %8.3f ms  %7.1f GB/s\n", n, ms, bytes / ms / 1e6); };
    rep("memset", timeit([&] { hipMemsetAsync(p, 1, bytes, 0); }, 10));
    rep("direct12 norot work8", timeit([&] { direct12<0, 8><<<nf / 4, 256>>>(p, nf); }, 10));
    rep("direct12 rot work8", timeit([&] { direct12<1, 8><<<nf / 4, 256>>>(p, nf); }, 10));
    rep("direct12 rot work32", timeit([&] { direct12<1, 32><<<nf / 4, 256>>>(p, nf); }, 10));
#define RUN(R, P, ROT, WORK, FL, G) rep("staged R=" #R " persist=" #P " rot=" #ROT " work=" #WORK " flush=" #FL " grid=" #G, timeit([&] { staged<R, P, ROT, WORK, FL><<<G, 256>>>(p, nf); }, 10))
    RUN(4, false, 0, 8, 0, 16384); RUN(4, false, 1, 8, 0, 16384); RUN(4, false, 2, 8, 0, 16384);
    RUN(8, false, 0, 8, 0, 16384); RUN(8, false, 1, 8, 0, 16384); RUN(8, false, 2, 8, 0, 16384);
    RUN(16, false, 2, 8, 0, 16384); RUN(32, false, 2, 8, 0, 16384);
    RUN(4, true, 2, 8, 0, 2048); RUN(8, true, 2, 8, 0, 2048); RUN(8, true, 2, 8, 0, 1536); RUN(8, true, 2, 8, 0, 1024);
    RUN(16, true, 2, 8, 0, 1024); RUN(16, true, 2, 8, 0, 768); RUN(32, true, 2, 8, 0, 512);
    RUN(8, false, 2, 8, 1, 16384); RUN(8, true, 2, 8, 1, 1536); RUN(16, true, 2, 8, 1, 768);
    RUN(8, false, 2, 32, 0, 16384); RUN(8, true, 2, 32, 0, 1536);
    hipFree(p); return 0;
}
