// Address-ordered work items + LDS staging (diagnostic): isolates what costs bandwidth in the real rasteriser.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int W = 240, H = 160, C = 3, ROWB = W * C, FRAME = ROWB * H;
struct alignas(64) Rec { uint32_t w[16]; };

// STAGE: 0 = registers only (no LDS), 1 = LDS staging; REC: load a 64-byte record per item; WORK: dummy VALU per row
template <int R, int STAGE, bool REC, int WORK>
__global__ __launch_bounds__(256) void ordered(const Rec* __restrict__ recs, uint8_t* __restrict__ out, int nframes) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * R * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* lds = lds_all + wave * R * ROWB;
    const int nw = gridDim.x * 4;
    constexpr int NU = H / R, CHUNKS = R * ROWB / 16;
    const int nitems = nframes * NU;
    for (int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave); q < nitems; q += nw) {
        const int f = q / NU, u = q - f * NU;
        uint32_t seed = f;
        if (REC) { const Rec rc = recs[f]; seed = rc.w[0] + rc.w[5] + rc.w[9] + rc.w[13]; }
        uint4* dst = reinterpret_cast<uint4*>(out + (size_t)f * FRAME + (size_t)u * R * ROWB);
        if (STAGE == 1) {
#pragma unroll 1
            for (int r = 0; r < R; r++) {
                uint32_t a = seed + r + u, b = lane, c = 7;
#pragma unroll
                for (int i = 0; i < WORK; i++) { a = a * 1664525u + b; b ^= a >> 3; c += b; }
                if (lane < 60) { uint32_t* p = reinterpret_cast<uint32_t*>(lds + r * ROWB + lane * 12); p[0] = a; p[1] = b; p[2] = c; }
            }
            const uint4* src = reinterpret_cast<const uint4*>(lds);
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int i = lane; i < CHUNKS; i += 64) dst[i] = src[i];
            __builtin_amdgcn_wave_barrier();
        } else {
            uint32_t a = seed + u, b = lane, c = 7;
#pragma unroll
            for (int i = 0; i < WORK * R / 4; i++) { a = a * 1664525u + b; b ^= a >> 3; c += b; }
            uint4 v = make_uint4(a, b, c, 1);
#pragma unroll 4
            for (int i = lane; i < CHUNKS; i += 64) dst[i] = v;
        }
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
    Rec* recs; CK(hipMalloc((void**)&recs, sizeof(Rec) * nf)); CK(hipMemset(recs, 1, sizeof(Rec) * nf));
    auto rep = [&](const char* n, float ms) { printf("%-56s %8.3f ms  %7.1f GB/s\n", n, ms, bytes / ms / 1e6); };
    rep("memset", timeit([&] { hipMemsetAsync(p, 1, bytes, 0); }, 10));
#define RUN(R, ST, REC, WORK, G) rep("ordered R=" #R " stage=" #ST " rec=" #REC " work=" #WORK " grid=" #G, timeit([&] { ordered<R, ST, REC, WORK><<<G, 256>>>(recs, p, nf); }, 10))
    RUN(8, 0, false, 0, 2048); RUN(8, 0, false, 8, 2048); RUN(8, 0, true, 8, 2048);
    RUN(8, 1, false, 8, 1536); RUN(8, 1, true, 8, 1536); RUN(8, 1, true, 8, 1024); RUN(8, 1, true, 8, 768); RUN(8, 1, true, 8, 512);
    RUN(8, 1, true, 2, 1536); RUN(8, 1, true, 16, 1536);
    RUN(4, 1, true, 8, 2048); RUN(16, 1, true, 8, 768); RUN(16, 1, true, 8, 512); RUN(32, 1, true, 8, 256); RUN(32, 1, true, 8, 512);
    hipFree(p); return 0;
}
