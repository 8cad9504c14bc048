// Pure-store emulation of the rasterisers' launch geometry: `split` waves share a frame of FRAME bytes cut into units of UNIT
// bytes; wave `part` stores units part, part + split, ... as 16-byte-per-lane stores (1 KiB per instruction).  Does the frame /
// unit geometry alone explain why SpaceInvaders' frames (201 600 B) stream slower than Breakout's (115 200 B)?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
// occupancy is throttled from the host with dynamic LDS (bytes per block); nap: s_sleep units between two store instructions
__global__ __launch_bounds__(256) void units(uint4* out, int nframes, int frame16, int unit16, int nunits, int split, int rot, int nap)
{
    extern __shared__ uint8_t lds_dyn[];
    if (nframes < 0) lds_dyn[threadIdx.x] = 1;          // keep the allocation alive
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= nframes) return;
    uint4* fr = out + (size_t)rel * frame16;
    const uint4 v = make_uint4(rel, part, 2, 3);
    if (nap >= 1000) {                                   // stagger: waves start their store phase at different times
        const int k = (int)(((unsigned)wid * 2654435761u) >> 29);          // 0..7
        for (int j = 0; j < k * (nap - 1000); j++) __builtin_amdgcn_s_sleep(32);     // 32 * 64 clocks ~ 0.85 us per round
        nap = 0;
    }
    for (int q = part; q < nunits; q += split) {
        const int u = rot ? (int)(((unsigned)rel * 7u + (unsigned)q) % (unsigned)nunits) : q;
        uint4* dst = fr + (size_t)u * unit16;
        for (int i = lane; i < unit16; i += 64) {
            dst[i] = v;
            if (nap == 1) __builtin_amdgcn_s_sleep(1);
            else if (nap == 4) __builtin_amdgcn_s_sleep(4);
            else if (nap == 16) __builtin_amdgcn_s_sleep(16);
            else if (nap == 64) __builtin_amdgcn_s_sleep(64);
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
    const int nf = 65536;
    const size_t cap = (size_t)nf * 201600;
    uint4* p; CK(hipMalloc((void**)&p, cap));
    struct Cfg { const char* name; int frame, unit, split, rot, lds, nap; };
    const Cfg cfgs[] = {
        {"breakout 5760 split10", 115200, 5760, 10, 0, 0, 0},
        {"breakout 5760 split10 stagger1", 115200, 5760, 10, 0, 0, 1001}, {"breakout 5760 split10 stagger2", 115200, 5760, 10, 0, 0, 1002},
        {"breakout 5760 split10 stagger4", 115200, 5760, 10, 0, 0, 1004}, {"breakout 5760 split10 stagger8", 115200, 5760, 10, 0, 0, 1008},
        {"si 5760 split7", 201600, 5760, 7, 0, 0, 0},
        {"si 5760 split7 stagger1", 201600, 5760, 7, 0, 0, 1001}, {"si 5760 split7 stagger2", 201600, 5760, 7, 0, 0, 1002},
        {"si 5760 split7 stagger4", 201600, 5760, 7, 0, 0, 1004}, {"si 5760 split7 stagger8", 201600, 5760, 7, 0, 0, 1008},
        {"si 5760 split5 stagger4", 201600, 5760, 5, 0, 0, 1004},
        {"amidar 4800 split9", 120000, 4800, 9, 0, 0, 0}, {"amidar 4800 split9 stagger2", 120000, 4800, 9, 0, 0, 1002}, {"amidar 4800 split9 stagger4", 120000, 4800, 9, 0, 0, 1004},
    };
    for (int round = 0; round < 2; round++) {
        { const size_t b = (size_t)nf * 201600; float ms = timeit([&] { hipMemsetAsync(p, 1, b, 0); }, 5); printf("%-40s %8.3f ms %7.1f GB/s\n", "memset 13.2 GB", ms, b / ms / 1e6); }
        for (const Cfg& c : cfgs) {
            const int nun = c.frame / c.unit;
            const int grid = (nf * c.split + 3) / 4;
            float ms = timeit([&] { units<<<grid, 256, c.lds>>>(p, nf, c.frame / 16, c.unit / 16, nun, c.split, c.rot, c.nap); }, 5);
            printf("%-40s %8.3f ms %7.1f GB/s\n", c.name, ms, (double)nf * c.frame / ms / 1e6);
        }
    }
    hipFree(p); return 0;
}
