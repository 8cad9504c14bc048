// Pure-store emulation of the rasterisers' launch geometry: `split` waves share a frame of FRAME bytes cut into units of UNIT
// bytes; wave `part` stores units part, part + split, ... as 16-byte-per-lane stores (1 KiB per instruction).  Does the frame /
// unit geometry alone explain why SpaceInvaders' frames (201 600 B) stream slower than Breakout's (115 200 B)?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ __launch_bounds__(256) void units(uint4* out, int nframes, int frame16, int unit16, int nunits, int split, int rot)
{
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= nframes) return;
    uint4* fr = out + (size_t)rel * frame16;
    const uint4 v = make_uint4(rel, part, 2, 3);
    for (int q = part; q < nunits; q += split) {
        const int u = rot ? (int)(((unsigned)rel * 7u + (unsigned)q) % (unsigned)nunits) : q;
        uint4* dst = fr + (size_t)u * unit16;
        for (int i = lane; i < unit16; i += 64) dst[i] = v;
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
    struct Cfg { const char* name; int frame, unit, split, rot; };
    const Cfg cfgs[] = {
        {"breakout 115200/5760 split10 rot", 115200, 5760, 10, 1}, {"breakout 115200/5760 split10", 115200, 5760, 10, 0},
        {"breakout 115200/5760 split5", 115200, 5760, 5, 0},
        {"si 201600/5760 split5", 201600, 5760, 5, 0}, {"si 201600/5760 split7", 201600, 5760, 7, 0}, {"si 201600/5760 split7 rot", 201600, 5760, 7, 1},
        {"si 201600/5760 split12", 201600, 5760, 12, 0}, {"si 201600/5760 split18 rot", 201600, 5760, 18, 1},
        {"si 201600/6720 split6", 201600, 6720, 6, 0}, {"si 201600/6720 split10", 201600, 6720, 10, 0}, {"si 201600/6720 split15", 201600, 6720, 15, 0},
        {"si 201600/9600 split7", 201600, 9600, 7, 0}, {"si 201600/9600 split11", 201600, 9600, 11, 0},
        {"si 201600/4800 split14", 201600, 4800, 14, 0}, {"si 201600/4800 split21", 201600, 4800, 21, 0},
        {"si 201600/2880 split14", 201600, 2880, 14, 0}, {"si 201600/2880 split35", 201600, 2880, 35, 0},
        {"si 201600/14400 split7", 201600, 14400, 7, 0}, {"si 201600/20160 split5", 201600, 20160, 5, 0}, {"si 201600/20160 split10", 201600, 20160, 10, 0},
        {"amidar 120000/4800 split9", 120000, 4800, 9, 0}, {"amidar 120000/4800 split13", 120000, 4800, 13, 0}, {"amidar 120000/6000 split10", 120000, 6000, 10, 0},
        {"amidar 120000/2400 split25", 120000, 2400, 25, 0},
    };
    for (int round = 0; round < 2; round++) {
        { const size_t b = (size_t)nf * 201600; float ms = timeit([&] { hipMemsetAsync(p, 1, b, 0); }, 5); printf("%-40s %8.3f ms %7.1f GB/s\n", "memset 13.2 GB", ms, b / ms / 1e6); }
        for (const Cfg& c : cfgs) {
            const int nun = c.frame / c.unit;
            const int grid = (nf * c.split + 3) / 4;
            float ms = timeit([&] { units<<<grid, 256>>>(p, nf, c.frame / 16, c.unit / 16, nun, c.split, c.rot); }, 5);
            printf("%-40s %8.3f ms %7.1f GB/s\n", c.name, ms, (double)nf * c.frame / ms / 1e6);
        }
    }
    hipFree(p); return 0;
}
