// Does hipExtAnyOrderLaunch let a kernel start while the previous kernel of the SAME stream still runs on gfx950?  (diagnostic)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1;}}while(0)
__global__ void long_kernel(uint64_t* t, int spin_us) { if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = wall_clock64(); while (wall_clock64() - t[0] < (uint64_t)spin_us * 100) {} t[1] = wall_clock64(); } }
__global__ void short_kernel(uint64_t* t) { if (threadIdx.x == 0 && blockIdx.x == 0) { t[2] = wall_clock64(); } }
__global__ void third_kernel(uint64_t* t) { if (threadIdx.x == 0 && blockIdx.x == 0) { t[3] = wall_clock64(); } }
int main() {
    uint64_t* t; CK(hipMalloc((void**)&t, 64)); CK(hipMemset(t, 0, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int flags = 0; flags < 2; flags++) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(long_kernel, dim3(1), dim3(64), 0, s, t, 200);
            hipExtLaunchKernelGGL(short_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, t);
            hipLaunchKernelGGL(third_kernel, dim3(1), dim3(64), 0, s, t);
            CK(hipStreamSynchronize(s));
            uint64_t h[4]; CK(hipMemcpy(h, t, 32, hipMemcpyDeviceToHost));
            printf("flags %d: long runs 0 .. %.1f us; second kernel starts at %.1f us; third (ordinary) at %.1f us\n", flags, (h[1] - h[0]) * 0.01, ((double)h[2] - (double)h[0]) * 0.01, ((double)h[3] - (double)h[0]) * 0.01);
        }
    }
    return 0;
}
