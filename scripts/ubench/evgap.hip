// What does an event dependency between two streams cost the stream that records / waits?  Gaps between consecutive kernels of
// stream s (end of A -> start of B) measured with the device's wall clock, for several ways of letting a kernel C on another
// stream run after A.  (diagnostic)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1;}}while(0)
__global__ void stamp_kernel(uint64_t* t, int slot, int spin_us) { if (threadIdx.x == 0 && blockIdx.x == 0) { const uint64_t a = wall_clock64(); t[2 * slot] = a; while (wall_clock64() - a < (uint64_t)spin_us * 100) {} t[2 * slot + 1] = wall_clock64(); } }
int main() {
    uint64_t* t; CK(hipMalloc((void**)&t, 4096)); CK(hipMemset(t, 0, 4096));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev, ev2; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    const char* names[] = {"A ; B", "A ; record ; B", "A ; record ; s2 waits, C on s2 ; B", "A launched with stopEvent ; s2 waits, C on s2 ; B",
                           "A ; record ; s2 waits, C on s2, record on s2 ; B ; s waits for C ; A' (the gather loop)",
                           "A with stopEvent ; s2 waits, C on s2, record on s2 ; B ; s waits for C ; A'"};
    for (int v = 0; v < 6; v++) {
        double gap_ab = 0, gap_ba = 0; int cnt = 0, bad = 0;
        for (int rep = 0; rep < 40; rep++) {
            // A = slot 0, B = slot 1, C = slot 2, A' = slot 3
            if (v == 3 || v == 5) hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, ev, 0, t, 0, 20);
            else hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, t, 0, 20);
            if (v == 1 || v == 2 || v == 4) CK(hipEventRecord(ev, s));
            if (v >= 2) { CK(hipStreamWaitEvent(s2, ev, 0)); hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, t, 2, 5); }
            if (v >= 4) CK(hipEventRecord(ev2, s2));
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, t, 1, 100);
            if (v >= 4) CK(hipStreamWaitEvent(s, ev2, 0));
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, t, 3, 20);
            CK(hipDeviceSynchronize());
            uint64_t h[8]; CK(hipMemcpy(h, t, 64, hipMemcpyDeviceToHost));
            if (rep >= 5) { gap_ab += ((double)h[2] - (double)h[1]) * 0.01; gap_ba += ((double)h[6] - (double)h[3]) * 0.01; cnt++; }
            if (v >= 2 && h[4] < h[1]) bad++;                       // C started before A ended: the dependency did not hold
            if (v >= 4 && h[6] < h[5]) bad++;                       // A' started before C ended
        }
        printf("%-92s end of A -> start of B %5.1f us; end of B -> start of A' %5.1f us; order violations %d\n", names[v], gap_ab / cnt, gap_ba / cnt, bad);
    }
    return 0;
}
