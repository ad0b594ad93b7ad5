// Does a stream that waits for an event recorded IN FRONT of a long kernel of another stream start before that kernel ends?
//   stream A: short kernel, record E, long kernel (spins ~500 us);   stream B: wait E, stamp kernel.
// Prints when B's kernel ran relative to A's long kernel (device wall clock, 100 MHz).
//   hipcc --offload-arch=gfx950 -O2 tools/event_probe.hip -o tools/event_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void stamp(unsigned long long *o) { if (threadIdx.x == 0) *o = wall_clock64(); }
__global__ void spin(unsigned long long *o, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) o[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) o[1] = wall_clock64();
}
int main(int argc, char **argv) {
    const int flags = argc > 1 ? atoi(argv[1]) : 0;       // 1: hipEventDisableTiming
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, flags ? hipEventDisableTiming : hipEventDefault));
    unsigned long long *d, h[4];
    CK(hipMalloc(&d, 4 * sizeof(*d)));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(d, 0, 4 * sizeof(*d)));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, d + 3);
        CK(hipEventRecord(e, a));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, d, 50000ull);
        CK(hipStreamWaitEvent(b, e, 0));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, d + 2);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("flags %d rep %d: long kernel %.1f us; B's kernel ran %.1f us after the long kernel STARTED (%.1f us relative to its end)\n", flags, rep,
               0.01 * (double)(h[1] - h[0]), 0.01 * ((double)h[2] - (double)h[0]), 0.01 * ((double)h[2] - (double)h[1]));
    }
    return 0;
}
