// How many HIP streams of one process run kernels concurrently on this box?  N streams, one ~1 ms single-wave kernel each
// (a fixed-length dependent FMA chain: it ends by itself), wall time of the batch: ~1 ms while every stream has a hardware
// queue of its own, k ms once k streams share one.   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void chain_kernel(double *out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
    if (x == 123.0) out[0] = x;
}
int main(int argc, char **argv) {
    const int high = argc > 1 ? atoi(argv[1]) : 0;     // number of high-priority streams among them (created first)
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    double *d;
    hipMalloc(&d, 64);
    const int iters = 300000;
    for (int N = 1; N <= 12; ++N) {
        std::vector<hipStream_t> st(N);
        for (int i = 0; i < N; ++i) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, i < high ? hi : lo);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, st[i], d, 1000);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, st[i], d, iters);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("N=%2d streams (%d high priority): %.2f ms\n", N, high < N ? high : N, ms);
        for (auto s : st) hipStreamDestroy(s);
    }
    return 0;
}
