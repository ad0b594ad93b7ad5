// Probe: cost of a chain of dependent small kernels on MI355X, by what the kernel does.
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_probe tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { double *a, *b, *v, *y; int n; };
__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

template <int MODE>
__global__ __launch_bounds__(256) void k(Args A, int step) {
    __shared__ double sv[1024], red[4];
    const int tid = threadIdx.x, n = A.n;
    if (MODE == 0) return;                                    // empty
    const double *in = (step & 1) ? A.b : A.a;
    double *out = (step & 1) ? A.a : A.b;
    const int row = blockIdx.x * 8 + (tid >> 6) * 2;
    double x[2][8];
    for (int r = 0; r < 2; ++r) for (int q = 0; q < 8; ++q) { int j = (tid & 63) + 64 * q; x[r][q] = (j < n) ? in[(long)(row + r) * n + j] : 0.0; }
    double pv = (tid < n) ? A.v[tid] : 0.0, pv2 = (tid + 256 < n) ? A.v[tid + 256] : 0.0;
    if (MODE == 1) {                                          // loads + stores only
        for (int r = 0; r < 2; ++r) for (int q = 0; q < 8; ++q) { int j = (tid & 63) + 64 * q; if (j < n) out[(long)(row + r) * n + j] = x[r][q] + pv; }
        return;
    }
    // MODE >= 2: two block reductions + LDS broadcast
    double p = pv * pv2;
    p = wave_sum(p); __syncthreads(); if ((tid & 63) == 0) red[tid >> 6] = p; __syncthreads();
    double dot = red[0] + red[1] + red[2] + red[3];
    sv[tid] = pv * dot; sv[tid + 256] = pv2 * dot; __syncthreads();
    double p2 = sv[(tid + 7) & 511];
    p2 = wave_sum(p2); __syncthreads(); if ((tid & 63) == 0) red[tid >> 6] = p2; __syncthreads();
    double nrm = red[0] + red[1] + red[2] + red[3];
    double sc = 1.0 / (sqrt(nrm * nrm + 1.0) + 1.0);
    sv[tid] = sv[tid] * sc; __syncthreads();
    double acc0 = 0, acc1 = 0;
    for (int q = 0; q < 8; ++q) { int j = (tid & 63) + 64 * q; if (j < n) { double s = sv[j & 511];
        double a0 = x[0][q] - s, a1 = x[1][q] - s; out[(long)row * n + j] = a0; out[(long)(row + 1) * n + j] = a1; acc0 += a0 * s; acc1 += a1 * s; } }
    acc0 = wave_sum(acc0); acc1 = wave_sum(acc1);
    if ((tid & 63) == 0) { A.y[row] = acc0; A.y[row + 1] = acc1; }
    if (MODE == 3 && blockIdx.x == 0) for (int j = tid; j < n; j += 256) A.v[j] = sv[j & 511] * 1e-3 + 0.5;
}

template <int MODE> void run(const char *name, Args A, int grid, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) k<MODE><<<grid, 256>>>(A, i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k<MODE><<<grid, 256>>>(A, i);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // same chain through a graph
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < reps; ++i) k<MODE><<<grid, 256, 0, s>>>(A, i);
    (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    float msg; (void)hipEventElapsedTime(&msg, e0, e1);
    printf("%-28s grid=%3d: eager %.2f us/launch, graph %.2f us/launch\n", name, grid, 1e3 * ms / reps, 1e3 * msg / reps);
}

int main() {
    const int n = 512; Args A; A.n = n;
    (void)hipMalloc(&A.a, n * n * 8); (void)hipMalloc(&A.b, n * n * 8); (void)hipMalloc(&A.v, 1024 * 8); (void)hipMalloc(&A.y, 1024 * 8);
    (void)hipMemset(A.a, 0, n * n * 8); (void)hipMemset(A.b, 0, n * n * 8); (void)hipMemset(A.v, 0, 1024 * 8);
    for (int grid : {64, 16}) {
        run<0>("empty", A, grid, 500);
        run<1>("loads+stores", A, grid, 500);
        run<2>("loads+2 reductions+stores", A, grid, 500);
        run<3>("full (+v handoff)", A, grid, 500);
    }
    return 0;
}
