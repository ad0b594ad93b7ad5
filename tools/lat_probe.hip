// Dependent-chain latency of the building blocks of the latency-bound eigensolver kernels (one wave on one CU, gfx950),
// in shader cycles measured with s_memtime.   hipcc -O3 --offload-arch=gfx950 -Igpcsd_amd/csrc tools/lat_probe.hip -o tools/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "devutil.hpp"
using namespace gpcsd;

enum { OP_FMA, OP_ADD, OP_MUL, OP_WAVESUM, OP_ROW16, OP_LDS_RT, OP_RSQ, OP_RCP, OP_FASTRCP, OP_SQRT, OP_DIV, OP_DPP1, OP_READLANE, OP_CNDMASK, OP_LDS_B128, OP_BARRIER };

template <int OP>
__global__ void chain(double *out, long *cyc, int iters, double seed) {
    __shared__ double lds[1024];
    double a = seed + 1e-9 * threadIdx.x;
    lds[threadIdx.x] = a;
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {                       // 16 links per trip: the ~28-cycle loop overhead is amortised
        if (OP == OP_FMA) a = fma(a, 0.999, 1e-3);
        if (OP == OP_ADD) a = a + 1e-3;
        if (OP == OP_MUL) a = a * 1.0000001;
        if (OP == OP_WAVESUM) a = wave_sum(a) * (1.0 / 64);
        if (OP == OP_ROW16) { a += dpp_mov<0xB1>(a); a += dpp_mov<0x4E>(a); a += dpp_mov<0x141>(a); a += dpp_mov<0x140>(a); a *= 1.0 / 16; }
        if (OP == OP_LDS_RT) { lds[threadIdx.x] = a; a = lds[threadIdx.x ^ 1]; }
        if (OP == OP_RSQ) a = __builtin_amdgcn_rsq(a) + 1.0;
        if (OP == OP_RCP) a = __builtin_amdgcn_rcp(a) + 1.0;
        if (OP == OP_FASTRCP) a = fast_rcp(a) + 1.0;
        if (OP == OP_SQRT) a = sqrt(a) + 1.0;
        if (OP == OP_DIV) a = 1.0 / a + 1.0;
        if (OP == OP_DPP1) a = dpp_mov<0xB1>(a);
        if (OP == OP_READLANE) a = lane_get(a, 5) + 1e-9 * threadIdx.x;
        if (OP == OP_CNDMASK) a = (__double2loint(a) & 1) ? a : -a;
        if (OP == OP_LDS_B128) { const double2 t = *reinterpret_cast<const double2 *>(lds + 2 * ((__double2loint(a) & 255))); a = t.x + t.y; }
        if (OP == OP_BARRIER) { __syncthreads(); a += 1.0; }
      }
    }
    const long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char *name, int threads = 64, double seed = 1.5) {
    double *out;
    long *cyc, h = 0;
    (void)hipMalloc(&out, 8 * 1024);
    (void)hipMalloc(&cyc, 8);
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, seed);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s threads=%4d: %7.1f cycles per link\n", name, threads, (double)h / (16.0 * iters));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run<OP_FMA>("v_fma_f64");
    run<OP_ADD>("v_add_f64");
    run<OP_MUL>("v_mul_f64");
    run<OP_DPP1>("dpp_mov<quad_perm> (2 x v_mov_b32 dpp)");
    run<OP_CNDMASK>("select on a double (and + cmp + 2 cndmask)");
    run<OP_READLANE>("lane_get + v_add_f64");
    run<OP_ROW16>("row16 sum (4 dpp stages) + mul");
    run<OP_WAVESUM>("wave_sum + mul");
    run<OP_LDS_RT>("LDS write b64 -> read b64 (same wave)");
    run<OP_LDS_B128>("LDS read b128 (address dependent) + add");
    run<OP_RSQ>("v_rsq_f64 + add");
    run<OP_RCP>("v_rcp_f64 + add");
    run<OP_FASTRCP>("fast_rcp + add");
    run<OP_SQRT>("IEEE sqrt + add");
    run<OP_DIV>("IEEE 1/x + add");
    run<OP_BARRIER>("s_barrier + add", 64);
    run<OP_BARRIER>("s_barrier + add", 256);
    run<OP_BARRIER>("s_barrier + add", 768);
    run<OP_BARRIER>("s_barrier + add", 1024);
    return 0;
}
