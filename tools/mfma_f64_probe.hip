// Probe: v_mfma_f64_16x16x4_f64 issue rate / dependent latency / sustained clock on MI355X (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_probe tools/mfma_f64_probe.hip
// Accumulators are pinned to VGPRs with inline asm so the loop holds nothing but MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MF(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

template <int NACC>
__global__ __launch_bounds__(256) void probe(double *out, unsigned long long *cyc, unsigned long long *rt, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    const double x = 1.0 + 1e-3 * threadIdx.x, y = 0.7 - 1e-3 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / NACC; ++u) {
            MF(a0);
            if (NACC > 1) MF(a1);
            if (NACC > 2) { MF(a2); MF(a3); }
            if (NACC > 4) { MF(a4); MF(a5); MF(a6); MF(a7); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    d4 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r[0] == 123.456) out[blockIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC>
void run(int waves_per_block, int blocks, int iters) {
    double *out; unsigned long long *cyc, *rt;
    (void)hipMalloc(&out, blocks * 8); (void)hipMalloc(&cyc, blocks * 8); (void)hipMalloc(&rt, blocks * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NACC><<<blocks, waves_per_block * 64>>>(out, cyc, rt, 100);
    (void)hipEventRecord(e0);
    probe<NACC><<<blocks, waves_per_block * 64>>>(out, cyc, rt, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(blocks), hr(blocks);
    (void)hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0, r = 0; for (int i = 0; i < blocks; ++i) { c += hc[i]; r += hr[i]; }
    c /= blocks; r /= blocks;
    double nm = (double)iters * 8;
    double flops = (double)blocks * waves_per_block * nm * 2048.0;
    printf("indep_acc=%d waves/SIMD=%.0f: %6.2f TF/s  cycles/mfma/wave=%6.1f  cycles/mfma/SIMD=%6.1f  clock=%.0f MHz (%.2f ms)\n", NACC,
           blocks * waves_per_block / 1024.0, flops / (ms * 1e-3) / 1e12, c / nm, c / nm / (blocks * waves_per_block / 1024.0), c / r * 100.0, ms);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(rt);
}

int main() {
    run<1>(4, 256, 20000); run<2>(4, 256, 20000); run<4>(4, 256, 20000); run<8>(4, 256, 20000);
    run<1>(4, 512, 20000); run<4>(4, 512, 20000); run<8>(4, 512, 20000);
    run<1>(4, 1024, 20000); run<4>(4, 1024, 20000);
    run<4>(4, 2048, 20000);
    run<8>(4, 256, 400000);   // long run: steady clock under load
    return 0;
}
