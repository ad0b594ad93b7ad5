// Part of capi.hip (included there: one translation unit, so the file-local helpers of capi.hip are in scope) --
// measurement: event scopes, tail clock stamps, MFMA / HBM probes.

// ------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------
extern "C" int gpcsd_prof_enable(gpcsd_ctx *c, int on) {
    GP_API_BEGIN(c)
    // 0 off.  1: fenced -- every fused call synchronises and collects its scopes, asynchronous calls are evaluated at once,
    // chains run eagerly (one scope per kernel family).  2: asynchronous -- scopes record their events on the streams they run
    // on and nothing else changes: queued and paired calls stay queued and paired; chains run eagerly so that the scopes inside
    // them (sytrd_rtail, eigh_stedc, ...) see their kernels.  3: as 2 with the chains replayed as hipGraphs, as in production:
    // only the scopes around whole chains and the GEMM tails record.  Modes 2 / 3 are collected by gpcsd_prof_get (which waits
    // for the recorded events).
    GP_REQUIRE(on >= 0 && on <= 3, -3, "prof_enable: mode must be 0..3");
    if (on >= 2 && !c->tail_clk_host) {
        const size_t bytes = (size_t)3 * 2 * gpcsd_ctx::TAIL_CLK_WGS * sizeof(unsigned long long);
        GP_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->tail_clk_host), bytes, hipHostMallocMapped));
        memset(c->tail_clk_host, 0, bytes);
        GP_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->tail_clk_dev), c->tail_clk_host, 0));
    }
    c->prof_mode = on;
    c->prof_on = (on != 0);
    return 0;
    GP_API_END(c)
}

// Duration of the last tridiagonalisation-tail launch of a chain from the workgroups' own wall-clock stamps (region 0: temporal
// chain, 1: spatial chain, 2: other): last end - first start over its workgroups, in ms; *nwg = workgroups, *flops = the (4/3)
// T^3 count of the launch.  Valid once the chain has finished (e.g. after gpcsd_loglik_parts_wait); profiling modes 2 / 3.
extern "C" int gpcsd_prof_tail_clock(gpcsd_ctx *c, int region, double *ms, int *nwg, double *flops) {
    GP_API_BEGIN(c)
    GP_REQUIRE(region >= 0 && region < 3 && ms, -3, "prof_tail_clock: bad arguments");
    GP_REQUIRE(c->tail_clk_host != nullptr, -4, "prof_tail_clock: profiling mode 2 / 3 has not been enabled on this context");
    int rate_khz = 0;
    GP_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c->device));
    GP_REQUIRE(rate_khz > 0, -5, "prof_tail_clock: the device reports no wall clock rate");
    const int n = c->tail_clk_count[region];
    const volatile unsigned long long *p = c->tail_clk_host + (size_t)region * 2 * gpcsd_ctx::TAIL_CLK_WGS;
    unsigned long long t0 = ~0ull, t1 = 0ull;
    for (int i = 0; i < n; ++i) {
        if (p[2 * i] == 0 || p[2 * i + 1] == 0) continue;          // (a workgroup that returned early stamps nothing)
        t0 = p[2 * i] < t0 ? p[2 * i] : t0;
        t1 = p[2 * i + 1] > t1 ? p[2 * i + 1] : t1;
    }
    *ms = (t1 > t0 && t0 != ~0ull) ? (double)(t1 - t0) / (double)rate_khz : 0.0;
    if (nwg) *nwg = n;
    if (flops) *flops = c->tail_clk_flops[region];
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_prof_reset(gpcsd_ctx *c) {
    GP_API_BEGIN(c)
    c->sync();
    c->prof_collect();
    c->prof.clear();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_prof_get(gpcsd_ctx *c, const char *name, double *ms, long *count, double *flops) {
    GP_API_BEGIN(c)
    c->prof_collect();
    auto it = c->prof.find(name ? name : "");
    if (it == c->prof.end()) return -2;
    if (ms) *ms = it->second.ms;
    if (count) *count = it->second.count;
    if (flops) *flops = it->second.flops;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_prof_names(gpcsd_ctx *c, char *buf, int buflen) {
    GP_API_BEGIN(c)
    std::string sres;
    for (auto &kv : c->prof) {
        if (!sres.empty()) sres += ";";
        sres += kv.first;
    }
    if (!buf || buflen <= 0) return (int)sres.size();
    snprintf(buf, buflen, "%s", sres.c_str());
    return 0;
    GP_API_END(c)
}

typedef double d4 __attribute__((ext_vector_type(4)));

// Back-to-back v_mfma_f64_16x16x4_f64 with the accumulators pinned to VGPRs (inline asm keeps hipcc from shuttling
// them through AGPRs every iteration); 4 independent chains per wave, 4 waves per SIMD.
#define GP_MF(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double *out, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + 1e-3 * threadIdx.x, y = 0.7 - 1e-3 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        GP_MF(a0); GP_MF(a1); GP_MF(a2); GP_MF(a3);
        GP_MF(a0); GP_MF(a1); GP_MF(a2); GP_MF(a3);
    }
    d4 r = a0 + a1 + a2 + a3;
    if (r[0] == 123.456) out[blockIdx.x] = r[0] + r[1] + r[2] + r[3];
}

extern "C" int gpcsd_mfma_f64_peak(gpcsd_ctx *c, double *tflops) {
    GP_API_BEGIN(c)
    GP_REQUIRE(tflops != nullptr, -3, "null output");
    double *o = c->buf<double>("peak_out", 4096);
    const int iters = 20000, blocks = 256 * 4;
    hipEvent_t e0 = c->get_event(), e1 = c->get_event();
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, c->stream, o, 100);   // warm-up
    GP_HIP(hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, c->stream, o, iters);
    GP_HIP(hipEventRecord(e1, c->stream));
    GP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GP_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    c->event_pool.push_back(e0);
    c->event_pool.push_back(e1);
    return 0;
    GP_API_END(c)
}

__global__ void copy_peak_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

extern "C" int gpcsd_hbm_copy_peak(gpcsd_ctx *c, long bytes, double *gbs) {
    GP_API_BEGIN(c)
    GP_REQUIRE(gbs != nullptr && bytes >= (1 << 20), -3, "hbm_copy_peak: need >= 1 MiB");
    const long n = bytes / 16;
    double2 *a = (double2 *)c->buf<double>("peak_a", n * 2);
    double2 *b = (double2 *)c->buf<double>("peak_b", n * 2);
    GP_HIP(hipMemsetAsync(a, 0, n * 16, c->stream));
    hipEvent_t e0 = c->get_event(), e1 = c->get_event();
    hipLaunchKernelGGL(copy_peak_kernel, dim3(2048), dim3(256), 0, c->stream, (const double2 *)a, b, n);
    GP_HIP(hipEventRecord(e0, c->stream));
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(copy_peak_kernel, dim3(2048), dim3(256), 0, c->stream, (const double2 *)a, b, n);
    GP_HIP(hipEventRecord(e1, c->stream));
    GP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GP_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbs = 2.0 * n * 16.0 * reps / (ms * 1e-3) / 1e9;
    c->event_pool.push_back(e0);
    c->event_pool.push_back(e1);
    return 0;
    GP_API_END(c)
}
