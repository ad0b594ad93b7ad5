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


// Device-resident timing of the blocked Cholesky (chol.hip) at order n: a well-conditioned SPD test matrix is generated on the
// device before every repetition (A_ij = exp(-|i - j| / 64) + [i == j]: an exponential-kernel Gram matrix plus a unit nugget),
// HIP events on the call's stream bracket potrf_device alone.  ms_out: mean over reps (after one untimed repetition).  With
// profiling enabled (gpcsd_prof_enable(ctx, 1)) the scopes potrf_diag_block / potrf_panel / potrf_syrk* / potrf_inv split it.
__global__ void spd_test_matrix_kernel(double *A, int n) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < (long)n * n) {
        const int i = (int)(e / n), j = (int)(e % n);
        const double d = (double)(i > j ? i - j : j - i);
        A[e] = exp(-d / 64.0) + (i == j ? 1.0 : 0.0);
    }
}

extern "C" int gpcsd_potrf_gate_timeouts(gpcsd_ctx *c, long *count) {
    GP_API_BEGIN(c)
    GP_REQUIRE(count != nullptr, -3, "potrf_gate_timeouts: null output");
    unsigned int w[2] = {0u, 0u};
    c->sync();
    if (c->h_chol_flag) {
        GP_HIP(hipDeviceSynchronize());
        c->copy_out(w, c->h_chol_flag, sizeof(w), c->stream);
    }
    *count = (long)w[1];
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_potrf_bench(gpcsd_ctx *c, int n, int reps, double *ms_out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(ms_out != nullptr && n >= 1 && n <= 32768 && reps >= 1, -3, "potrf_bench: bad arguments");
    double *A = c->buf<double>("dense_K", (size_t)n * n);
    int *st = c->buf<int>("status", 4);
    hipStream_t s = c->stream;
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), s));
    hipEvent_t e0 = c->get_event(), e1 = c->get_event();
    double total = 0.0;
    for (int r = 0; r <= reps; ++r) {
        hipLaunchKernelGGL(spd_test_matrix_kernel, dim3(ceil_div((long)n * n, 256)), dim3(256), 0, s, A, n);
        GP_HIP(hipEventRecord(e0, s));
        potrf_device(c, A, n, st, s);
        GP_HIP(hipEventRecord(e1, s));
        GP_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        GP_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
    }
    c->event_pool.push_back(e0);
    c->event_pool.push_back(e1);
    *ms_out = total / reps;
    return finish_status(c, st);
    GP_API_END(c)
}

// Phase split of the 128 x 128 factor + invert workgroup (chol.hip: diag128_kernel) on the leading block of the SPD test matrix:
// out10 = {load, serial panels, rank-16 updates, store L, diagonal inverses, level 16, level 32, level 64, store X, total} in us.
extern "C" int gpcsd_potrf_diag_probe(gpcsd_ctx *c, double *out10) {
    GP_API_BEGIN(c)
    GP_REQUIRE(out10 != nullptr, -3, "null output");
    const int n = 128;
    double *A = c->buf<double>("dense_K", (size_t)n * n);
    double *X = c->buf<double>("chol_Linv", (size_t)256 * 256);
    int *st = c->buf<int>("status", 4);
    unsigned long long *clk = c->buf<unsigned long long>("diag_clk", 16);
    hipStream_t s = c->stream;
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), s));
    unsigned long long h[16];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spd_test_matrix_kernel, dim3(ceil_div((long)n * n, 256)), dim3(256), 0, s, A, n);
        GP_HIP(hipMemsetAsync(clk, 0, 16 * sizeof(unsigned long long), s));
        potrf_diag128_probe(c, A, n, X, st, clk, s);
        c->download(h, clk, sizeof(h));
        c->sync();
    }
    int rate_khz = 0;
    GP_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c->device));
    const double us = 1e3 / (double)rate_khz;
    out10[0] = (h[1] - h[0]) * us;
    out10[1] = h[2] * us;
    out10[2] = h[3] * us;
    out10[3] = (h[4] - h[1]) * us - out10[1] - out10[2];
    out10[4] = (h[5] - h[4]) * us;
    out10[5] = (h[6] - h[5]) * us;
    out10[6] = (h[7] - h[6]) * us;
    out10[7] = (h[8] - h[7]) * us;
    out10[8] = (h[9] - h[8]) * us;
    out10[9] = (h[9] - h[0]) * us;
    return finish_status(c, st);
    GP_API_END(c)
}
