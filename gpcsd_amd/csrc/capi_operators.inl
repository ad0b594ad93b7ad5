// Part of capi.hip (included there: one translation unit, so the file-local helpers of capi.hip are in scope) --
// operator surface (stand-alone, host in / host out), host temporal Grams, sharding helpers, context knobs.

// ------------------------------------------------------------------------------------------------
// operator surface
// ------------------------------------------------------------------------------------------------
extern "C" int gpcsd_b_fwd_1d(gpcsd_ctx *c, const double *r, long n, double R, double *out) {
    GP_API_BEGIN(c)
    if (n <= 0) return 0;
    double *d = c->upload<double>("op_in0", r, n);
    double *o = c->buf<double>("op_out", n);
    k_b_fwd_1d(c, d, n, R, o, c->stream);
    c->download(out, o, n * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_trad_csd(gpcsd_ctx *c, const double *lfp, long n_outer, long n_axis, long n_inner, int edge_nan, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(n_outer >= 0 && n_axis >= 0 && n_inner >= 0, -3, "trad_csd: negative extent");
    const long n = n_outer * n_axis * n_inner;
    if (n == 0) return 0;
    GP_REQUIRE(lfp && out, -3, "trad_csd: null array");
    double *d = c->upload<double>("op_in0", lfp, n);
    double *o = c->buf<double>("op_out", n);
    k_second_diff(c, d, n_outer, n_axis, n_inner, edge_nan ? -__builtin_nan("") : -0.0, o, c->stream);
    c->download(out, o, n * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_b_fwd_2d(gpcsd_ctx *c, const double *d1, const double *d2, const double *w, long n, double R, double eps,
                              double *out) {
    GP_API_BEGIN(c)
    if (n <= 0) return 0;
    double *dd1 = nullptr, *dd2 = nullptr, *dw = nullptr;
    if (w) dw = c->upload<double>("op_in0", w, n);
    else {
        GP_REQUIRE(d1 && d2, -3, "b_fwd_2d: need delta1 and delta2 when w is NULL");
        dd1 = c->upload<double>("op_in0", d1, n);
        dd2 = c->upload<double>("op_in1", d2, n);
    }
    double *o = c->buf<double>("op_out", n);
    k_b_fwd_2d(c, dd1, dd2, dw, n, R, eps, o, c->stream);
    c->download(out, o, n * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_gram_temporal(gpcsd_ctx *c, int kind, const double *t, int n, const double *tp, int m, double ell,
                                   double sigma2, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(kind == GPCSD_KIND_SE || kind == GPCSD_KIND_MATERN, -3, "unknown temporal kernel kind %d", kind);
    if (n <= 0 || m <= 0) return 0;
    double *dt = c->upload<double>("op_in0", t, n);
    double *dtp = c->upload<double>("op_in1", tp, m);
    double *o = c->buf<double>("op_out", (size_t)n * m);
    k_temporal_gram(c, 1, &kind, &ell, &sigma2, dt, n, dtp, m, o, c->stream);
    c->download(out, o, (size_t)n * m * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_ks_csd_1d(gpcsd_ctx *c, const double *x, int nx, double ell, double *out) {
    GP_API_BEGIN(c)
    double *dx = c->upload<double>("op_in0", x, nx);
    double *o = c->buf<double>("op_out", (size_t)nx * nx);
    k_se_1d(c, dx, nx, dx, nx, ell, o, c->stream);
    c->download(out, o, (size_t)nx * nx * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_ks_csd_2d(gpcsd_ctx *c, const double *xy, int nx, double ell1, double ell2, double *out) {
    GP_API_BEGIN(c)
    double *dx = c->upload<double>("op_in0", xy, (size_t)nx * 2);
    double *o = c->buf<double>("op_out", (size_t)nx * nx);
    k_se_2d(c, dx, nullptr, nx, 0, dx, nullptr, nx, 0, ell1, ell2, o, c->stream);
    c->download(out, o, (size_t)nx * nx * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

static Geo upload_geo_1d(gpcsd_ctx *c, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl) {
    GP_REQUIRE(x && gl_x && gl_w && nx > 0 && ngl > 0, -3, "bad 1D geometry arguments");
    Geo g;
    g.dim = 1; g.nx = nx; g.ngl1 = ngl;
    g.x = c->upload<double>("op_x", x, nx);
    g.gx1 = c->upload<double>("op_gx1", gl_x, ngl);
    g.gw1 = c->upload<double>("op_gw1", gl_w, ngl);
    return g;
}

static Geo upload_geo_2d(gpcsd_ctx *c, const double *xy, int nx, const double *gx1, const double *gw1, int ngl1,
                         const double *gx2, const double *gw2, int ngl2) {
    GP_REQUIRE(xy && gx1 && gw1 && gx2 && gw2 && nx > 0 && ngl1 > 0 && ngl2 > 0, -3, "bad 2D geometry arguments");
    Geo g;
    g.dim = 2; g.nx = nx; g.ngl1 = ngl1; g.ngl2 = ngl2;
    g.x = c->upload<double>("op_x", xy, (size_t)nx * 2);
    g.gx1 = c->upload<double>("op_gx1", gx1, ngl1);
    g.gw1 = c->upload<double>("op_gw1", gw1, ngl1);
    g.gx2 = c->upload<double>("op_gx2", gx2, ngl2);
    g.gw2 = c->upload<double>("op_gw2", gw2, ngl2);
    return g;
}

extern "C" int gpcsd_kphi_1d(gpcsd_ctx *c, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl, double R,
                             double ell, const double *xp, int nxp, double *out) {
    GP_API_BEGIN(c)
    Geo g = upload_geo_1d(c, x, nx, gl_x, gl_w, ngl);
    const double *dxp = nullptr;
    if (xp) {
        GP_REQUIRE(nxp > 0, -3, "kphi_1d: nxp must be positive");
        dxp = c->upload<double>("op_xp", xp, nxp);
    }
    const int n2 = xp ? nxp : nx;
    double *o = c->buf<double>("op_out", (size_t)nx * n2);
    build_kphi(c, g, R, 0.0, &ell, dxp, nxp, 0.0, o, c->stream);
    c->download(out, o, (size_t)nx * n2 * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_kphig_1d(gpcsd_ctx *c, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl,
                              const double *z, int nz, double R, double ell, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(z && nz > 0, -3, "kphig_1d: bad z");
    Geo g = upload_geo_1d(c, x, nx, gl_x, gl_w, ngl);
    double *dz = c->upload<double>("op_xp", z, nz);
    double *o = c->buf<double>("op_out", (size_t)nx * nz);
    build_kphig(c, g, R, 0.0, &ell, dz, nz, o, c->stream);
    c->download(out, o, (size_t)nx * nz * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_kphi_2d(gpcsd_ctx *c, const double *xy, int nx, const double *gl_x1, const double *gl_w1, int ngl1,
                             const double *gl_x2, const double *gl_w2, int ngl2, double R, double eps, double ell1, double ell2,
                             const double *xp, int nxp, double *out) {
    GP_API_BEGIN(c)
    Geo g = upload_geo_2d(c, xy, nx, gl_x1, gl_w1, ngl1, gl_x2, gl_w2, ngl2);
    const double *dxp = nullptr;
    if (xp) {
        GP_REQUIRE(nxp > 0, -3, "kphi_2d: nxp must be positive");
        dxp = c->upload<double>("op_xp", xp, (size_t)nxp * 2);
    }
    const int n2 = xp ? nxp : nx;
    const double ell[2] = {ell1, ell2};
    double *o = c->buf<double>("op_out", (size_t)nx * n2);
    build_kphi(c, g, R, eps, ell, dxp, nxp, 0.0, o, c->stream);
    c->download(out, o, (size_t)nx * n2 * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_kphig_2d(gpcsd_ctx *c, const double *xy, int nx, const double *gl_x1, const double *gl_w1, int ngl1,
                              const double *gl_x2, const double *gl_w2, int ngl2, const double *z, int nz, double R, double eps,
                              double ell1, double ell2, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(z && nz > 0, -3, "kphig_2d: bad z");
    Geo g = upload_geo_2d(c, xy, nx, gl_x1, gl_w1, ngl1, gl_x2, gl_w2, ngl2);
    double *dz = c->upload<double>("op_xp", z, (size_t)nz * 2);
    const double ell[2] = {ell1, ell2};
    double *o = c->buf<double>("op_out", (size_t)nx * nz);
    build_kphig(c, g, R, eps, ell, dz, nz, o, c->stream);
    c->download(out, o, (size_t)nx * nz * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_eigh(gpcsd_ctx *c, const double *A, int n, double *evals, double *evecs) {
    GP_API_BEGIN(c)
    GP_REQUIRE(A && evals && evecs && n > 0, -3, "eigh: bad arguments");
    double *dA = c->upload<double>("op_in0", A, (size_t)n * n);
    double *dw = c->buf<double>("op_w", n);
    double *dV = c->buf<double>("op_out", (size_t)n * n);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), c->stream));
    eigh_device(c, dA, n, dw, dV, st, c->stream, "eigh");
    c->download(evals, dw, n * sizeof(double));
    c->download(evecs, dV, (size_t)n * n * sizeof(double));
    return finish_status(c, st);
    GP_API_END(c)
}

// numpy.linalg.eigh (utility_functions.py:58-59) of a matrix the CALLER vouches is positive semi-definite (a Gram matrix): the one
// way a caller's matrix may take the tridiagonalisation's rank-revealing early exit (gpcsd_tail_early_exit; n <= 192 only).
extern "C" int gpcsd_eigh_psd(gpcsd_ctx *c, const double *A, int n, double *evals, double *evecs) {
    if (!c) return -3;
    struct Claim {
        gpcsd_ctx *c;
        explicit Claim(gpcsd_ctx *cc) : c(cc) { c->claim_psd = true; }
        ~Claim() { c->claim_psd = false; }
    } claim(c);
    return gpcsd_eigh(c, A, n, evals, evecs);
}

// `count` independent symmetric matrices of the same order in ONE chain of launches (the replicated-class machinery behind
// gpcsd_loglik_grad_batch, exposed for tests): A (count, n, n) -> evals (count, n), evecs (count, n, n), status (count):
// 0 ok, > 0 numerical failure of that matrix alone.
extern "C" int gpcsd_eigh_batch(gpcsd_ctx *c, const double *A, int n, int count, double *evals, double *evecs, int *status) {
    GP_API_BEGIN(c)
    GP_REQUIRE(A && evals && evecs && status && n > 0 && count > 0, -3, "eigh_batch: bad arguments");
    const size_t nn = (size_t)n * n;
    double *dA = c->upload<double>("op_in0", A, nn * count);
    double *dw = c->buf<double>("op_w", (size_t)n * count);
    double *dV = c->buf<double>("op_out", nn * count);
    int *st = c->buf<int>("status_batch", (size_t)count);
    GP_HIP(hipMemsetAsync(st, 0, (size_t)count * sizeof(int), c->stream));
    eigh_pair_device(c, dA, n, dw, dV, nullptr, nullptr, 0, nullptr, nullptr, nullptr, st, c->stream, true, count, 1);
    c->download(evals, dw, (size_t)n * count * sizeof(double));
    c->download(evecs, dV, nn * count * sizeof(double));
    c->download(status, st, (size_t)count * sizeof(int));
    c->sync();
    if (c->prof_mode == 1) c->prof_collect();
    return 0;
    GP_API_END(c)
}

// diagnostics: the stages of the large-n eigensolver on their own (tests compare them with LAPACK-free identities)
extern "C" int gpcsd_debug_sytrd(gpcsd_ctx *c, const double *A, int n, double *d, double *e, double *V, double *tau) {
    GP_API_BEGIN(c)
    GP_REQUIRE(A && d && e && V && tau && n > 0, -3, "debug_sytrd: bad arguments");
    double *dA = c->upload<double>("op_in0", A, (size_t)n * n);
    double *dd = c->buf<double>("dbg_d", n), *de = c->buf<double>("dbg_e", n), *dt = c->buf<double>("dbg_tau", n);
    double *dV = c->buf<double>("op_out", (size_t)n * n);
    sytrd_device(c, dA, n, dd, de, dV, dt, c->stream);
    c->download(d, dd, n * sizeof(double));
    c->download(e, de, n * sizeof(double));
    c->download(tau, dt, n * sizeof(double));
    c->download(V, dV, (size_t)n * n * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_debug_stedc(gpcsd_ctx *c, const double *d, const double *e, int n, double *w, double *Z) {
    GP_API_BEGIN(c)
    GP_REQUIRE(d && e && w && Z && n > 0, -3, "debug_stedc: bad arguments");
    double *dd = c->upload<double>("dbg_d", d, n);
    double *de = c->buf<double>("dbg_e", n);
    GP_HIP(hipMemsetAsync(de, 0, n * sizeof(double), c->stream));
    if (n > 1) c->copy_in(de, e, (n - 1) * sizeof(double), c->stream);
    double *dw = c->buf<double>("op_w", n);
    double *dZ = c->buf<double>("op_out", (size_t)n * n);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), c->stream));
    stedc_device(c, dd, de, n, dw, dZ, st, c->stream, "dbg");
    c->download(w, dw, n * sizeof(double));
    c->download(Z, dZ, (size_t)n * n * sizeof(double));
    return finish_status(c, st);
    GP_API_END(c)
}

extern "C" int gpcsd_eig_D(gpcsd_ctx *c, const double *Ks, int nx, const double *Kt, int nt, const double *sig2n, int n_sig,
                           double *Qs, double *Qt, double *Dvec) {
    GP_API_BEGIN(c)
    GP_REQUIRE(Ks && Kt && sig2n && nx > 0 && nt > 0 && (n_sig == 1 || n_sig == nx), -3, "eig_D: bad arguments");
    double *dKs = c->upload<double>("Ks", Ks, (size_t)nx * nx);
    double *dKt = c->upload<double>("Kt", Kt, (size_t)nt * nt);
    double *dsig = c->upload<double>("sig2n", sig2n, n_sig);
    double *dQs = c->buf<double>("Qs", (size_t)nx * nx), *dQt = c->buf<double>("Qt", (size_t)nt * nt);
    double *es = c->buf<double>("es", nx), *et = c->buf<double>("et", nt);
    double *D = c->buf<double>("D", (size_t)nx * nt);
    double *scal = c->buf<double>("scalars", 64);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), c->stream));
    eig_pair_D(c, dKs, nx, dKt, nt, dsig, n_sig, dQs, es, dQt, et, D, nullptr, scal, st);
    if (Qs) c->download(Qs, dQs, (size_t)nx * nx * sizeof(double));
    if (Qt) c->download(Qt, dQt, (size_t)nt * nt * sizeof(double));
    if (Dvec) c->download(Dvec, D, (size_t)nx * nt * sizeof(double));
    return finish_status(c, st);
    GP_API_END(c)
}

extern "C" int gpcsd_whitened_quad(gpcsd_ctx *c, const double *Qs, int nx, const double *Qt, int nt, const double *Dvec,
                                   const double *resid, int nb, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(Qs && Qt && Dvec && resid && out && nx > 0 && nt > 0 && nb > 0, -3, "whitened_quad: bad arguments");
    hipStream_t s = c->stream;
    double *dQs = c->upload<double>("wq_Qs", Qs, (size_t)nx * nx);
    double *dQt = c->upload<double>("wq_Qt", Qt, (size_t)nt * nt);
    double *dD = c->upload<double>("wq_D", Dvec, (size_t)nx * nt);
    double *raw = c->upload<double>("wq_raw", resid, (size_t)nx * nt * nb);
    const long BT = (long)nb * nt;
    double *Y = c->buf<double>("wq_Y", (size_t)nx * BT);
    k_swap_last2(c, raw, Y, nx, nt, nb, s);                 // (x, t, b) -> (x, b, t): both projections become flat GEMMs
    double *W = c->buf<double>("wq_W", (size_t)nx * BT), *Al = c->buf<double>("wq_alpha", (size_t)nx * BT);
    GemmDesc g1;                                            // W = Qs^T Y
    g1.M = nx; g1.N = (int)BT; g1.K = nx;
    g1.A = dQs; g1.lda = nx; g1.transA = true; g1.B = Y; g1.ldb = BT; g1.C = W; g1.ldc = BT;
    g1.prof_name = "gemm_wq_spatial";
    gemm_f64(c, g1, s);
    GemmDesc g2;                                            // alpha[(x,b)][i] = sum_t W[(x,b)][t] Qt[t][i]
    g2.M = nx * nb; g2.N = nt; g2.K = nt;
    g2.A = W; g2.lda = nt; g2.B = dQt; g2.ldb = nt; g2.C = Al; g2.ldc = nt;
    g2.prof_name = "gemm_wq_temporal";
    gemm_f64(c, g2, s);
    double *dq = c->buf<double>("wq_out", nb);
    k_per_trial_quad(c, Al, dD, nx, nb, nt, dq, s);
    c->download(out, dq, (size_t)nb * sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_potrf(gpcsd_ctx *c, const double *A, int n, double *L) {
    GP_API_BEGIN(c)
    GP_REQUIRE(A && L && n > 0, -3, "potrf: bad arguments");
    double *dA = c->upload<double>("op_in0", A, (size_t)n * n);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), c->stream));
    potrf_device(c, dA, n, st, c->stream);
    c->download(L, dA, (size_t)n * n * sizeof(double));
    return finish_status(c, st);
    GP_API_END(c)
}

extern "C" int gpcsd_logdet_chol(gpcsd_ctx *c, const double *L, int n, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(L && out && n > 0, -3, "logdet_chol: bad arguments");
    double *dL = c->upload<double>("op_in0", L, (size_t)n * n);
    double *scal = c->buf<double>("scalars", 64);
    logdet_chol_device(c, dL, n, scal, c->stream);
    c->download(out, scal, sizeof(double));
    c->sync();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_trsm_lower(gpcsd_ctx *c, const double *L, int n, const double *B, int nrhs, double *X) {
    GP_API_BEGIN(c)
    GP_REQUIRE(L && B && X && n > 0 && nrhs > 0, -3, "trsm_lower: bad arguments");
    double *dL = c->upload<double>("op_in0", L, (size_t)n * n);
    double *dB = c->upload<double>("op_in1", B, (size_t)n * nrhs);
    trsm_lower_device(c, dL, n, dB, nrhs, c->stream);
    c->download(X, dB, (size_t)n * nrhs * sizeof(double));
    c->sync();
    if (c->prof_mode == 1) c->prof_collect();
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_gemm(gpcsd_ctx *c, int transA, int transB, int M, int N, int K, const double *A, const double *B,
                          double *C) {
    GP_API_BEGIN(c)
    GP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, -3, "gemm: bad arguments");
    double *dA = c->upload<double>("op_in0", A, (size_t)M * K);
    double *dB = c->upload<double>("op_in1", B, (size_t)K * N);
    double *dC = c->buf<double>("op_out", (size_t)M * N);
    GemmDesc g;
    g.M = M; g.N = N; g.K = K;
    g.A = dA; g.transA = transA != 0; g.lda = transA ? M : K;
    g.B = dB; g.transB = transB != 0; g.ldb = transB ? K : N;
    g.C = dC; g.ldc = N;
    gemm_f64(c, g, c->stream);
    c->download(C, dC, (size_t)M * N * sizeof(double));
    c->sync();
    if (c->prof_mode == 1) c->prof_collect();
    return 0;
    GP_API_END(c)
}

__global__ void fill_pattern_kernel(double *p, long n, double a) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 6364136223846793005ull + 1442695040888963407ull;
        h ^= h >> 29;
        p[i] = a * ((double)(h & 0xFFFFFF) / 8388608.0 - 1.0);          // pseudo-random in [-a, a)
    }
}

// Time the fp64 MFMA GEMM on device-resident pseudo-random operands: average ms per launch over `reps` launches.
// cfg = 0 picks the tile configuration automatically, 1..6 forces one (tuning aid; see gemm_f64.hip).
extern "C" int gpcsd_gemm_bench(gpcsd_ctx *c, int transA, int transB, int M, int N, int K, int cfg, int reps, double *ms_out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(M > 0 && N > 0 && K > 0 && reps > 0 && ms_out, -3, "gemm_bench: bad arguments");
    double *dA = c->buf<double>("bench_A", (size_t)M * K);
    double *dB = c->buf<double>("bench_B", (size_t)K * N);
    double *dC = c->buf<double>("bench_C", (size_t)M * N);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, c->stream, dA, (long)M * K, 1.0);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, c->stream, dB, (long)K * N, 0.5);
    GemmDesc g;
    g.M = M; g.N = N; g.K = K;
    g.A = dA; g.transA = transA != 0; g.lda = transA ? M : K;
    g.B = dB; g.transB = transB != 0; g.ldb = transB ? K : N;
    g.C = dC; g.ldc = N;
    g.cfg = cfg;
    g.prof_name = "gemm_bench";
    gemm_f64(c, g, c->stream);                      // warm-up
    hipEvent_t e0 = c->get_event(), e1 = c->get_event();
    GP_HIP(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) gemm_f64(c, g, c->stream);
    GP_HIP(hipEventRecord(e1, c->stream));
    GP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GP_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / reps;
    c->event_pool.push_back(e0);
    c->event_pool.push_back(e1);
    return 0;
    GP_API_END(c)
}

// K[(x,i),(x',i')] = Ks[x,x'] Kt[i,i'] + sig2n delta
__global__ void kron_plus_diag_kernel(const double *__restrict__ Ks, int nx, const double *__restrict__ Kt, int nt, double sig2n,
                                      double *__restrict__ K) {
    const long N = (long)nx * nt;
    const long row = blockIdx.y;
    const int x = (int)(row / nt), i = (int)(row % nt);
    for (long col = blockIdx.x * (long)blockDim.x + threadIdx.x; col < N; col += (long)gridDim.x * blockDim.x) {
        const int xp = (int)(col / nt), ip = (int)(col % nt);
        double v = Ks[(long)x * nx + xp] * Kt[(long)i * nt + ip];
        if (col == row) v += sig2n;
        K[row * N + col] = v;
    }
}

extern "C" int gpcsd_loglik_dense_chol(gpcsd_ctx *c, const double *Ks, int nx, const double *Kt, int nt, double sig2n,
                                       const double *lfp, int ntrials, double *out) {
    GP_API_BEGIN(c)
    GP_REQUIRE(Ks && Kt && lfp && out && nx > 0 && nt > 0 && ntrials > 0, -3, "loglik_dense_chol: bad arguments");
    const long N = (long)nx * nt;
    GP_REQUIRE(N <= 16384, -3, "loglik_dense_chol: N = nx*nt = %ld too large for the dense cross-check (max 16384)", N);
    GP_REQUIRE(N <= 65535, -3, "grid limit");
    hipStream_t s = c->stream;
    double *dKs = c->upload<double>("Ks", Ks, (size_t)nx * nx);
    double *dKt = c->upload<double>("Kt", Kt, (size_t)nt * nt);
    double *K = c->buf<double>("dense_K", (size_t)N * N);
    double *y = c->upload<double>("dense_y", lfp, (size_t)N * ntrials);   // (nx,nt,R) C-order == (N, R)
    double *scal = c->buf<double>("scalars", 64);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(kron_plus_diag_kernel, dim3(ceil_div(N, 256) > 64 ? 64 : ceil_div(N, 256), (unsigned)N), dim3(256), 0, s,
                       (const double *)dKs, nx, (const double *)dKt, nt, sig2n, K);
    potrf_device(c, K, (int)N, st, s);
    logdet_chol_device(c, K, (int)N, scal, s);
    trsm_lower_device(c, K, (int)N, y, ntrials, s);
    sumsq_device(c, y, N * ntrials, scal + 1, s);
    double h[2];
    c->download(h, scal, sizeof(h));
    int rc = finish_status(c, st);
    *out = -0.5 * ntrials * h[0] - 0.5 * h[1];
    return rc;
    GP_API_END(c)
}

extern "C" int gpcsd_set_host_temporal_gram(gpcsd_ctx *c, const double *Kt, int nt, const double *Kt_cross, int ncomp,
                                            int ntstar) {
    GP_API_BEGIN(c)
    ++c->grid_epoch;                            // a new host Gram is a new temporal problem
    if (!Kt) {                                  // back to the built-in SE / Matern builders
        c->host_kt_on = false;
        c->host_kt.clear();
        c->host_kt_cross.clear();
        c->host_kt_nt = c->host_kt_C = c->host_kt_ntstar = 0;
        c->host_dkt.clear();
        c->host_dkt_n = 0;
        return 0;
    }
    c->host_dkt.clear();                        // derivatives belong to the Gram they were handed over with
    c->host_dkt_n = 0;
    GP_REQUIRE(nt > 0, -3, "set_host_temporal_gram: nt must be positive");
    GP_REQUIRE(!Kt_cross || (ncomp >= 1 && ncomp <= GPCSD_MAX_TEMPORAL && ntstar > 0), -3,
               "set_host_temporal_gram: bad cross-Gram shape (%d, %d, %d)", ncomp, ntstar, nt);
    c->host_kt.assign(Kt, Kt + (size_t)nt * nt);
    c->host_kt_nt = nt;
    if (Kt_cross) {
        c->host_kt_cross.assign(Kt_cross, Kt_cross + (size_t)ncomp * ntstar * nt);
        c->host_kt_C = ncomp;
        c->host_kt_ntstar = ntstar;
    } else {
        c->host_kt_cross.clear();
        c->host_kt_C = c->host_kt_ntstar = 0;
    }
    c->host_kt_on = true;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_set_host_temporal_dgram(gpcsd_ctx *c, const double *dKt, int nt, int nmat) {
    GP_API_BEGIN(c)
    if (!dKt) {
        c->host_dkt.clear();
        c->host_dkt_n = 0;
        return 0;
    }
    GP_REQUIRE(c->host_kt_on && nt == c->host_kt_nt, -3,
               "set_host_temporal_dgram: hand the Gram matrix over first (gpcsd_set_host_temporal_gram) -- nt=%d, Gram nt=%d", nt,
               c->host_kt_nt);
    GP_REQUIRE(nmat >= 1 && nmat <= 2 * GPCSD_MAX_TEMPORAL, -3, "set_host_temporal_dgram: %d derivative matrices (1..%d)", nmat,
               2 * GPCSD_MAX_TEMPORAL);
    c->host_dkt.assign(dKt, dKt + (size_t)nmat * nt * nt);
    c->host_dkt_n = nmat;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_set_gram_precision(gpcsd_ctx *c, int bits) {
    GP_API_BEGIN(c)
    GP_REQUIRE(bits == 32 || bits == 64, -3, "gram precision must be 32 or 64 bits (got %d)", bits);
    if (c->gram_fp32 != (bits == 32)) ++c->grid_epoch;
    c->gram_fp32 = bits == 32;
    return 0;
    GP_API_END(c)
}

// ---- multi-GPU without Python (one process per GPU, any launcher): trials are independent, so a rank needs nothing but its
// block of trials and a sum of one double per evaluation.  Contiguous blocks, the first (ntrials mod world) ranks get one
// extra trial -- the partition of gpcsd_amd.dist.TrialSharding.block.
extern "C" int gpcsd_shard_block(int ntrials, int rank, int world, int *first, int *count) {
    if (!first || !count || ntrials < 0 || world < 1 || rank < 0 || rank >= world) return -3;
    const int base = ntrials / world, extra = ntrials % world;
    *first = rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return 0;
}

// loglik of ALL trials from the pieces gpcsd_loglik_parts returns on each rank: sum log D (identical on every rank: the
// decompositions are deterministic replicas) and the sum over ranks of the partial quadratic terms.   gpcsd1d.py:122,127-128
extern "C" int gpcsd_combine_loglik(int ntrials_total, double sumlog, double quad_sum_over_ranks, double *out) {
    if (!out) return -3;
    *out = -0.5 * (double)ntrials_total * sumlog - 0.5 * quad_sum_over_ranks;
    return 0;
}

extern "C" int gpcsd_decomposition_cache(gpcsd_ctx *c, int on, long *hits) {
    GP_API_BEGIN(c)
    if (on >= 0) {
        c->decomp_cache_on = on != 0;
        c->decomp_gen[0] = c->decomp_gen[1] = -1;
    }
    if (hits) *hits = c->decomp_cache_hits;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_ll_tridiag(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    GP_REQUIRE(on <= 2, -3, "ll_tridiag: mode must be 0 (off), 1 (on), 2 (by size) or negative (query)");
    if (on >= 0 && c->ll_tridiag_mode != on) {
        if (int rc = drain_async(c)) return rc;            // the two forms order their streams differently: start from an idle context
        GP_HIP(hipStreamSynchronize(c->stream2));
        GP_HIP(hipStreamSynchronize(c->stream3));
        GP_HIP(hipStreamSynchronize(c->stream4));
        c->ll_tridiag_mode = on;
        c->decomp_gen[0] = c->decomp_gen[1] = -1;          // (the forms compute the temporal eigenvectors differently: no reuse across)
        c->q_gen = -1;
    }
    if (calls) *calls = c->ll_tridiag_calls;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_tail_early_exit(gpcsd_ctx *c, int on, int *previous) {
    GP_API_BEGIN(c)
    if (previous) *previous = c->tail_early_exit ? 1 : 0;
    if (on >= 0 && c->tail_early_exit != (on != 0)) {
        if (int rc = drain_async(c)) return rc;            // start from an idle context
        GP_HIP(hipStreamSynchronize(c->stream2));
        GP_HIP(hipStreamSynchronize(c->stream3));
        GP_HIP(hipStreamSynchronize(c->stream4));
        c->tail_early_exit = on != 0;
        ++c->alloc_epoch;                                  // captured chains carry the flag in their kernel arguments: retire them
        c->decomp_gen[0] = c->decomp_gen[1] = -1;          // (and nothing decomposed under the other setting is reused)
        c->q_gen = -1;
    }
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_debug_fault_stage2(gpcsd_ctx *c, int on) {
    GP_API_BEGIN(c)
    if (int rc = drain_async(c)) return rc;
    GP_HIP(hipStreamSynchronize(c->stream2));
    GP_HIP(hipStreamSynchronize(c->stream4));
    c->fault_stage2 = on != 0;
    ++c->alloc_epoch;                                      // (a captured stage 2 holds or lacks the injected launch)
    c->decomp_gen[0] = c->decomp_gen[1] = -1;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_pair_share_x(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    if (on >= 0) c->pair_share_x = on != 0;
    if (calls) *calls = c->pair_shared_x_calls;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_predict_chunked_copy(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    if (on >= 0) c->pred_chunked = on != 0;
    if (calls) *calls = c->pred_chunked_calls;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_pair_share_s(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    if (on >= 0) c->pair_share_s = on != 0;
    if (calls) *calls = c->pair_shared_s_calls;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_q_pipeline(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    if (on >= 0) c->q_pipe = on != 0;
    if (calls) *calls = c->q_pipe_calls;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_bounce_stats(gpcsd_ctx *c, long *bytes) {
    GP_API_BEGIN(c)
    if (bytes) *bytes = c->bounced_bytes;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_q_pipeline_stats(gpcsd_ctx *c, int *on, long *timeouts, long long gate_ticks) {
    GP_API_BEGIN(c)
    if (gate_ticks >= 0) c->q_gate_ticks = (unsigned long long)gate_ticks;
    if (on) *on = c->q_pipe ? 1 : 0;
    if (timeouts) *timeouts = c->q_pipe_timeouts;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_fold_gemm(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    if (on >= 0) c->fold_gemm_on = on != 0;
    if (calls) *calls = c->fold_gemm_calls;
    return 0;
    GP_API_END(c)
}
