// extern "C" surface of libgpcsd_hip.so (include/gpcsd_hip.h) and the host-side orchestration of the hot path.
// This file: error handling, the internal pipelines shared by the call families (Gram assembly, front half, folded-basis
// helpers), context and resident data.  The call families themselves are the capi_*.inl parts included at the end.
#include <functional>
#include <cmath>
#include <mutex>

#include "kernels.hpp"
#include "jacobi.hpp"

using namespace gpcsd;

static std::string g_last_error;
static std::mutex g_err_mutex;

static int fail(gpcsd_ctx *c, const HipError &e) {
    if (c) c->last_error = e.msg;
    std::lock_guard<std::mutex> lk(g_err_mutex);
    g_last_error = e.msg;
    return e.code;
}

#define GP_API_BEGIN(ctx)                                                                      \
    if (!(ctx)) return fail(nullptr, HipError{-1, "null context"});                            \
    try {                                                                                      \
        GP_HIP(hipSetDevice((ctx)->device));
#define GP_API_END(ctx)                                                                        \
    }                                                                                          \
    catch (const HipError &e) { drain_after_failure(ctx); return fail((ctx), e); }             \
    catch (const std::exception &e) { drain_after_failure(ctx); return fail((ctx), HipError{-99, e.what()}); }

// A call that throws after queueing work must not return while kernels or asynchronous copies that read the caller's
// buffers are still in flight, and must not leave stale work on stream2 for the next call to race with (best effort).
static void pair_prefetch_drop(gpcsd_ctx *c);
static void drain_after_failure(gpcsd_ctx *c) {
    if (!c) return;
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->stream4) (void)hipStreamSynchronize(c->stream4);
    if (c->stream5) (void)hipStreamSynchronize(c->stream5);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    pair_prefetch_drop(c);
    c->status_zeroed = false;
    c->async_pending = false;                   // everything is drained: no deferred status
    // (an outstanding asynchronous loglik stays collectable: its result has landed by now)
    c->decomp_gen[0] = c->decomp_gen[1] = -1;   // whatever the failed call left behind is not reused
}

static int drain_async(gpcsd_ctx *c);

// Start the next generation of side's decomposition (0 spatial, 1 temporal) on `chain`: see gpcsd_ctx::par.  Flips the
// current generation -- callers fetch output buffers and fold views AFTER this -- and orders the chain behind every reader
// of the generation it is about to overwrite.  after_main_now: the chain also needs something queued on the main stream
// just now (an upload of this call, cleared status words): wait for the main stream's current position instead.
static void begin_generation(gpcsd_ctx *c, int side, hipStream_t chain, bool after_main_now) {
    pair_prefetch_drop(c);                 // a prefetched front half is overtaken by this one (its outputs are about to be rewritten)
    const int p = (c->par[side] ^= 1);
    GP_HIP(hipEventRecord(c->ev_mark[side][p], c->stream));
    if (chain != c->stream) GP_HIP(hipStreamWaitEvent(chain, c->ev_mark[side][after_main_now ? p : (p ^ 1)], 0));
}

// name of a per-generation output buffer of `side`
static std::string gen_name(const gpcsd_ctx *c, int side, const char *base) { return std::string(base) + (c->par[side] ? "#1" : "#0"); }

void gpcsd_ctx::timeline_dump() {
    if (timeline.empty()) return;
    // keep the last ~4 calls' worth of marks
    const size_t keep = 64, n0 = timeline.size() > keep ? timeline.size() - keep : 0;
    for (size_t i = 0; i < timeline.size(); ++i) (void)hipEventSynchronize(timeline[i].second);
    for (size_t i = n0; i < timeline.size(); ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, timeline[n0].second, timeline[i].second);
        fprintf(stderr, "[timeline] %9.1f us  %s\n", 1e3 * ms, timeline[i].first.c_str());
    }
    for (auto &kv : timeline) event_pool.push_back(kv.second);
    timeline.clear();
}

void gpcsd_ctx::prof_collect() {
    for (auto &kv : prof) {
        ProfEntry &p = kv.second;
        for (size_t i = 0; i < p.pending.size(); ++i) {
            float ms = 0.f;
            if (hipEventSynchronize(p.pending[i].second) == hipSuccess &&
                hipEventElapsedTime(&ms, p.pending[i].first, p.pending[i].second) == hipSuccess) {
                p.ms += ms;
                p.count += 1;
                p.flops += p.pending_flops[i];
            }
            event_pool.push_back(p.pending[i].first);
            event_pool.push_back(p.pending[i].second);
        }
        p.pending.clear();
        p.pending_flops.clear();
    }
}

// ------------------------------------------------------------------------------------------------
// internal pipelines (device pointers)
// ------------------------------------------------------------------------------------------------
namespace {

struct Geo {     // device-side geometry description (either resident in ctx or uploaded per operator call)
    int dim = 0;
    const double *x = nullptr;   // (nx) or (nx,2)
    int nx = 0;
    const double *gx1 = nullptr, *gw1 = nullptr, *gx2 = nullptr, *gw2 = nullptr;
    int ngl1 = 0, ngl2 = 0;
    int G() const { return dim == 1 ? ngl1 : ngl1 * ngl2; }
};

Geo resident_geo(gpcsd_ctx *c) {
    GP_REQUIRE(c->dim == 1 || c->dim == 2, -4, "geometry not set (call gpcsd_set_geometry_1d/2d)");
    Geo g;
    g.dim = c->dim;
    g.nx = c->geo_nx;
    g.x = (const double *)c->bufs["geo_x"].p;
    g.gx1 = (const double *)c->bufs["geo_gx1"].p;
    g.gw1 = (const double *)c->bufs["geo_gw1"].p;
    g.ngl1 = c->ngl1;
    if (c->dim == 2) {
        g.gx2 = (const double *)c->bufs["geo_gx2"].p;
        g.gw2 = (const double *)c->bufs["geo_gw2"].p;
        g.ngl2 = c->ngl2;
    }
    return g;
}

// A(n, G) forward weights for the point set pts (n points)
void fwd_weights(gpcsd_ctx *c, const Geo &g, const double *pts, int n, double R, double eps, double *A, hipStream_t s) {
    if (g.dim == 1) k_fwd_weights_1d(c, pts, n, g.gx1, g.gw1, g.ngl1, R, A, s);
    else k_fwd_weights_2d(c, pts, n, g.gx1, g.gw1, g.ngl1, g.gx2, g.gw2, g.ngl2, R, eps, A, s);
}

// Kphi(nx, nxp) = A Kgl Axp^T (+ jitter I when square and jitter != 0)     covariances.py:74-96 / :204-232
// pfx names the scratch buffers: the spatial chain ("ks_", on its own stream) and the cross-covariance builds of predict
// ("kx_", main stream) run concurrently and must not share them.
void build_kphi(gpcsd_ctx *c, const Geo &g, double R, double eps, const double *ell, const double *xp, int nxp, double jitter,
                double *out, hipStream_t s, const char *pfx = "kx_") {
    const int G = g.G();
    const std::string P(pfx);
    double *A = c->buf<double>(P + "A", (size_t)g.nx * G);
    double *T = c->buf<double>(P + "T", (size_t)g.nx * G);
    fwd_weights(c, g, g.x, g.nx, R, eps, A, s);
    if (g.dim == 1) {
        double *Kgl = c->buf<double>(P + "Kgl", (size_t)G * G);
        k_se_1d(c, g.gx1, G, g.gx1, G, ell[0], Kgl, s);
        GemmDesc d1;                               // T = A Kgl
        d1.M = g.nx; d1.N = G; d1.K = G;
        d1.A = A; d1.lda = G; d1.B = Kgl; d1.ldb = G; d1.C = T; d1.ldc = G;
        d1.prof_name = "gemm_Ks_AKgl";
        gemm_f64(c, d1, s);
    } else {
        // On the GL tensor grid Kgl = K1 (x) K2 (covariances.py:216: a product of one factor per axis), so
        //   T[x][(h1,h2)] = sum_{g1} K1[g1][h1] ( sum_{g2} A[x][(g1,g2)] K2[g2][h2] ):
        // two small products (2 nx G (ngl1 + ngl2) flops: 74 MF at 384 x 20 x 60) instead of the 2 nx G^2 = 1.1 GF flat one,
        // and Kgl (1200^2 exponentials) is never formed.  Same sums re-associated: agrees with the flat product to rounding.
        const int n1 = g.ngl1, n2 = g.ngl2;
        double *K1 = c->buf<double>(P + "K1", (size_t)n1 * n1), *K2 = c->buf<double>(P + "K2", (size_t)n2 * n2);
        double *U = c->buf<double>(P + "U", (size_t)g.nx * G);
        k_se_axis(c, g.gx1, n1, ell[0], K1, s);
        k_se_axis(c, g.gx2, n2, ell[1], K2, s);
        // GPCSD_KS_CFG=ab: tile configurations of the two small products (a: A K2, b: K1^T U; 0 = automatic, i.e. the 32 x 32 / BK 64
        // tile of latency-bound launches -- 64 KB of LDS per workgroup, which beside another stream's flood of 64 x 64 tiles only
        // finds room where two of those have retired on one CU: 72 us in the step loop for 12 alone).  A/B.
        static const char *ks_cfg = getenv("GPCSD_KS_CFG");
        GemmDesc u;                                // U[(x,g1)][h2] = sum_g2 A[(x,g1)][g2] K2[g2][h2]
        u.M = g.nx * n1; u.N = n2; u.K = n2;
        u.A = A; u.lda = n2; u.B = K2; u.ldb = n2; u.C = U; u.ldc = n2;
        if (ks_cfg && ks_cfg[0] > '0') u.cfg = ks_cfg[0] - '0';
        u.prof_name = "gemm_Ks_AK2";
        gemm_f64(c, u, s);
        GemmDesc v;                                // T_x (n1 x n2) = K1^T U_x, one small product per electrode
        v.M = n1; v.N = n2; v.K = n1;
        v.A = K1; v.lda = n1; v.transA = true; v.B = U; v.ldb = n2; v.C = T; v.ldc = n2;
        v.batch = g.nx; v.sA = 0; v.sB = G; v.sC = G;
        if (ks_cfg && ks_cfg[0] && ks_cfg[1] > '0') v.cfg = ks_cfg[1] - '0';
        v.prof_name = "gemm_Ks_K1U";
        gemm_f64(c, v, s);
    }
    const double *Axp = A;
    int n2 = g.nx;
    if (xp) {
        double *A2 = c->buf<double>(P + "Axp", (size_t)nxp * G);
        fwd_weights(c, g, xp, nxp, R, eps, A2, s);
        Axp = A2;
        n2 = nxp;
    }
    GemmDesc d2;                                   // out = T Axp^T
    d2.M = g.nx; d2.N = n2; d2.K = G;
    d2.A = T; d2.lda = G; d2.B = Axp; d2.ldb = G; d2.transB = true; d2.C = out; d2.ldc = n2;
    d2.prof_name = "gemm_Ks_TAt";
    // Few output tiles over a long contraction (384 x 384 over the 1200 quadrature nodes: 36 workgroups walking 150 K tiles each,
    // 67 us at the head of the spatial chain): the K range in `parts` batches of one launch into partial products, then one
    // fixed-order sum.
    int parts = 1;
    static const bool splitk = !(getenv("GPCSD_KS_SPLITK") && getenv("GPCSD_KS_SPLITK")[0] == '0');
    if (splitk && (long)ceil_div(g.nx, 64) * ceil_div(n2, 64) <= 64 && G >= 512)
        for (int q : {8, 6, 5, 4, 3, 2})
            if (G % q == 0 && G / q >= 128) { parts = q; break; }
    if (parts > 1) {
        const long nn = (long)g.nx * n2;
        double *Pp = c->buf<double>(P + "TAt_parts", (size_t)nn * parts);
        d2.K = G / parts;
        d2.batch = parts; d2.sA = d2.K; d2.sB = d2.K; d2.sC = nn;
        d2.C = Pp;
        gemm_f64(c, d2, s);
        k_sum_partials(c, out, Pp, nn, parts, s);
    } else {
        gemm_f64(c, d2, s);
    }
    if (jitter != 0.0 && n2 == g.nx) k_add_diag(c, out, g.nx, jitter, s);
}

// Kphig(nx, nz) = A Kcross, Kcross[g, z] = SE(gl_g, z)                     covariances.py:58-72 / :188-202
void build_kphig(gpcsd_ctx *c, const Geo &g, double R, double eps, const double *ell, const double *z, int nz, double *out,
                 hipStream_t s) {
    const int G = g.G();
    double *A = c->buf<double>("kx_A", (size_t)g.nx * G);
    double *Kc = c->buf<double>("kx_Kcross", (size_t)G * nz);
    fwd_weights(c, g, g.x, g.nx, R, eps, A, s);
    if (g.dim == 1) k_se_1d(c, g.gx1, G, z, nz, ell[0], Kc, s);
    else k_se_2d(c, g.gx1, g.gx2, G, g.ngl2, z, nullptr, nz, 0, ell[0], ell[1], Kc, s);
    GemmDesc d;
    d.M = g.nx; d.N = nz; d.K = G;
    d.A = A; d.lda = G; d.B = Kc; d.ldb = nz; d.C = out; d.ldc = nz;
    d.prof_name = "gemm_Kphig";
    gemm_f64(c, d, s);
}

void build_ks_csd(gpcsd_ctx *c, const Geo &g, const double *ell, double *out, hipStream_t s) {
    if (g.dim == 1) k_se_1d(c, g.x, g.nx, g.x, g.nx, ell[0], out, s);
    else k_se_2d(c, g.x, nullptr, g.nx, 0, g.x, nullptr, g.nx, 0, ell[0], ell[1], out, s);
}

void build_kt(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *t, int n, const double *tp, int m, double *out,
              hipStream_t s) {
    k_temporal_gram(c, hp->n_temporal, hp->kind, hp->ell_t, hp->sigma2_t, t, n, tp, m, out, s);
}

// The temporal chain's input in one launch (k_temporal_fold_fill): applies when the time grid's reflection symmetry folds the
// eigenproblem and the library's own Gram builders evaluate the kernels.  GPCSD_TFILL=0: the separate Gram -> fold ->
// absmax -> scale launches (A/B).
static bool temporal_fill_applies(gpcsd_ctx *c, const SymDev *sym_t, int nt, bool host_kt) {
    static const bool off = getenv("GPCSD_TFILL") && getenv("GPCSD_TFILL")[0] == '0';
    return !off && sym_t && !host_kt && eigh_fold_view(c, 1, sym_t, nt).on;
}
static void temporal_fill(gpcsd_ctx *c, const gpcsd_hparams *const *hps, int nrep, const double *t, int nt, const SymDev &sy,
                          int *status, int status_stride, hipStream_t s) {
    TemporalSet sets[2];
    for (int r = 0; r < nrep; ++r) {
        sets[r].ncomp = hps[r]->n_temporal;
        for (int i = 0; i < GPCSD_MAX_TEMPORAL; ++i) {
            const bool on = i < hps[r]->n_temporal;
            sets[r].kind[i] = on ? hps[r]->kind[i] : 0;
            sets[r].ell[i] = on ? hps[r]->ell_t[i] : 1.0;
            sets[r].sigma2[i] = on ? hps[r]->sigma2_t[i] : 0.0;
        }
    }
    const char *const *tg = eigh_fold_tags(c, 1);
    const EigArenaView as = eigh_arena_view(c, tg[0], sy.ns, nrep), aa = eigh_arena_view(c, tg[1], sy.na, nrep);
    k_temporal_fold_fill(c, sets, nrep, t, nt, sy, as, aa, status, status_stride, s);
}

// The spatial chain's input the same way (k_psd_fold_fill): Ks is assembled WITHOUT the jitter, which the fill adds to the
// folded blocks' diagonals -- the paired call's two replicas (Ks + jitter I, Ks) then need no copy of Ks.  GPCSD_SFILL=0: the
// separate add_diag -> fold -> absmax -> scale launches (A/B).
static bool spatial_fill_applies(gpcsd_ctx *c, const SymDev *sym_s, int nx) {
    static const bool off = getenv("GPCSD_SFILL") && getenv("GPCSD_SFILL")[0] == '0';
    return !off && sym_s && eigh_fold_view(c, 0, sym_s, nx).on;
}
static void spatial_fill(gpcsd_ctx *c, const double *Ks, int nx, long sK, int nrep, const double *jitter, const SymDev &sy,
                         int *status, int status_stride, hipStream_t s) {
    const char *const *tg = eigh_fold_tags(c, 0);
    const EigArenaView as = eigh_arena_view(c, tg[0], sy.ns, nrep), aa = eigh_arena_view(c, tg[1], sy.na, nrep);
    k_psd_fold_fill(c, Ks, nx, sK, nrep, jitter, sy, as, aa, status, status_stride, s);
}

bool uses_host_kt(const gpcsd_hparams *hp);

// Kt*_c = cov_c.compute_Kt(tstar) (ntstar, nt) of component cc (gpcsd1d.py:277): built on the device for SE / Matern, copied
// from the caller's matrices when the temporal covariances are user-defined
void temporal_cross_gram(gpcsd_ctx *c, const gpcsd_hparams *hp, int cc, const double *dts, int ntstar, const double *t, int nt,
                         double *out, hipStream_t s) {
    if (uses_host_kt(hp)) {
        GP_REQUIRE(c->host_kt_C == hp->n_temporal && c->host_kt_ntstar == ntstar && c->host_kt_nt == nt &&
                       c->host_kt_cross.size() == (size_t)hp->n_temporal * ntstar * nt, -3,
                   "predict with user-defined temporal covariances needs the per-component cross Grams "
                   "(gpcsd_set_host_temporal_gram: Kt_cross of shape (%d, %d, %d))", hp->n_temporal, ntstar, nt);
        c->copy_in(out, c->host_kt_cross.data() + (size_t)cc * ntstar * nt, (size_t)ntstar * nt * sizeof(double), s);
        return;
    }
    k_temporal_gram(c, 1, &hp->kind[cc], &hp->ell_t[cc], &hp->sigma2_t[cc], dts, ntstar, t, nt, out, s);
}

void check_hp(gpcsd_ctx *c, const gpcsd_hparams *hp, int nx) {
    GP_REQUIRE(hp != nullptr, -3, "null hparams");
    GP_REQUIRE(hp->n_temporal >= 1 && hp->n_temporal <= GPCSD_MAX_TEMPORAL, -3, "n_temporal=%d outside [1,%d]",
               hp->n_temporal, GPCSD_MAX_TEMPORAL);
    GP_REQUIRE(hp->sig2n != nullptr && (hp->n_sig2n == 1 || hp->n_sig2n == nx), -3,
               "sig2n must have 1 or nx=%d entries (got %d)", nx, hp->n_sig2n);
    for (int i = 0; i < hp->n_temporal; ++i) {
        const int k = hp->kind[i];
        GP_REQUIRE(k == GPCSD_KIND_SE || k == GPCSD_KIND_MATERN || k == GPCSD_KIND_HOST, -3, "unknown temporal kernel kind %d", k);
        GP_REQUIRE(k != GPCSD_KIND_HOST || c->host_kt_on, -3,
                   "temporal component %d is GPCSD_KIND_HOST but no Gram matrix was supplied (gpcsd_set_host_temporal_gram)", i);
    }
}

// true when the temporal Gram matrices of this call come from the caller (any component of kind HOST)
bool uses_host_kt(const gpcsd_hparams *hp) {
    if (!hp) return false;
    for (int i = 0; i < hp->n_temporal; ++i)
        if (hp->kind[i] == GPCSD_KIND_HOST) return true;
    return false;
}

// ---- reflection symmetry of a point set (host): find the involution i -> P(i) with pts[P(i)] = 2*centre - pts[i]
// along the dimensions flagged in `reflect`, and upload its orbit tables.  Returns an empty SymDev if there is none.
SymDev find_symmetry(gpcsd_ctx *c, const std::string &name, const double *pts, int n, int dim, const double *centre,
                     const bool *reflect) {
    SymDev out;
    if (n < 2) return out;
    double scale = 0.0;
    for (int i = 0; i < n * dim; ++i) scale = std::max(scale, std::fabs(pts[i] - centre[i % dim]));
    const double tol = 1e-9 * std::max(scale, 1e-300);
    std::vector<int> perm(n, -1);
    for (int i = 0; i < n; ++i) {
        double tgt[2];
        for (int d = 0; d < dim; ++d) tgt[d] = reflect[d] ? 2.0 * centre[d] - pts[i * dim + d] : pts[i * dim + d];
        int best = -1;
        double bd = 1e300;
        for (int j = 0; j < n; ++j) {
            double dist = 0.0;
            for (int d = 0; d < dim; ++d) dist = std::max(dist, std::fabs(pts[j * dim + d] - tgt[d]));
            if (dist < bd) {
                bd = dist;
                best = j;
            }
        }
        if (best < 0 || bd > tol) return out;
        perm[i] = best;
    }
    int npairs = 0;
    for (int i = 0; i < n; ++i) {
        if (perm[perm[i]] != i) return out;
        if (perm[i] > i) ++npairs;
    }
    if (npairs == 0) return out;
    const int ns = n - npairs, na = npairs;
    std::vector<int> tbl(2 * ns + 2 * n);
    int *rep_i = tbl.data(), *rep_j = rep_i + ns, *orb = rep_j + ns, *sgn = orb + n;
    int a = 0;
    for (int i = 0; i < n; ++i)
        if (perm[i] > i) {
            rep_i[a] = i; rep_j[a] = perm[i];
            orb[i] = a; sgn[i] = 1;
            orb[perm[i]] = a; sgn[perm[i]] = -1;
            ++a;
        }
    for (int i = 0; i < n; ++i)
        if (perm[i] == i) {
            rep_i[a] = i; rep_j[a] = i;
            orb[i] = a; sgn[i] = 0;
            ++a;
        }
    int *d = c->buf<int>(name, tbl.size());
    c->copy_in(d, tbl.data(), tbl.size() * sizeof(int), c->stream);
    GP_HIP(hipStreamSynchronize(c->stream));
    out.ns = ns; out.na = na;
    out.rep_i = d; out.rep_j = d + ns; out.orb = d + 2 * ns; out.sgn = d + 2 * ns + n;
    c->sym_host[d] = std::vector<int>(tbl.begin(), tbl.begin() + 2 * ns);      // (rep_i | rep_j: the chunked copy of gpcsd_predict)
    return out;
}

bool rule_is_symmetric(const double *gx, const double *gw, int n, double *centre) {
    *centre = 0.5 * (gx[0] + gx[n - 1]);
    const double scale = std::max(std::fabs(gx[n - 1] - gx[0]), 1e-300);
    for (int k = 0; k < n; ++k) {
        if (std::fabs(gx[k] + gx[n - 1 - k] - 2.0 * *centre) > 1e-11 * scale) return false;
        if (std::fabs(gw[k] - gw[n - 1 - k]) > 1e-11 * std::fabs(gw[k])) return false;
    }
    return true;
}

// Eigen-decompose Ks (stream) and Kt (stream2) concurrently; D and sum(log D).
// Inputs Ks, Kt are destroyed.  Outputs: Qs, es, Qt, et, D, sumlog (device).
void eig_pair_D(gpcsd_ctx *c, double *Ks, int nx, double *Kt, int nt, const double *d_sig, int nsig, double *Qs, double *es,
                double *Qt, double *et, double *D, double *Dinv, double *d_sumlog, int *d_status, const SymDev *sym_s = nullptr,
                const SymDev *sym_t = nullptr, bool need_merged = true) {
    {
        // all problems share every launch of the per-column tridiagonalisation (batched), so one stream suffices
        ProfScope ps(c, "eigh_pair", 9.0 * ((double)nx * nx * nx + (double)nt * nt * nt), c->stream);
        eigh_pair_device(c, Ks, nx, es, Qs, sym_s, Kt, nt, et, Qt, sym_t, d_status, c->stream, need_merged);
    }
    k_build_D(c, es, nx, et, nt, d_sig, nsig, D, Dinv, d_sumlog, c->stream);
}

struct EigState {
    double *Qs, *Qt, *es, *et, *D, *Dinv, *scal;   // scal[0] = sumlog, scal[1] = quad, ...; Dinv = 1/D for the GEMM epilogues
    int *status;
    // two-stream front half: the temporal chain (Kt, its eigen-decomposition) runs on stream2 and has not been waited
    // for yet; join_temporal() makes Qt / et / D available on the main stream
    bool pending = false;          // D has not been formed yet (two-stream front half: join_temporal does it)
    bool wait_temporal = false;    // the temporal chain was queued on stream2 in this call: the main stream must wait for it
    bool wait_spatial = false;     // the spatial chain was queued on stream3 in this call: join_spatial() before using Qs / es
    const double *d_sig = nullptr;
    int nsig = 0;
    // the log-likelihood may take the shifted-tridiagonal tail (loglik_tri_*): the temporal chain ran in stages (or its
    // stage-1 outputs are still those of this temporal problem), replica `tri_rep` of the temporal classes is this call's
    bool tri = false, wait_q = false;
    int tri_rep = 0, tri_count = 1;
    // tri: stage 1 ran with progress words and NO stage 3 was queued -- loglik_tri_pre queues stage 5 instead (queue_q_pipeline:
    // T factors and Q on stream4, X = Y~ Q block of columns by block on the main stream), with the arguments of the stage-1 call
    bool pipe_pending = false;
    struct {
        double *Kt = nullptr, *et = nullptr, *Qt = nullptr;
        const SymDev *sym_t = nullptr;
        int *status = nullptr;
        int nt = 0, nT = 1, stride = 0, rep = 0, q_gen = -1;
        bool need_merged = false;
    } pa;
};

// Stage 5 instead of stage 3 (gpcsd_ctx::q_pipe): both halves whole in the register tail, the temporal product first in the
// log-likelihood's tail (GPCSD_LL_ORDER=0), and the caller has promised to form X through loglik_tri_pre (q_pipe_want).
static int ll_order() {                    // GPCSD_LL_ORDER: order of the log-likelihood's two products (capi_fused.inl)
    static const int o = getenv("GPCSD_LL_ORDER") ? atoi(getenv("GPCSD_LL_ORDER")) : 0;
    return o;
}
// Stage 5's kernels form T and Q for every tridiagonal-form consumer they can hold (both halves whole in the register tail, and
// only where a prediction takes the tridiagonal form as well: a chain with an eigenvector-form consumer keeps stage 3 -- its stage
// 4 reads stage 3's T factors) -- pipelined under stage 1 or not, so that Q, X and everything after them have the same bits
// whichever way a call was queued (alone, in a pair, announced by gpcsd_prefetch_pair).
static bool q_stage5_applies(const gpcsd_ctx *c, const SymDev *sym_t) {
    return sym_t && std::max(sym_t->ns, sym_t->na) <= eigh_regtail_rows() &&
           k_tridiag_solve_pass(std::max(sym_t->ns, sym_t->na), c->ntrials) > 0;
}
// ... and pipelined (gpcsd_ctx::q_pipe) when the caller has promised to form X through loglik_tri_pre (q_pipe_want) with the
// temporal product first in the log-likelihood's tail (GPCSD_LL_ORDER=0)
static bool q_pipe_applies(const gpcsd_ctx *c, const SymDev *sym_t) {
    return c->q_pipe && c->q_pipe_want && ll_order() == 0 && q_stage5_applies(c, sym_t);
}
// Stage 5 unpipelined: behind the end of stage 1, Q only (the caller forms X itself)
static void queue_stage5_plain(gpcsd_ctx *c, double *Kt, int nt, double *et, double *Qt, const SymDev *sym_t, int *status,
                               bool need_merged, int nT, int stride) {
    hipStream_t sq = c->stream4;
    GP_HIP(hipStreamWaitEvent(sq, c->ev_t1, 0));
    GP_HIP(hipStreamWaitEvent(sq, c->ev_pc, 0));      // (stream5's readers of the Q about to be rewritten)
    c->q_pipe_x = gpcsd_ctx::QPipeX();
    eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, status, sq, need_merged, nT, stride, -1, 2, /*stage=*/5);
}
// Queue stage 5 (EigState::pipe_pending): on stream4, behind ev_t0, T factor and forward apply panel by panel under the running
// stage 1; on the main stream, behind an event per panel, X[:, panel's columns] = src Q[:, panel's columns] -> xname (eigh_dc.hip).
// The main stream is in order behind every earlier reader of X and behind whatever built src.
static void queue_q_pipeline(gpcsd_ctx *c, EigState &e, const double *src, const char *xname) {
    hipStream_t sq = c->stream4;
    GP_HIP(hipStreamWaitEvent(sq, c->ev_t0, 0));
    GP_HIP(hipStreamWaitEvent(sq, c->ev_pc, 0));      // the side stream's last readers of Q (the previous predictions' Pcat products)
    gpcsd_ctx::QPipeX x;
    x.in = src;
    x.out = c->buf<double>(xname, (size_t)c->nx * c->ntrials * c->nt);
    x.M = c->nx * c->ntrials; x.ld = e.pa.nt;
    x.c0[0] = 0; x.c0[1] = e.pa.sym_t->ns;
    x.rep = e.pa.rep;
    c->q_pipe_x = x;
    try {
        eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, e.pa.Kt, e.pa.nt, e.pa.et, e.pa.Qt, e.pa.sym_t, e.pa.status, sq,
                         e.pa.need_merged, e.pa.nT, e.pa.stride, -1, 2, /*stage=*/5);
    } catch (...) {
        c->q_pipe_x = gpcsd_ctx::QPipeX();
        throw;
    }
    c->q_pipe_x = gpcsd_ctx::QPipeX();
    GP_HIP(hipEventRecord(c->ev_q[c->tgen], sq));
    c->tl("Q end", sq);
    c->q_queued[c->tgen] = true;
    c->q_gen = e.pa.q_gen;
    e.pipe_pending = false;
    e.wait_q = false;                      // (the main stream has passed the last panel's event)
    c->t1_wait_pending = true;             // ... but not the end of stage 1 itself: the readers of d / e wait for ev_t1
    ++c->q_pipe_calls;
}

// The prediction in the basis U (x) Q as well (k_tridiag_solve instead of (W V) / D): with it NO consumer of a staged temporal
// chain reads the spectrum or the eigenvectors, and the chain ends at the tridiagonalisation + Q -- divide & conquer and
// back-transformation (~25 dependent launches, 0.3-0.5 ms beside the GEMM tails at 250-row halves) are not queued at all.
// GPCSD_PRED_TRIDIAG=0: the eigenvector form of the prediction (A/B).
static bool predict_tridiag_applies(int np_s, int np_a, int R) {
    static const bool off = getenv("GPCSD_PRED_TRIDIAG") && getenv("GPCSD_PRED_TRIDIAG")[0] == '0';
    return !off && k_tridiag_solve_pass(std::max(np_s, np_a), R) > 0;
}

// The log-likelihood in the basis U (x) Q (k_ll_tridiag) instead of U (x) V: it needs the temporal chain only up to the
// tridiagonalisation + Q, not the divide & conquer and the back-transformation, which only a prediction's tail waits for.  The
// temporal chain then runs in stages (eigh_pair_device): 1, 2, 4 on stream2, 3 (T factors, Q) beside 2 on stream4.  Measured
// (DESIGN 4.9):
// the paired step 1.134 -> 1.101 ms at cfg3 (50 trials), 0.969 -> 0.929 ms at cfg2; with 400 resident trials the step is bound by
// its GEMMs and the extra pass over W costs 3 % (5.08 against 4.92 ms).  Hence mode 2 (the default): on while the resident block
// is small enough for the step to be latency-bound.  gpcsd_ll_tridiag(ctx, 0 | 1 | 2, ..) / GPCSD_LL_TRIDIAG=0 | 1 force it.
// Where the forms cross over (alternating runs, ms per paired step, eigenvector | tridiagonal form): 384 x 500 with 50 trials
// 1.13 | 1.09, with 75 trials 1.358 | 1.390; 24 x 500 with 200 trials 0.98 | 0.94, with 400 trials 0.975 | 1.085 -- the
// measure max(nx, 64) * nt * ntrials separates the four at 11 M.
static constexpr long LL_TRI_AUTO_MAX = 11L << 20;
static bool ll_tridiag_enabled(const gpcsd_ctx *c) {
    return c->ll_tridiag_mode == 1 ||
           (c->ll_tridiag_mode == 2 && (long)std::max(c->nx, 64) * c->nt * c->ntrials <= LL_TRI_AUTO_MAX);
}

// Before a temporal chain overwrites the single-buffered outputs of a staged predecessor (reflectors, T factors, the
// tridiagonal and its scale, Q): wait on its stream for the readers beside the chain -- the log-likelihood tail on the main
// stream (X = Y~ Q, the recurrences) and stage 3 on stream4 (reads reflectors, writes T factors and Q).
static void staged_chain_guard(gpcsd_ctx *c, hipStream_t s2) {
    if (c->tri_reader_queued[c->tgen]) {
        GP_HIP(hipStreamWaitEvent(s2, c->ev_tri_done[c->tgen], 0));
        c->tri_reader_queued[c->tgen] = false;
    }
    if (c->q_queued[c->tgen]) {
        GP_HIP(hipStreamWaitEvent(s2, c->ev_q[c->tgen], 0));
        c->q_queued[c->tgen] = false;
    }
}

// A staged temporal chain WITH stages 2 and 4 (a consumer in the eigenvector form) is about to start on s2: the late status words
// those stages report into (gpcsd_ctx::STATUS_LATE) are cleared there, behind the predecessor on that stream -- unless
// asynchronous work is outstanding, whose status stays sticky until a synchronising call has collected it.  A chain all of whose
// consumers take the tridiagonal form queues no such stages (and clears nothing): late words are only ever written by the
// paired call with an eigenvector-form prediction or by a non-tridiagonal consumer, and are collected by the call that joins
// that chain.
static void clear_late_status(gpcsd_ctx *c, int *status, hipStream_t s2, bool staged) {
    if (c->async_pending || !staged) return;
    GP_HIP(hipMemsetAsync(status + gpcsd_ctx::STATUS_LATE, 0, (gpcsd_ctx::STATUS_N - gpcsd_ctx::STATUS_LATE) * sizeof(int), s2));
}

// Main stream waits for the spatial chain of this call (no-op when it was reused from the cache or already joined).
static void join_spatial(gpcsd_ctx *c, EigState &e) {
    if (e.wait_spatial) {
        c->tl("main before join S", c->stream);
        GP_HIP(hipStreamWaitEvent(c->stream, c->ev_sjoin, 0));
        c->tl("main after join S", c->stream);
    }
    e.wait_spatial = false;
}

// Decomposition cache: true when side `slot` (0 spatial, 1 temporal) was left in the context's buffers by the previous
// front half with the same key and nothing has used that solver slot since.  Records the key for the next call otherwise.
static bool decomp_cached(gpcsd_ctx *c, int slot, const void *key, size_t bytes) {
    std::vector<unsigned char> &k = c->decomp_key[slot];
    const bool hit = c->decomp_cache_on && c->decomp_gen[slot] == c->eig_gen[slot] && k.size() == bytes &&
                     memcmp(k.data(), key, bytes) == 0;
    if (hit) {
        ++c->decomp_cache_hits;
        return true;
    }
    k.assign((const unsigned char *)key, (const unsigned char *)key + bytes);
    return false;
}

static bool two_stream_front() {            // GPCSD_TWO_STREAM=0: single batched chain (A/B comparisons)
    static const bool off = getenv("GPCSD_TWO_STREAM") && getenv("GPCSD_TWO_STREAM")[0] == '0';
    return !off;
}

// Shared front half of loglik / predict: Ks (+jitter), Kt, eigen-decompositions, D.
//
// The spatial and the temporal side are independent until D = es (x) et + sig2n, and both are chains of small
// latency-bound launches.  Each runs on a stream of its own: stream2 builds Kt and decomposes it, stream3 assembles Ks
// (three GEMMs, ~0.1 ms at 384 electrodes) and decomposes it.  The caller works on the main stream: whatever needs neither
// side first (cross-covariances of predict), join_spatial() before the first use of Qs / es, join_temporal() right before
// the first use of Qt / et / D.  The outputs of both chains are double-buffered (gpcsd_ctx::par), so behind a call that
// returned with work in flight (gpcsd_predict_resident) the chains of this call start at once, beside that call's GEMM tail.
// need_merged = false: the caller runs the folded-basis GEMMs and never reads the merged Qs / Qt / es / et of a folded side
// (one small launch less at the end of each chain).  NOTE: D from the single-stream front half is then in merged order of
// stale spectra -- such callers rebuild it in fold order (join_temporal with a FoldMode).
// join_s = false: the caller calls join_spatial() itself.  Fold views (fold_mode) must be taken AFTER this returns.
// want_tri: the caller's tail works from the temporal tridiagonalisation + Q alone (EigState::tri tells it whether it may).
// merged_s: the spatial side's merged spectrum and eigenvectors are wanted although the temporal side's are not (-1: as need_merged).
EigState front_half(gpcsd_ctx *c, const gpcsd_hparams *hp, double jitter, bool need_merged = true, bool join_s = true,
                    bool want_tri = false, int merged_s = -1) {
    const bool need_merged_s = merged_s < 0 ? need_merged : merged_s != 0;
    const Geo g = resident_geo(c);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    GP_REQUIRE(c->time_nt == c->nt, -4, "time grid has %d points but lfp has nt=%d", c->time_nt, c->nt);
    GP_REQUIRE(g.nx == c->nx, -4, "geometry has %d electrodes but lfp has nx=%d", g.nx, c->nx);
    check_hp(c, hp, c->nx);
    const int nx = c->nx, nt = c->nt;
    hipStream_t s = c->stream;
    EigState e;
    double *Ks = c->buf<double>("Ks", (size_t)nx * nx);          // chain-private inputs (destroyed by the solver)
    double *Kt = c->buf<double>("Kt", (size_t)nt * nt);
    auto outputs = [&]() {                                        // of the current generations
        e.Qs = c->buf<double>(gen_name(c, 0, "Qs"), (size_t)nx * nx);
        e.es = c->buf<double>(gen_name(c, 0, "es"), nx);
        e.Qt = c->buf<double>(gen_name(c, 1, "Qt"), (size_t)nt * nt);
        e.et = c->buf<double>(gen_name(c, 1, "et"), nt);
    };
    e.D = c->buf<double>("D", (size_t)nx * nt);
    e.Dinv = c->buf<double>("Dinv", (size_t)nx * nt);
    // scalars and status words share one allocation so that a call ends with ONE small device-to-host copy
    e.scal = c->buf<double>("scal_status", gpcsd_ctx::RESULT_DOUBLES);
    e.status = reinterpret_cast<int *>(e.scal + gpcsd_ctx::SCAL_N);
    const double *t = (const double *)c->bufs["time_t"].p;
    const bool host_kt = uses_host_kt(hp);
    if (host_kt)
        GP_REQUIRE(c->host_kt_nt == nt && (int)c->host_kt.size() == nt * nt, -3,
                   "host temporal Gram is %d x %d but the resident data has nt=%d", c->host_kt_nt, c->host_kt_nt, nt);
    // a caller-supplied Gram need not commute with the reflection of the time grid (non-stationary kernels): no folding
    const SymDev *sym_s = c->sym_s.ns > 0 ? &c->sym_s : nullptr, *sym_t = (c->sym_t.ns > 0 && !host_kt) ? &c->sym_t : nullptr;
    auto make_kt = [&](hipStream_t st) {
        if (host_kt) c->copy_in(Kt, c->host_kt.data(), (size_t)nt * nt * sizeof(double), st);
        else build_kt(c, hp, t, nt, t, nt, Kt, st);
    };
    // The status words are zeroed at the END of the previous call (finish_call / finish_status), off the critical path of
    // this one; only a call that did not end that way (first call, an exception in between) clears them here -- and then the
    // chains, which report into them, have to be ordered behind that.  (Behind an asynchronous predict they hold its
    // uncollected status and stay as they are.)
    const bool clear_now = !c->status_zeroed && !c->async_pending;
    if (clear_now) GP_HIP(hipMemsetAsync(e.status, 0, gpcsd_ctx::STATUS_N * sizeof(int), s));
    c->status_zeroed = false;
    if (!two_stream_front()) {
        double *d_sig = c->upload_cached<double>("sig2n", hp->sig2n, hp->n_sig2n);
        outputs();
        build_kphi(c, g, hp->R, hp->eps, hp->ell_s, nullptr, 0, jitter, Ks, s, "ks_");
        make_kt(s);
        // the symmetries come from the resident geometry / time grid, so they hold for the Grams built from them
        eig_pair_D(c, Ks, nx, Kt, nt, d_sig, hp->n_sig2n, e.Qs, e.es, e.Qt, e.et, e.D, e.Dinv, e.scal, e.status, sym_s, sym_t,
                   need_merged || need_merged_s);
        e.d_sig = d_sig;
        e.nsig = hp->n_sig2n;
        return e;
    }
    // The temporal chain is the critical path: it is queued first, before any upload of this call.  It takes its
    // hyper-parameters by value and reports numerical failure in its own status word (status[1]; the spatial chain uses
    // status[0]).
    hipStream_t s2 = c->stream2, s3 = c->stream3;
    struct {                                   // everything the temporal side's result depends on
        long epoch;
        int nt, ncomp, kind[GPCSD_MAX_TEMPORAL], merged, fold, host;
        double ell[GPCSD_MAX_TEMPORAL], s2[GPCSD_MAX_TEMPORAL];
    } kt_key;
    memset(&kt_key, 0, sizeof(kt_key));
    kt_key.epoch = c->grid_epoch; kt_key.nt = nt; kt_key.ncomp = hp->n_temporal; kt_key.merged = need_merged;
    kt_key.fold = sym_t != nullptr; kt_key.host = host_kt;
    for (int i = 0; i < hp->n_temporal; ++i) {
        kt_key.kind[i] = hp->kind[i];
        kt_key.ell[i] = hp->ell_t[i];
        kt_key.s2[i] = hp->sigma2_t[i];
    }
    // can this call's consumer take the tridiagonal form?  (decided before the cache is asked: a cached side that stopped at the
    // tridiagonalisation only serves such consumers)
    const bool tfill0 = temporal_fill_applies(c, sym_t, nt, host_kt);
    const bool staged0 = tfill0 && ll_tridiag_enabled(c) && eigh_stageable(sym_t, nt);
    const bool tri_consumer = staged0 && want_tri && !need_merged && hp->n_sig2n == 1;
    bool run_t = !decomp_cached(c, 1, &kt_key, sizeof(kt_key));
    if (!run_t && !tri_consumer && !c->decomp_t_full) {      // cached, but only as far as a tridiagonal-form consumer needs
        run_t = true;
        --c->decomp_cache_hits;
    }
    if (!run_t && tri_consumer && c->q_gen != c->eig_gen[1]) {   // cached by an unstaged chain: no Q / tridiagonal in the buffers
        run_t = true;
        --c->decomp_cache_hits;
    }
    struct {
        long epoch;
        int nx, merged, fold;
        double R, eps, ell[2], jitter;
    } ks_key;
    memset(&ks_key, 0, sizeof(ks_key));
    ks_key.epoch = c->grid_epoch; ks_key.nx = nx; ks_key.merged = need_merged_s; ks_key.fold = sym_s != nullptr;
    ks_key.R = hp->R; ks_key.eps = g.dim == 2 ? hp->eps : 0.0; ks_key.ell[0] = hp->ell_s[0];
    ks_key.ell[1] = g.dim == 2 ? hp->ell_s[1] : 0.0; ks_key.jitter = jitter;
    const bool run_s = !decomp_cached(c, 0, &ks_key, sizeof(ks_key));
    if (run_t) begin_generation(c, 1, s2, clear_now);
    if (run_s) begin_generation(c, 0, s3, clear_now);
    outputs();
    c->tl("call start (main)", s);
    if (run_t) {
        c->tl("T chain start (s2)", s2);
        const bool tfill = tfill0;
        c->tgen ^= 1;                        // the other generation of the temporal class arenas (gpcsd_ctx::tgen)
        staged_chain_guard(c, s2);
        if (tfill) temporal_fill(c, &hp, 1, t, nt, *sym_t, e.status + 1, 0, s2);
        else make_kt(s2);
        // staged whenever it applies: the T factors are then a launch of their own instead of riding in the leaf launch (same
        // bits either way).  A consumer in the tridiagonal form gets stages 1 and 3 only; anybody else all four.
        const bool staged = staged0;
        clear_late_status(c, e.status, s2, staged && !tri_consumer);
        {
            ProfScope ps(c, "eigh_temporal", 9.0 * (double)nt * nt * nt, s2);
            if (staged) {
                // stage 1 (tridiagonalisation), then on this stream stage 2 (divide & conquer) and BESIDE it, on stream4, stage 3
                // (T factors, Q), then stage 4 (back-transformation) behind both.  A log-likelihood in the tridiagonal form
                // starts its tail behind stage 3 and never waits for stages 2 and 4.
                const bool tri = tri_consumer;
                int *late = e.status + gpcsd_ctx::STATUS_LATE;    // stages 2 and 4 report here (gpcsd_ctx::STATUS_LATE)
                // stage 5 instead of stage 3: the tail publishes its progress, the caller's loglik_tri_pre queues the rest
                const bool st5 = tri && q_stage5_applies(c, sym_t);                      // T, Q by stage 5's kernels
                const bool pipe = st5 && q_pipe_applies(c, sym_t);                       // ... under the running stage 1
                if (pipe) GP_HIP(hipEventRecord(c->ev_t0, s2));
                c->pipe_req = st5 ? 1 : 0;
                eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, e.et, e.Qt, sym_t, e.status + 1, s2, need_merged, 1, 0,
                                 -1, 2, /*stage=*/1);
                c->pipe_req = 0;
                GP_HIP(hipEventRecord(c->ev_t1, s2));
                if (!tri)
                    eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, e.et, e.Qt, sym_t, late + 1, s2, need_merged, 1, 0,
                                     -1, 2, /*stage=*/2);
                {
                    hipStream_t sq = c->stream4;
                    if (pipe) {
                        e.pipe_pending = true;
                        e.pa.Kt = Kt; e.pa.nt = nt; e.pa.et = e.et; e.pa.Qt = e.Qt; e.pa.sym_t = sym_t; e.pa.status = e.status + 1;
                        e.pa.need_merged = need_merged; e.pa.nT = 1; e.pa.stride = 0; e.pa.rep = 0; e.pa.q_gen = c->eig_gen[1];
                        c->q_queued[c->tgen] = false;
                        c->q_gen = -1;         // (until stage 5 is queued)
                    } else {
                        if (st5) {
                            queue_stage5_plain(c, Kt, nt, e.et, e.Qt, sym_t, e.status + 1, need_merged, 1, 0);
                        } else {
                            GP_HIP(hipStreamWaitEvent(sq, c->ev_t1, 0));
                            GP_HIP(hipStreamWaitEvent(sq, c->ev_pc, 0));      // (stream5's readers of the Q about to be rewritten)
                            eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, e.et, e.Qt, sym_t, e.status + 1, sq,
                                             need_merged, 1, 0, -1, 2, /*stage=*/3);
                        }
                        GP_HIP(hipEventRecord(c->ev_q[c->tgen], sq));
                        c->q_queued[c->tgen] = true;
                        c->q_gen = c->eig_gen[1];
                    }
                }
                if (!tri) {
                    GP_HIP(hipStreamWaitEvent(s2, c->ev_q[c->tgen], 0));
                    eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, e.et, e.Qt, sym_t, late + 1, s2, need_merged, 1, 0,
                                     -1, 2, /*stage=*/4);
                }
                e.tri = e.wait_q = tri;
                c->decomp_t_full = !tri;
            } else {
                eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, e.et, e.Qt, sym_t, e.status + 1, s2, need_merged, 1, 0,
                                 -1, tfill ? 2 : 0);
                c->decomp_t_full = true;
            }
        }
        GP_HIP(hipEventRecord(c->ev_join, s2));
        c->tl("T chain end (s2)", s2);
        c->decomp_gen[1] = c->eig_gen[1];
        e.wait_temporal = true;
    }
    if (run_s) {
        c->tl("S chain start (s3)", s3);
        const bool sfill = spatial_fill_applies(c, sym_s, nx);
        build_kphi(c, g, hp->R, hp->eps, hp->ell_s, nullptr, 0, sfill ? 0.0 : jitter, Ks, s3, "ks_");
        if (sfill) spatial_fill(c, Ks, nx, 0, 1, &jitter, *sym_s, e.status, 0, s3);
        {
            ProfScope ps(c, "eigh_spatial", 9.0 * (double)nx * nx * nx, s3);
            eigh_pair_device(c, Ks, nx, e.es, e.Qs, sym_s, nullptr, 0, nullptr, nullptr, nullptr, e.status, s3, need_merged_s, 1, 0, -1,
                             sfill ? 1 : 0);
        }
        GP_HIP(hipEventRecord(c->ev_sjoin, s3));
        c->tl("S chain end (s3)", s3);
        c->decomp_gen[0] = c->eig_gen[0];
        e.wait_spatial = true;
    }
    // a temporal side served from the cache: the chain that produced it may still be running its second stage (a log-likelihood
    // in the tridiagonal form returns without waiting for it) -- readers of its eigenvectors wait for that chain's end as usual
    if (!run_t) e.wait_temporal = true;
    if (!run_t && tri_consumer) {    // the temporal side is reused from the cache and its stage-1 outputs are those of this problem
        e.tri = true;
        e.wait_q = c->q_queued[c->tgen];      // (the stage 3 that left Q there may still be running)
    }
    e.d_sig = c->upload_cached<double>("sig2n", hp->sig2n, hp->n_sig2n);
    e.nsig = hp->n_sig2n;
    e.pending = true;
    if (join_s) join_spatial(c, e);
    return e;
}

// ---- folded-basis GEMMs -------------------------------------------------------------------------------------------
// When both Gram matrices were decomposed through their reflection symmetry (eigh.hip, symmetry folding), the half-size
// eigenvector blocks Us/Ua, Vs/Va are all the flat GEMMs need: with F = fold operator (orthogonal), Qs = Fs^T diag(Us, Ua) Pi
// (Pi = the rank merge of the two spectra), so
//     Qs^T Y Qt = Pi_s^T [diag(Us,Ua)^T (Fs Y Ft^T) diag(Vs,Va)] Pi_t ,
// and loglik / predict only ever need sums over all (x', i') pairs or products in which Pi cancels.  The data is folded
// once per geometry (Fs Y Ft^T, one HBM pass), every projection becomes two half-size GEMMs (half the flops), D is built
// from the spectra in fold order.  Needs a scalar noise variance: a per-electrode list is indexed by eigen-RANK in the
// reference, which only the merged order provides.  GPCSD_NO_FOLD_GEMM=1 switches it off (A/B, tests).
// One side may be unfolded (GPCSD1D: 24 electrodes go through Jacobi, 500 time points fold): it then takes part as a single
// "symmetric" block of full size -- its fold operator is the identity (every index a fixed point), U = Q, w = the merged
// spectrum -- and every kernel and GEMM below runs unchanged; empty antisymmetric blocks are skipped.
struct FoldMode {
    bool on = false;
    FoldView fs, ft;            // .on: the side is really folded; otherwise ns = n, na = 0, U = Q, w = eigenvalues
    SymDev sym_s, sym_t;        // effective fold tables of the two sides (identity for an unfolded side)
    int sig() const { return 1 + (fs.on ? 2 : 0) + (ft.on ? 4 : 0); }
};

// identity "symmetry" of n points: every index its own orbit (fixed point)
static SymDev identity_sym(gpcsd_ctx *c, int n) {
    const std::string name = "sym_id_" + std::to_string(n);
    int *d = c->buf<int>(name, (size_t)4 * n);
    int &have = c->int_cache[name];
    if (have != n) {
        std::vector<int> tbl((size_t)4 * n);
        for (int i = 0; i < n; ++i) {
            tbl[i] = i;                  // rep_i
            tbl[n + i] = i;              // rep_j
            tbl[2 * n + i] = i;          // orb
            tbl[3 * n + i] = 0;          // sgn
        }
        c->copy_in(d, tbl.data(), tbl.size() * sizeof(int), c->stream);
        GP_HIP(hipStreamSynchronize(c->stream));
        have = n;
        c->sym_host[d] = std::vector<int>(tbl.begin(), tbl.begin() + 2 * n);
    }
    SymDev sy;
    sy.ns = n; sy.na = 0;
    sy.rep_i = d; sy.rep_j = d + n; sy.orb = d + 2 * n; sy.sgn = d + 3 * n;
    return sy;
}

// fold_spatial = false: the spatial side takes part unfolded (one full-size block, merged eigenvectors) although the electrodes have
// a reflection symmetry -- for prediction sites that do not share it (the cross-covariances then do not commute with the pair of
// reflections, but everything on the temporal side still folds).
static FoldMode fold_mode(gpcsd_ctx *c, const gpcsd_hparams *hp, bool fold_spatial = true) {
    static const bool off = getenv("GPCSD_NO_FOLD_GEMM") && getenv("GPCSD_NO_FOLD_GEMM")[0] == '1';
    FoldMode fm;
    if (off || !c->fold_gemm_on || hp->n_sig2n != 1 || c->nx <= 0 || c->nt <= 0) return fm;
    if (c->sym_s.ns > 0 && fold_spatial) fm.fs = eigh_fold_view(c, 0, &c->sym_s, c->nx);
    if (c->sym_t.ns > 0 && !uses_host_kt(hp)) fm.ft = eigh_fold_view(c, 1, &c->sym_t, c->nt);
    if (!fm.fs.on && !fm.ft.on) return fm;
    fm.on = true;
    if (fm.fs.on) fm.sym_s = c->sym_s;
    else {
        fm.sym_s = identity_sym(c, c->nx);
        fm.fs.ns = c->nx; fm.fs.na = 0;
        fm.fs.w = c->buf<double>(gen_name(c, 0, "es"), c->nx);       // front_half's buffers: merged spectrum / full eigenvectors
        fm.fs.U = c->buf<double>(gen_name(c, 0, "Qs"), (size_t)c->nx * c->nx);
    }
    if (fm.ft.on) fm.sym_t = c->sym_t;
    else {
        fm.sym_t = identity_sym(c, c->nt);
        fm.ft.ns = c->nt; fm.ft.na = 0;
        fm.ft.w = c->buf<double>(gen_name(c, 1, "et"), c->nt);
        fm.ft.U = c->buf<double>(gen_name(c, 1, "Qt"), (size_t)c->nt * c->nt);
    }
    return fm;
}

// Fs Y Ft^T in the layout of the resident data ([fold x][r][fold t]); rebuilt when data, geometry or time grid change
static const double *folded_lfp(gpcsd_ctx *c, const FoldMode &fm) {
    // one copy per combination of folded sides (a model whose prediction sites do not share the electrodes' symmetry alternates
    // between "both sides folded" for loglik and "time only" for predict: neither evicts the other)
    const int sig = fm.sig();
    double *Yf = c->buf<double>("lfp_fold" + std::to_string(sig), (size_t)c->nx * c->ntrials * c->nt);
    if (!(c->lfp_fold_sig & (1 << sig))) {
        k_fold_lfp(c, c->d_lfp, c->nx, c->ntrials, c->nt, fm.sym_s, fm.sym_t, Yf, c->stream);
        c->lfp_fold_sig |= 1 << sig;
    }
    return Yf;
}

// The two parity blocks of one folded product.  Equal block shapes (even grids: the usual case) go out as ONE batched
// launch -- twice the tiles per launch, so the last partial wave of workgroups weighs half as much -- otherwise as two.
// Returns true when batched (an EPI_QUAD pair then leaves the whole sum in g0.quad_out, else g1.quad_out holds the rest).
static bool gemm_pair(gpcsd_ctx *c, GemmDesc g0, const GemmDesc &g1, hipStream_t s) {
    if (g1.M <= 0 || g1.N <= 0 || g1.K <= 0) {                       // unfolded side: no antisymmetric block
        gemm_f64(c, g0, s);
        return true;
    }
    if (g0.M == g1.M && g0.N == g1.N && g0.K == g1.K && g0.lda == g1.lda && g0.ldb == g1.ldb && g0.ldc == g1.ldc) {
        g0.batch = 2;
        g0.sA = g1.A - g0.A;
        g0.sB = g1.B - g0.B;
        g0.sC = (g0.C && g1.C) ? g1.C - g0.C : 0;
        g0.sD = (g0.D && g1.D) ? g1.D - g0.D : 0;
        gemm_f64(c, g0, s);
        return true;
    }
    gemm_f64(c, g0, s);
    gemm_f64(c, g1, s);
    return false;
}

// out[p-block rows] = U_p^T in[p-block rows] for p = symmetric, antisymmetric: the spatial projection in the folded basis
static void fold_proj_spatial(gpcsd_ctx *c, const FoldView &fs, const double *in, double *out, long ncols, hipStream_t s) {
    GemmDesc g[2];
    for (int p = 0; p < 2; ++p) {
        const int np = p ? fs.na : fs.ns;
        const long r0 = p ? fs.ns : 0;
        g[p].M = np; g[p].N = (int)ncols; g[p].K = np;
        g[p].A = fs.U + (p ? (size_t)fs.ns * fs.ns : 0); g[p].lda = np; g[p].transA = true;
        g[p].B = in + r0 * ncols; g[p].ldb = ncols;
        g[p].C = out + r0 * ncols; g[p].ldc = ncols;
        g[p].prof_name = "gemm_proj_spatial";
    }
    gemm_pair(c, g[0], g[1], s);
}

// Main stream waits for the temporal chain; then D and sum(log D) -- from the spectra in fold order when fm is on.
// No-op after the single-stream front half unless the fold order is asked for.
// sumlog = false: the caller either does not need sum(log D) (predict) or folds the final sum of the partials into a later
// launch (loglik: the reduce of the quadratic form); returns the number of partials left in "buildD_partials" (0: none built).
int join_temporal(gpcsd_ctx *c, EigState &e, const FoldMode *fm = nullptr, bool sumlog = true) {
    join_spatial(c, e);
    if (e.wait_temporal) {
        c->tl("main before join T", c->stream);
        GP_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->tl("main after join T", c->stream);
    }
    e.wait_temporal = false;
    double *out = sumlog ? e.scal : nullptr;
    int np = 0;
    if (fm && fm->on) np = k_build_D(c, fm->fs.w, c->nx, fm->ft.w, c->nt, e.d_sig, e.nsig, e.D, e.Dinv, out, c->stream);
    else if (e.pending) np = k_build_D(c, e.es, c->nx, e.et, c->nt, e.d_sig, e.nsig, e.D, e.Dinv, out, c->stream);
    e.pending = false;
    return np;
}

// Fold the status words of a fused call into one code (0: fine).  late: the call joined the whole temporal chain, so the words of
// its stages 2 and 4 are complete and count as well.
static int fold_status(const int *st, bool late) {
    int r = 0;
    const int n = late ? gpcsd_ctx::STATUS_N : gpcsd_ctx::STATUS_LATE;
    for (int i = 0; i < n && r == 0; ++i) r = st[i];   // [0]/[2] spatial, [1]/[3] temporal, [4..7] late stages of the temporal chain
    return r;
}

// End of a fused call: one copy brings back the leading `nscal` scalars and the status words, then the stream is drained.
// A log-likelihood in the tridiagonal form (e.tri) has no stages 2 and 4 of its own (front_half does not queue them for such a
// consumer): it neither reads nor clears the late status words (gpcsd_ctx::STATUS_LATE), which may belong to an earlier paired
// call's eigenvector-form prediction that nobody has joined yet.
int finish_call(gpcsd_ctx *c, const EigState &e, double *scal_out, int nscal) {
    double *host = c->h_result;                        // pinned: a true asynchronous copy, no staging
    const bool late = !e.tri;
    c->tl("sync call end (main)", c->stream);
    c->download(host, e.scal, gpcsd_ctx::RESULT_DOUBLES * sizeof(double));
    // clean status words for the next call, after the copy
    GP_HIP(hipMemsetAsync(e.status, 0, (late ? gpcsd_ctx::STATUS_N : gpcsd_ctx::STATUS_LATE) * sizeof(int), c->stream));
    c->sync();
    c->status_zeroed = true;
    c->async_pending = false;              // whatever an asynchronous predict left in the status words has been collected now
    if (c->prof_mode == 1) c->prof_collect();
    for (int i = 0; i < nscal; ++i) scal_out[i] = host[i];
    int st[gpcsd_ctx::STATUS_N];
    memcpy(st, host + gpcsd_ctx::SCAL_N, sizeof(st));
    const int bad = fold_status(st, late);
    if (bad != 0) {
        char b[128];
        snprintf(b, sizeof(b), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", bad);
        c->last_error = b;
        c->decomp_gen[0] = c->decomp_gen[1] = -1;     // a failed decomposition is never reused
        return bad > 0 ? bad : 1;
    }
    return 0;
}

int finish_status(gpcsd_ctx *c, const int *d_status, int nwords = 4) {
    int st[gpcsd_ctx::STATUS_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    c->download(st, d_status, nwords * sizeof(int));
    c->sync();
    if (c->prof_mode == 1) c->prof_collect();
    const int bad = fold_status(st, true);     // (words beyond nwords are zero)
    if (bad != 0) {
        char b[128];
        snprintf(b, sizeof(b), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", bad);
        c->last_error = b;
        c->decomp_gen[0] = c->decomp_gen[1] = -1;     // a failed decomposition is never reused (a retry with the same hp re-solves)
        return bad > 0 ? bad : 1;
    }
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" int gpcsd_version(void) { return 100; }

extern "C" const char *gpcsd_last_error(gpcsd_ctx *ctx) {
    if (ctx) return ctx->last_error.c_str();
    return g_last_error.c_str();
}

// ---- the streams of a context ---------------------------------------------------------------------------------------------
// A context runs on four streams (main; the temporal and the spatial chain; the side stream).  Which hardware queue -- and
// behind it which pipe of the command processor -- a stream lands on is decided by the runtime and the kernel driver when the
// stream is created.  A closed context's streams are therefore not destroyed: they go back to a per-device pool and the next
// context of the process takes them over (last in, first out), so that models opened one after another run on the very same
// queues, whatever number of models the process has opened before; opening a model costs no queue creation either.  Contexts
// that are alive together get sets of their own.  GPCSD_STREAM_POOL=0: create and destroy per context (A/B).  Pooled streams
// live until the process ends.
// (What the pool does NOT cure, measured with tools/two_models_probe.py: on some boxes the step loop of a model that is not the
// first one of its process contains ONE stall of 5-25 ms in which no kernel of the process runs (rocprofv3 kernel trace; the host
// sits in hipEventSynchronize) -- 0.63 -> 0.87-0.92 ms per cfg2 step over a 100-step loop.  With or without the pool, Python's
// collector off, malloc trimming off, NUMA balancing off on the box; other boxes show it in one loop of twelve.  DESIGN 6.)
struct StreamSet {
    hipStream_t s[4];
};
static std::mutex g_stream_pool_mu;
static std::map<int, std::vector<StreamSet>> g_stream_pool;
static long g_stream_sets_created = 0, g_stream_sets_reused = 0;
static bool stream_pool_on() {
    static const bool on = !(getenv("GPCSD_STREAM_POOL") && getenv("GPCSD_STREAM_POOL")[0] == '0');
    return on;
}
static StreamSet stream_set_acquire(int device) {
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        auto &v = g_stream_pool[device];
        if (stream_pool_on() && !v.empty()) {
            const StreamSet ss = v.back();
            v.pop_back();
            ++g_stream_sets_reused;
            return ss;
        }
    }
    StreamSet ss{};
    try {
    int prio_least = 0, prio_greatest = 0;
    GP_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    // GPCSD_RESERVE_CUS=k (A/B): the main stream -- every flood of GEMM tiles -- may not use the first k compute units (bit i of the
    // mask = XCD i % 8, so k = 8 takes one CU of every XCD): the chains' small launches then find a drained CU at once instead of
    // waiting for a tile to retire (their launches take 2-3 x as long in the step loop as alone), at k / 256 of the GEMM rate.
    static const int reserve = getenv("GPCSD_RESERVE_CUS") ? atoi(getenv("GPCSD_RESERVE_CUS")) : 0;
    if (reserve > 0) {
        hipDeviceProp_t prop;
        GP_HIP(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0xffffffffu);
        for (int i = 0; i < std::min(reserve, ncu - 8); ++i) mask[i / 32] &= ~(1u << (i % 32));
        if (ncu % 32) mask[words - 1] &= (1u << (ncu % 32)) - 1u;
        GP_HIP(hipExtStreamCreateWithCUMask(&ss.s[0], (uint32_t)words, mask.data()));
    } else {
        GP_HIP(hipStreamCreateWithFlags(&ss.s[0], hipStreamNonBlocking));
    }
    GP_HIP(hipStreamCreateWithPriority(&ss.s[1], hipStreamNonBlocking, prio_greatest));
    GP_HIP(hipStreamCreateWithPriority(&ss.s[2], hipStreamNonBlocking, prio_greatest));
    const char *ev = getenv("GPCSD_S4_PRIO");
    if (ev && ev[0] == '0') GP_HIP(hipStreamCreateWithFlags(&ss.s[3], hipStreamNonBlocking));
    else GP_HIP(hipStreamCreateWithPriority(&ss.s[3], hipStreamNonBlocking, prio_greatest));
    } catch (...) {                        // a stream that could not be created: the ones that were do not leak
        for (int i = 0; i < 4; ++i)
            if (ss.s[i]) (void)hipStreamDestroy(ss.s[i]);
        throw;
    }
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    ++g_stream_sets_created;
    return ss;
}
// (the caller has drained the streams)
static void stream_set_release(int device, const StreamSet &ss) {
    if (!ss.s[0] && !ss.s[1] && !ss.s[2] && !ss.s[3]) return;
    if (stream_pool_on() && ss.s[0] && ss.s[1] && ss.s[2] && ss.s[3]) {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        g_stream_pool[device].push_back(ss);
        return;
    }
    for (int i = 0; i < 4; ++i)
        if (ss.s[i]) (void)hipStreamDestroy(ss.s[i]);
}

/* which: 0 main, 1 temporal chain, 2 spatial chain, 3 side stream -> the hipStream_t as an integer; which = -1: sets created so
 * far in this process, -2: sets taken over from a closed context */
extern "C" int gpcsd_ctx_stream_handle(gpcsd_ctx *c, int which, unsigned long long *out) {
    if (!out) return -3;
    if (which < 0) {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        *out = (unsigned long long)(which == -1 ? g_stream_sets_created : g_stream_sets_reused);
        return which >= -2 ? 0 : -3;
    }
    if (!c || which > 3) return -3;
    hipStream_t s[4] = {c->stream, c->stream2, c->stream3, c->stream4};
    *out = (unsigned long long)(uintptr_t)s[which];
    return 0;
}

extern "C" int gpcsd_ctx_create(int device, gpcsd_ctx **out) {
    if (!out) return -1;
    *out = nullptr;
    gpcsd_ctx *c = nullptr;
    try {
        int ndev = 0;
        GP_HIP(hipGetDeviceCount(&ndev));
        GP_REQUIRE(ndev > 0, -5, "no HIP device visible");
        GP_REQUIRE(device >= 0 && device < ndev, -5, "device %d out of range (have %d)", device, ndev);
        GP_HIP(hipSetDevice(device));
        c = new gpcsd_ctx();
        c->device = device;
        c->timeline_on = getenv("GPCSD_TIMELINE") && getenv("GPCSD_TIMELINE")[0] == '1';
        if (const char *ev = getenv("GPCSD_LL_TRIDIAG")) c->ll_tridiag_mode = ev[0] == '0' ? 0 : ev[0] == '1' ? 1 : 2;
        if (const char *ev = getenv("GPCSD_TAIL_EARLY_EXIT")) c->tail_early_exit = ev[0] != '0';
        if (const char *ev = getenv("GPCSD_PAIR_SHARE_X")) c->pair_share_x = ev[0] != '0';
        if (const char *ev = getenv("GPCSD_PAIR_SHARE_S")) c->pair_share_s = ev[0] != '0';
        if (const char *ev = getenv("GPCSD_Q_PIPE")) c->q_pipe = ev[0] != '0';
        if (const char *ev = getenv("GPCSD_QPIPE_GATE_TICKS")) c->q_gate_ticks = strtoull(ev, nullptr, 10);
        if (const char *ev = getenv("GPCSD_PRED_CHUNKED")) c->pred_chunked = ev[0] != '0';
        // The two chains are the critical path and made of small launches; when they run beside another call's GEMM tail
        // (thousands of workgroups) each of those launches would otherwise queue behind the tiles: high priority.
        // stream4: stage 3 / stage 5 of a staged temporal chain (on the log-likelihood's critical path, beside the previous call's
        // products) and the prediction's two small side products -- high priority as well (GPCSD_S4_PRIO=0: A/B)
        // The four streams come from a per-device pool that contexts return theirs to (stream_set_acquire).
        {
            const StreamSet ss = stream_set_acquire(device);
            c->stream = ss.s[0];
            c->stream2 = ss.s[1];
            c->stream3 = ss.s[2];
            c->stream4 = ss.s[3];
        }
        // (stream5 = stream4.  A fifth stream for the prediction's side products -- any priority -- made the STEP slower, 1.5 instead
        // of 0.95 ms at cfg3, with or without stage 5: measured, not understood; the runtime's mapping of streams to hardware queues
        // is not ours to see.  The name stays: the side products' stream.)
        c->stream5 = c->stream4;
        GP_HIP(hipEventCreateWithFlags(&c->ev_sjoin, hipEventDisableTiming));
        for (int i = 0; i < 4; ++i) GP_HIP(hipEventCreateWithFlags(&c->ev_mark[i / 2][i % 2], hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_pc, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_chol_a, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_chol_d, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) GP_HIP(hipEventCreateWithFlags(&c->ev_q[i], hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_t0, hipEventDisableTiming));
        for (int i = 0; i < 8; ++i) GP_HIP(hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_prelude, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_m1, hipEventDisableTiming));
        if (const char *ev = getenv("GPCSD_QPIPE_MASK")) c->q_pipe_mask = atoi(ev);
        for (int i = 0; i < 2; ++i) GP_HIP(hipEventCreateWithFlags(&c->ev_tri_done[i], hipEventDisableTiming));
        GP_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_result), gpcsd_ctx::RESULT_DOUBLES * sizeof(double), hipHostMallocDefault));
        GP_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_ll), gpcsd_ctx::LL_SLOTS * gpcsd_ctx::RESULT_DOUBLES * sizeof(double),
                             hipHostMallocDefault));
        for (auto &sl : c->ll_slot) GP_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        GP_HIP(hipMalloc(reinterpret_cast<void **>(&c->h_chol_flag), 64));          // (device memory: agent-scope atomics, chol.hip)
        GP_HIP(hipMemset(c->h_chol_flag, 0, 64));
        *out = c;
        return 0;
    } catch (const HipError &e) {
        // the same teardown as a context that was created: events, page-locked blocks and streams made so far (every handle of a
        // fresh gpcsd_ctx is null until it is created, and gpcsd_ctx_destroy checks each)
        if (c) (void)gpcsd_ctx_destroy(c);
        return fail(nullptr, e);
    }
}

extern "C" int gpcsd_ctx_destroy(gpcsd_ctx *c) {
    if (c) pair_prefetch_drop(c);
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : c->bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (c->d_lfp) (void)hipFree(c->d_lfp);
    for (auto &kv : c->prof)
        for (auto &p : kv.second.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto &kv : c->graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (c->ev_sjoin) (void)hipEventDestroy(c->ev_sjoin);
    for (int i = 0; i < 4; ++i)
        if (c->ev_mark[i / 2][i % 2]) (void)hipEventDestroy(c->ev_mark[i / 2][i % 2]);
    if (c->stream5 && c->stream5 != c->stream4) (void)hipStreamDestroy(c->stream5);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_aux) (void)hipEventDestroy(c->ev_aux);
    if (c->ev_pc) (void)hipEventDestroy(c->ev_pc);
    if (c->ev_chol_a) (void)hipEventDestroy(c->ev_chol_a);
    if (c->ev_chol_d) (void)hipEventDestroy(c->ev_chol_d);
    for (int i = 0; i < 2; ++i)
        if (c->ev_q[i]) (void)hipEventDestroy(c->ev_q[i]);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    for (int i = 0; i < 8; ++i)
        if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]);
    if (c->ev_prelude) (void)hipEventDestroy(c->ev_prelude);
    if (c->ev_m1) (void)hipEventDestroy(c->ev_m1);
    for (int i = 0; i < 2; ++i)
        if (c->ev_tri_done[i]) (void)hipEventDestroy(c->ev_tri_done[i]);
    stream_set_release(c->device, StreamSet{{c->stream, c->stream2, c->stream3, c->stream4}});      // (drained above)
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->h_ll) (void)hipHostFree(c->h_ll);
    if (c->h_chol_flag) (void)hipFree(c->h_chol_flag);
    if (c->stage_ring) (void)hipHostFree(c->stage_ring);
    for (auto &kv : c->pinned_bufs)
        if (kv.second.first) (void)hipHostFree(kv.second.first);
    for (int k = 0; k < 2; ++k) {
        if (c->bounce[k]) (void)hipHostFree(c->bounce[k]);
        if (c->bounce_ev[k]) (void)hipEventDestroy(c->bounce_ev[k]);
    }
    if (c->tail_clk_host) (void)hipHostFree(c->tail_clk_host);
    for (auto &sl : c->ll_slot)
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    delete c;
    return 0;
}

// Page-locked host memory for results (the Python layer keeps a pool of these blocks and hands them out as the NumPy arrays
// predict() returns): a device-to-host copy into pageable memory is staged by the runtime at ~9 GB/s and pays a page fault
// per fresh 4 KiB page, into a pinned block it is one DMA at link speed.  No context needed (the blocks outlive contexts).
extern "C" int gpcsd_host_alloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return -3;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(nullptr, HipError{-100 - (int)e, std::string("hipHostMalloc: ") + hipGetErrorString(e)});
        *out = nullptr;
        return -100 - (int)e;
    }
    return 0;
}

// PCI address of a device ("0000:c5:00.0"), for a host that wants to place itself on the device's NUMA node
// (/sys/bus/pci/devices/<address>/numa_node): no context needed.
extern "C" int gpcsd_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 16) return -3;
    buf[0] = 0;
    const hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(nullptr, HipError{-100 - (int)e, std::string("hipDeviceGetPCIBusId: ") + hipGetErrorString(e)});
        return -100 - (int)e;
    }
    return 0;
}

extern "C" int gpcsd_host_free(void *p) {
    if (!p) return 0;
    return hipHostFree(p) == hipSuccess ? 0 : -1;
}

extern "C" int gpcsd_device_synchronize(gpcsd_ctx *c) {
    GP_API_BEGIN(c)
    GP_HIP(hipStreamSynchronize(c->stream2));
    GP_HIP(hipStreamSynchronize(c->stream3));
    GP_HIP(hipStreamSynchronize(c->stream4));
    GP_HIP(hipStreamSynchronize(c->stream5));
    c->sync();
    c->timeline_dump();
    return drain_async(c);
    GP_API_END(c)
}

// ------------------------------------------------------------------------------------------------
// resident data
// ------------------------------------------------------------------------------------------------
extern "C" int gpcsd_set_lfp(gpcsd_ctx *c, const double *lfp, int nx, int nt, int ntrials) {
    GP_API_BEGIN(c)
    GP_REQUIRE(lfp && nx > 0 && nt > 0 && ntrials > 0, -3, "set_lfp: bad shape (%d,%d,%d)", nx, nt, ntrials);
    const size_t n = (size_t)nx * nt * ntrials;
    double *stage = c->upload<double>("lfp_stage", lfp, n);
    if (c->d_lfp) {
        c->sync();
        GP_HIP(hipFree(c->d_lfp));
        c->d_lfp = nullptr;
    }
    GP_HIP(hipMalloc((void **)&c->d_lfp, n * sizeof(double)));
    k_swap_last2(c, stage, c->d_lfp, nx, nt, ntrials, c->stream);     // (x,t,r) -> (x,r,t)
    c->sync();
    c->nx = nx; c->nt = nt; c->ntrials = ntrials;
    c->lfp_fold_sig = 0;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_set_geometry_1d(gpcsd_ctx *c, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl) {
    GP_API_BEGIN(c)
    GP_REQUIRE(x && gl_x && gl_w && nx > 0 && ngl > 0, -3, "set_geometry_1d: bad arguments");
    c->upload<double>("geo_x", x, nx);
    c->upload<double>("geo_gx1", gl_x, ngl);
    c->upload<double>("geo_gw1", gl_w, ngl);
    c->sync();
    c->dim = 1; c->geo_nx = nx; c->ngl1 = ngl; c->ngl2 = 0;
    ++c->grid_epoch;
    // electrodes mirror-symmetric about the centre of a symmetric quadrature rule -> Ks commutes with the reflection
    c->sym_s = SymDev();
    c->sym_z = SymDev();
    c->sym_z_pts.clear();
    c->lfp_fold_sig = 0;
    c->geo_host.assign(x, x + nx);
    double ctr;
    if (rule_is_symmetric(gl_x, gl_w, ngl, &ctr)) {
        const bool refl[1] = {true};
        c->sym_s = find_symmetry(c, "sym_s_tbl", x, nx, 1, &ctr, refl);
        c->sym_s_ctr[0] = ctr;
        c->sym_s_refl[0] = true;
    }
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_set_geometry_2d(gpcsd_ctx *c, const double *xy, int nx, const double *gl_x1, const double *gl_w1, int ngl1,
                                     const double *gl_x2, const double *gl_w2, int ngl2) {
    GP_API_BEGIN(c)
    GP_REQUIRE(xy && gl_x1 && gl_w1 && gl_x2 && gl_w2 && nx > 0 && ngl1 > 0 && ngl2 > 0, -3, "set_geometry_2d: bad arguments");
    c->upload<double>("geo_x", xy, (size_t)nx * 2);
    c->upload<double>("geo_gx1", gl_x1, ngl1);
    c->upload<double>("geo_gw1", gl_w1, ngl1);
    c->upload<double>("geo_gx2", gl_x2, ngl2);
    c->upload<double>("geo_gw2", gl_w2, ngl2);
    c->sync();
    c->dim = 2; c->geo_nx = nx; c->ngl1 = ngl1; c->ngl2 = ngl2;
    ++c->grid_epoch;
    // reflections about the centre of the (symmetric) tensor quadrature rule that map the electrode set onto itself:
    // point reflection first (the Neuropixels checkerboard has it), then single-axis mirrors
    c->sym_s = SymDev();
    c->sym_z = SymDev();
    c->sym_z_pts.clear();
    c->lfp_fold_sig = 0;
    c->geo_host.assign(xy, xy + (size_t)nx * 2);
    double ctr[2];
    const bool s1 = rule_is_symmetric(gl_x1, gl_w1, ngl1, &ctr[0]), s2 = rule_is_symmetric(gl_x2, gl_w2, ngl2, &ctr[1]);
    const bool cand[3][2] = {{true, true}, {false, true}, {true, false}};
    for (int k = 0; k < 3 && c->sym_s.ns == 0; ++k) {
        if ((cand[k][0] && !s1) || (cand[k][1] && !s2)) continue;
        c->sym_s = find_symmetry(c, "sym_s_tbl", xy, nx, 2, ctr, cand[k]);
        c->sym_s_ctr[0] = ctr[0]; c->sym_s_ctr[1] = ctr[1];
        c->sym_s_refl[0] = cand[k][0]; c->sym_s_refl[1] = cand[k][1];
    }
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_set_time(gpcsd_ctx *c, const double *t, int nt) {
    GP_API_BEGIN(c)
    GP_REQUIRE(t && nt > 0, -3, "set_time: bad arguments");
    c->upload<double>("time_t", t, nt);
    c->sync();
    c->time_nt = nt;
    ++c->grid_epoch;
    c->time_host.assign(t, t + nt);
    c->lfp_fold_sig = 0;
    // a time grid symmetric about its midpoint (any uniform grid) makes every stationary Kt centro-symmetric
    double lo = t[0], hi = t[0];
    for (int i = 1; i < nt; ++i) {
        lo = std::min(lo, t[i]);
        hi = std::max(hi, t[i]);
    }
    const double ctr = 0.5 * (lo + hi);
    const bool refl[1] = {true};
    c->sym_t = find_symmetry(c, "sym_t_tbl", t, nt, 1, &ctr, refl);
    return 0;
    GP_API_END(c)
}

#include "capi_operators.inl"
#include "capi_fused.inl"
#include "capi_grad.inl"
#include "capi_measure.inl"
#include "capi_dist.inl"
