// extern "C" surface of libgpcsd_hip.so (include/gpcsd_hip.h) and the host-side orchestration of the hot path.
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
static void drain_after_failure(gpcsd_ctx *c) {
    if (!c) return;
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->stream4) (void)hipStreamSynchronize(c->stream4);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
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
        GemmDesc u;                                // U[(x,g1)][h2] = sum_g2 A[(x,g1)][g2] K2[g2][h2]
        u.M = g.nx * n1; u.N = n2; u.K = n2;
        u.A = A; u.lda = n2; u.B = K2; u.ldb = n2; u.C = U; u.ldc = n2;
        u.prof_name = "gemm_Ks_AK2";
        gemm_f64(c, u, s);
        GemmDesc v;                                // T_x (n1 x n2) = K1^T U_x, one small product per electrode
        v.M = n1; v.N = n2; v.K = n1;
        v.A = K1; v.lda = n1; v.transA = true; v.B = U; v.ldb = n2; v.C = T; v.ldc = n2;
        v.batch = g.nx; v.sA = 0; v.sB = G; v.sC = G;
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
    gemm_f64(c, d2, s);
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
    const char *const *tg = eigh_fold_tags(1);
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
    const char *const *tg = eigh_fold_tags(0);
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
        GP_HIP(hipMemcpyAsync(out, c->host_kt_cross.data() + (size_t)cc * ntstar * nt, (size_t)ntstar * nt * sizeof(double),
                              hipMemcpyHostToDevice, s));
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
    GP_HIP(hipMemcpyAsync(d, tbl.data(), tbl.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    GP_HIP(hipStreamSynchronize(c->stream));
    out.ns = ns; out.na = na;
    out.rep_i = d; out.rep_j = d + ns; out.orb = d + 2 * ns; out.sgn = d + 2 * ns + n;
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
};

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
EigState front_half(gpcsd_ctx *c, const gpcsd_hparams *hp, double jitter, bool need_merged = true, bool join_s = true) {
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
    e.scal = c->buf<double>("scal_status", 64 + 2);
    e.status = reinterpret_cast<int *>(e.scal + 64);
    const double *t = (const double *)c->bufs["time_t"].p;
    const bool host_kt = uses_host_kt(hp);
    if (host_kt)
        GP_REQUIRE(c->host_kt_nt == nt && (int)c->host_kt.size() == nt * nt, -3,
                   "host temporal Gram is %d x %d but the resident data has nt=%d", c->host_kt_nt, c->host_kt_nt, nt);
    // a caller-supplied Gram need not commute with the reflection of the time grid (non-stationary kernels): no folding
    const SymDev *sym_s = c->sym_s.ns > 0 ? &c->sym_s : nullptr, *sym_t = (c->sym_t.ns > 0 && !host_kt) ? &c->sym_t : nullptr;
    auto make_kt = [&](hipStream_t st) {
        if (host_kt) GP_HIP(hipMemcpyAsync(Kt, c->host_kt.data(), (size_t)nt * nt * sizeof(double), hipMemcpyHostToDevice, st));
        else build_kt(c, hp, t, nt, t, nt, Kt, st);
    };
    // The status words are zeroed at the END of the previous call (finish_call / finish_status), off the critical path of
    // this one; only a call that did not end that way (first call, an exception in between) clears them here -- and then the
    // chains, which report into them, have to be ordered behind that.  (Behind an asynchronous predict they hold its
    // uncollected status and stay as they are.)
    const bool clear_now = !c->status_zeroed && !c->async_pending;
    if (clear_now) GP_HIP(hipMemsetAsync(e.status, 0, 4 * sizeof(int), s));
    c->status_zeroed = false;
    if (!two_stream_front()) {
        double *d_sig = c->upload_cached<double>("sig2n", hp->sig2n, hp->n_sig2n);
        outputs();
        build_kphi(c, g, hp->R, hp->eps, hp->ell_s, nullptr, 0, jitter, Ks, s, "ks_");
        make_kt(s);
        // the symmetries come from the resident geometry / time grid, so they hold for the Grams built from them
        eig_pair_D(c, Ks, nx, Kt, nt, d_sig, hp->n_sig2n, e.Qs, e.es, e.Qt, e.et, e.D, e.Dinv, e.scal, e.status, sym_s, sym_t,
                   need_merged);
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
    const bool run_t = !decomp_cached(c, 1, &kt_key, sizeof(kt_key));
    struct {
        long epoch;
        int nx, merged, fold;
        double R, eps, ell[2], jitter;
    } ks_key;
    memset(&ks_key, 0, sizeof(ks_key));
    ks_key.epoch = c->grid_epoch; ks_key.nx = nx; ks_key.merged = need_merged; ks_key.fold = sym_s != nullptr;
    ks_key.R = hp->R; ks_key.eps = g.dim == 2 ? hp->eps : 0.0; ks_key.ell[0] = hp->ell_s[0];
    ks_key.ell[1] = g.dim == 2 ? hp->ell_s[1] : 0.0; ks_key.jitter = jitter;
    const bool run_s = !decomp_cached(c, 0, &ks_key, sizeof(ks_key));
    if (run_t) begin_generation(c, 1, s2, clear_now);
    if (run_s) begin_generation(c, 0, s3, clear_now);
    outputs();
    c->tl("call start (main)", s);
    if (run_t) {
        c->tl("T chain start (s2)", s2);
        const bool tfill = temporal_fill_applies(c, sym_t, nt, host_kt);
        if (tfill) temporal_fill(c, &hp, 1, t, nt, *sym_t, e.status + 1, 0, s2);
        else make_kt(s2);
        {
            ProfScope ps(c, "eigh_temporal", 9.0 * (double)nt * nt * nt, s2);
            eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, e.et, e.Qt, sym_t, e.status + 1, s2, need_merged, 1, 0,
                             -1, tfill ? 2 : 0);
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
            eigh_pair_device(c, Ks, nx, e.es, e.Qs, sym_s, nullptr, 0, nullptr, nullptr, nullptr, e.status, s3, need_merged, 1, 0, -1,
                             sfill ? 1 : 0);
        }
        GP_HIP(hipEventRecord(c->ev_sjoin, s3));
        c->tl("S chain end (s3)", s3);
        c->decomp_gen[0] = c->eig_gen[0];
        e.wait_spatial = true;
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
        GP_HIP(hipMemcpyAsync(d, tbl.data(), tbl.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        GP_HIP(hipStreamSynchronize(c->stream));
        have = n;
    }
    SymDev sy;
    sy.ns = n; sy.na = 0;
    sy.rep_i = d; sy.rep_j = d + n; sy.orb = d + 2 * n; sy.sgn = d + 3 * n;
    return sy;
}

static FoldMode fold_mode(gpcsd_ctx *c, const gpcsd_hparams *hp) {
    static const bool off = getenv("GPCSD_NO_FOLD_GEMM") && getenv("GPCSD_NO_FOLD_GEMM")[0] == '1';
    FoldMode fm;
    if (off || !c->fold_gemm_on || hp->n_sig2n != 1 || c->nx <= 0 || c->nt <= 0) return fm;
    if (c->sym_s.ns > 0) fm.fs = eigh_fold_view(c, 0, &c->sym_s, c->nx);
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
    double *Yf = c->buf<double>("lfp_fold", (size_t)c->nx * c->ntrials * c->nt);
    if (c->lfp_fold_sig != fm.sig()) {                               // 0: invalid; else which sides the resident copy folds
        k_fold_lfp(c, c->d_lfp, c->nx, c->ntrials, c->nt, fm.sym_s, fm.sym_t, Yf, c->stream);
        c->lfp_fold_sig = fm.sig();
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

// End of a fused call: one copy brings back the leading `nscal` scalars and the status words, then the stream is drained.
int finish_call(gpcsd_ctx *c, const EigState &e, double *scal_out, int nscal) {
    double *host = c->h_result;                        // pinned: a true asynchronous copy, no staging
    c->tl("sync call end (main)", c->stream);
    c->download(host, e.scal, 66 * sizeof(double));
    GP_HIP(hipMemsetAsync(e.status, 0, 4 * sizeof(int), c->stream));   // clean status words for the next call, after the copy
    c->sync();
    c->status_zeroed = true;
    c->async_pending = false;              // whatever an asynchronous predict left in the status words has been collected now
    if (c->prof_mode == 1) c->prof_collect();
    for (int i = 0; i < nscal; ++i) scal_out[i] = host[i];
    int st[4];
    memcpy(st, host + 64, sizeof(st));
    for (int i = 1; i < 4 && st[0] == 0; ++i) st[0] = st[i];   // [1]: temporal chain; [2], [3]: second replica of a paired call
    if (st[0] != 0) {
        char b[128];
        snprintf(b, sizeof(b), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", st[0]);
        c->last_error = b;
        c->decomp_gen[0] = c->decomp_gen[1] = -1;     // a failed decomposition is never reused
        return st[0] > 0 ? st[0] : 1;
    }
    return 0;
}

int finish_status(gpcsd_ctx *c, const int *d_status) {
    int st[4];
    c->download(st, d_status, sizeof(st));
    c->sync();
    if (c->prof_mode == 1) c->prof_collect();
    for (int i = 1; i < 4 && st[0] == 0; ++i) st[0] = st[i];   // [1]: temporal chain; [2], [3]: second replica of a paired call
    if (st[0] != 0) {
        char b[128];
        snprintf(b, sizeof(b), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", st[0]);
        c->last_error = b;
        c->decomp_gen[0] = c->decomp_gen[1] = -1;     // a failed decomposition is never reused (a retry with the same hp re-solves)
        return st[0] > 0 ? st[0] : 1;
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
        GP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        // The two chains are the critical path and made of small launches; when they run beside another call's GEMM tail
        // (thousands of workgroups) each of those launches would otherwise queue behind the tiles: high priority.
        int prio_least = 0, prio_greatest = 0;
        GP_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        GP_HIP(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_greatest));
        GP_HIP(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_greatest));
        GP_HIP(hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking));
        GP_HIP(hipEventCreateWithFlags(&c->ev_sjoin, hipEventDisableTiming));
        for (int i = 0; i < 4; ++i) GP_HIP(hipEventCreateWithFlags(&c->ev_mark[i / 2][i % 2], hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&c->ev_pc, hipEventDisableTiming));
        GP_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_result), 66 * sizeof(double), hipHostMallocDefault));
        GP_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_ll), gpcsd_ctx::LL_SLOTS * 66 * sizeof(double), hipHostMallocDefault));
        for (auto &sl : c->ll_slot) GP_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        *out = c;
        return 0;
    } catch (const HipError &e) {
        delete c;
        return fail(nullptr, e);
    }
}

extern "C" int gpcsd_ctx_destroy(gpcsd_ctx *c) {
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
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_aux) (void)hipEventDestroy(c->ev_aux);
    if (c->ev_pc) (void)hipEventDestroy(c->ev_pc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->h_ll) (void)hipHostFree(c->h_ll);
    if (c->stage_ring) (void)hipHostFree(c->stage_ring);
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

extern "C" int gpcsd_host_free(void *p) {
    if (!p) return 0;
    return hipHostFree(p) == hipSuccess ? 0 : -1;
}

extern "C" int gpcsd_device_synchronize(gpcsd_ctx *c) {
    GP_API_BEGIN(c)
    GP_HIP(hipStreamSynchronize(c->stream2));
    GP_HIP(hipStreamSynchronize(c->stream3));
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
    if (n > 1) GP_HIP(hipMemcpyAsync(de, e, (n - 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
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

extern "C" int gpcsd_fold_gemm(gpcsd_ctx *c, int on, long *calls) {
    GP_API_BEGIN(c)
    if (on >= 0) c->fold_gemm_on = on != 0;
    if (calls) *calls = c->fold_gemm_calls;
    return 0;
    GP_API_END(c)
}

// ------------------------------------------------------------------------------------------------
// fused hot calls
// ------------------------------------------------------------------------------------------------
// End of an asynchronous loglik: the scalars and status words go to the pinned block behind an event; nothing is waited for
// and the status words are left alone (the chains of later calls may already be reporting into them).
static int finish_loglik_async(gpcsd_ctx *c, const EigState &e, bool two) {
    const int k = (c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS;       // callers have checked that a slot is free
    gpcsd_ctx::LlSlot &sl = c->ll_slot[k];
    GP_HIP(hipMemcpyAsync(c->h_ll + 66 * k, e.scal, 66 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GP_HIP(hipEventRecord(sl.ev, c->stream));
    ++c->ll_count;
    sl.done = false;
    sl.two = two;
    c->async_pending = true;
    c->status_zeroed = false;
    return 0;
}

// Folded-basis tail of the log-likelihood: the two projections as 2 + 2 half-size GEMMs, the quadratic form as two partial
// sums (one when the parity blocks went out as one batched launch: returns true), sum(log D) folded into the reduce launch.
static bool loglik_fold_tail(gpcsd_ctx *c, EigState &e, const FoldMode &fm, const double *Yf, double *W) {
    const int nx = c->nx, nt = c->nt, R = c->ntrials;
    hipStream_t s = c->stream;
    ++c->fold_gemm_calls;
    fold_proj_spatial(c, fm.fs, Yf, W, (long)R * nt, s);
    const int nparts = join_temporal(c, e, &fm, false);         // sum(log D): summed by the reduce launch of the GEMM below
    GemmDesc g2[2];
    g2[0].extra_sum_in = c->buf<double>("buildD_partials", 256);
    g2[0].extra_sum_n = nparts;
    g2[0].extra_sum_out = e.scal;
    for (int p = 0; p < 2; ++p) {
        const int np = p ? fm.ft.na : fm.ft.ns, c0 = p ? fm.ft.ns : 0;
        g2[p].M = nx * R; g2[p].N = np; g2[p].K = np;
        g2[p].A = W + c0; g2[p].lda = nt;
        g2[p].B = fm.ft.U + (p ? (size_t)fm.ft.ns * fm.ft.ns : 0); g2[p].ldb = np;
        g2[p].epi = EPI_QUAD; g2[p].D = e.Dinv + c0; g2[p].rdiv = R; g2[p].ldd = nt; g2[p].quad_out = e.scal + 1 + p;
        g2[p].prof_name = "gemm_proj_temporal_quad";
    }
    return gemm_pair(c, g2[0], g2[1], s);
}

static int loglik_parts_impl(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out2, bool async) {
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    if (async && c->prof_mode == 1) {     // fenced profiling (mode 1): evaluate now, hand the result over at the wait
        gpcsd_ctx::LlSlot &sl = c->ll_slot[(c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS];
        sl.rc = loglik_parts_impl(c, hp, sl.out, false);
        sl.done = true;
        ++c->ll_count;
        return 0;
    }
    const FoldMode fm0 = fold_mode(c, hp);                          // the decision; its views are of the previous generation
    const double *Yf = fm0.on ? folded_lfp(c, fm0) : nullptr;
    EigState e = front_half(c, hp, hp->jitter, !fm0.on);
    const FoldMode fm = fold_mode(c, hp);                           // views of the generation the front half just launched
    const int nx = c->nx, nt = c->nt, R = c->ntrials;
    hipStream_t s = c->stream;
    double *W = c->buf<double>("proj_W", (size_t)nx * R * nt);
    if (fm.on) {
        const bool batched = loglik_fold_tail(c, e, fm, Yf, W);
        if (async) return finish_loglik_async(c, e, !batched);
        double h3[3] = {0.0, 0.0, 0.0};
        const int rc = finish_call(c, e, h3, 3);
        out2[0] = h3[0];
        out2[1] = batched ? h3[1] : h3[1] + h3[2];
        return rc;
    }
    GemmDesc g1;                          // W[x'][(r,t)] = sum_x Qs[x][x'] Y[x][(r,t)]        (gpcsd1d.py:125 inner dot)
    g1.M = nx; g1.N = R * nt; g1.K = nx;
    g1.A = e.Qs; g1.lda = nx; g1.transA = true;
    g1.B = c->d_lfp; g1.ldb = (long)R * nt;
    g1.C = W; g1.ldc = (long)R * nt;
    g1.prof_name = "gemm_proj_spatial";
    gemm_f64(c, g1, s);
    join_temporal(c, e);
    GemmDesc g2;                          // alpha[(x',r)][i'] = sum_t W[(x',r)][t] Qt[t][i'];  quad = sum alpha^2 / D
    g2.M = nx * R; g2.N = nt; g2.K = nt;
    g2.A = W; g2.lda = nt;
    g2.B = e.Qt; g2.ldb = nt;
    g2.epi = EPI_QUAD; g2.D = e.Dinv; g2.rdiv = R; g2.ldd = nt; g2.quad_out = e.scal + 1;
    g2.prof_name = "gemm_proj_temporal_quad";
    gemm_f64(c, g2, s);
    if (async) return finish_loglik_async(c, e, false);
    return finish_call(c, e, out2, 2);
}

extern "C" int gpcsd_loglik_parts(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out2) {
    GP_API_BEGIN(c)
    GP_REQUIRE(out2 != nullptr, -3, "null output");
    return loglik_parts_impl(c, hp, out2, false);
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_parts_async(gpcsd_ctx *c, const gpcsd_hparams *hp) {
    if (c && c->ll_count >= gpcsd_ctx::LL_SLOTS)   // refused before anything is touched: the outstanding ones stay collectable
        return fail(c, HipError{-3, "loglik_parts_async: too many asynchronous evaluations outstanding (collect with "
                                    "gpcsd_loglik_parts_wait)"});
    GP_API_BEGIN(c)
    return loglik_parts_impl(c, hp, nullptr, true);
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_parts_wait(gpcsd_ctx *c, double *out2) {
    if (c && (!out2 || c->ll_count == 0))
        return fail(c, HipError{-3, out2 ? "loglik_parts_wait: no asynchronous evaluation pending" : "null output"});
    GP_API_BEGIN(c)
    const int k = c->ll_head;
    gpcsd_ctx::LlSlot &sl = c->ll_slot[k];
    c->ll_head = (k + 1) % gpcsd_ctx::LL_SLOTS;
    --c->ll_count;
    if (sl.done) {
        out2[0] = sl.out[0];
        out2[1] = sl.out[1];
        return sl.rc;
    }
    GP_HIP(hipEventSynchronize(sl.ev));
    const double *host = c->h_ll + 66 * k;
    out2[0] = host[0];
    out2[1] = sl.two ? host[1] + host[2] : host[1];
    int st[4];
    memcpy(st, host + 64, sizeof(st));
    for (int i = 1; i < 4 && st[0] == 0; ++i) st[0] = st[i];
    if (st[0] != 0) {                     // this evaluation's, or an earlier asynchronous call's that nobody collected yet
        char b[160];
        snprintf(b, sizeof(b), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", st[0]);
        c->last_error = b;
        // The status words are sticky while asynchronous work is outstanding (nobody may clear them under a running chain).
        // Now that a failure has been reported: drain everything and clear them, so that evaluations queued from here on
        // start clean.  Evaluations that were ALREADY outstanding copied the words as they stood and report the failure too
        // (a failed wait poisons the ones queued before it returned; documented in gpcsd_hip.h).
        drain_after_failure(c);
        if (int *dst = reinterpret_cast<int *>(c->buf<double>("scal_status", 64 + 2) + 64)) {
            GP_HIP(hipMemsetAsync(dst, 0, 4 * sizeof(int), c->stream));
            GP_HIP(hipStreamSynchronize(c->stream));
            c->status_zeroed = true;
        }
        return st[0] > 0 ? st[0] : 1;
    }
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_loglik(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out) {
    if (!out) return fail(c, HipError{-3, "null output"});
    double p[2] = {0.0, 0.0};
    int rc = gpcsd_loglik_parts(c, hp, p);
    if (rc < 0) return rc;
    *out = -0.5 * (double)c->ntrials * p[0] - 0.5 * p[1];      // gpcsd1d.py:122,127-128
    return rc;
}

// Reflection symmetry of the prediction sites under the SAME reflection as the electrodes (then the cross-covariances
// commute with the pair of involutions and fold as well).  Cached on the site coordinates; ns == 0: none.
static const SymDev &site_symmetry(gpcsd_ctx *c, const double *z, int nz, int dim) {
    const size_t cnt = (size_t)nz * dim;
    if (c->sym_z_pts.size() == cnt && memcmp(c->sym_z_pts.data(), z, cnt * sizeof(double)) == 0) return c->sym_z;
    c->sym_z_pts.assign(z, z + cnt);
    if (c->geo_host.size() == cnt && memcmp(c->geo_host.data(), z, cnt * sizeof(double)) == 0) c->sym_z = c->sym_s;
    else c->sym_z = find_symmetry(c, "sym_z_tbl", z, nz, dim, c->sym_s_ctr, c->sym_s_refl);
    return c->sym_z;
}

// predict_impl in the folded basis (see FoldMode).  Prediction sites and times must share the symmetry of the grids:
//   out_c = Fz^T [ diag_p( (Kc_pp^T U_p) ) Bm~ diag_q( V_q^T Kt*_c,qq ) ] Ft   with Bm~ = (diag(U)^T Y~ diag(V)) / D~ ,
// every flat GEMM split in its two parity blocks; the last pass unfolds sites and times while it transposes.
static int predict_fold(gpcsd_ctx *c, const gpcsd_hparams *hp, EigState &e, const FoldMode &fm, const double *Yf, const SymDev &sz,
                        const double *dz, int nz, const double *dts, int type, bool want_lists, bool async,
                        const std::function<void()> *after_spatial_join = nullptr) {
    const Geo g = resident_geo(c);
    const int nx = c->nx, nt = c->nt, R = c->ntrials, C = hp->n_temporal;
    const long RT = (long)R * nt;
    const int ns = fm.fs.ns, na = fm.fs.na, nts = fm.ft.ns, nta = fm.ft.na, nzs = sz.ns, nza = sz.na;
    hipStream_t s = c->stream;
    double *W = c->buf<double>("proj_W", (size_t)nx * RT);
    double *Bm = c->buf<double>("pred_B", (size_t)nx * RT);
    const double *t = (const double *)c->bufs["time_t"].p;
    double *Kc = c->buf<double>("pred_Kcross", (size_t)nx * nz);
    const size_t kcf_sz = (size_t)ns * nzs + (size_t)na * nza;
    double *Kcf = c->buf<double>("pred_Kcross_fold", 2 * kcf_sz);
    double *S = c->buf<double>("pred_S", (size_t)nz * RT);
    // comp~ and Pcat keep every (parity, component) block of columns on a 128-byte boundary (block widths padded to a multiple
    // of 16 doubles): the final relayout pass reads comp~ in 16-column pieces per trial row, and unaligned blocks (250 columns)
    // made every piece straddle two cache lines -- 291 MB fetched for 154 MB of comp~ per cfg3 step
    const int ntsP = (nts + 15) & ~15, ntaP = (nta + 15) & ~15;
    // (+16: a row stride that is a power of two -- 1024 doubles at nt = 500 -- walks the same HBM channels row after row)
    const long ldcomp = (long)C * (ntsP + ntaP) + 16;
    double *comp = c->buf<double>("pred_comp", std::max((size_t)C * nz * RT, (size_t)nz * R * ldcomp));
    double *Kts = c->buf<double>("pred_Ktstar", (size_t)C * nt * nt);
    const size_t ktf_sz = (size_t)nts * nts + (size_t)nta * nta;
    double *Ktf = c->buf<double>("pred_Ktstar_fold", (size_t)C * ktf_sz);
    const size_t m1_sz = (size_t)nzs * ns + (size_t)nza * na;
    double *M1 = c->buf<double>("pred_M1", 2 * std::max(m1_sz, (size_t)nz * nx));
    const size_t pc_s = (size_t)nts * C * ntsP;                     // Pcat_sym: nts rows of C * ntsP columns; Pcat_anti follows
    double *Pc = c->buf<double>("pred_Pc", std::max(pc_s + (size_t)nta * C * ntaP, (size_t)C * nt * nt));
    const size_t out_elems = (size_t)nz * RT;
    ++c->fold_gemm_calls;
    // what needs neither decomposition runs first, beside both chains: the cross-covariances and the prediction-time Grams,
    // folded
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        double *kf = Kcf + (size_t)(which - 1) * kcf_sz;
        if (which == 1) build_kphig(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, Kc, s);        // gpcsd1d.py:273
        else build_kphi(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, 0.0, Kc, s);               // gpcsd1d.py:275
        k_sym_fold_rect(c, Kc, nz, fm.sym_s, sz, kf, kf + (size_t)ns * nzs, s);
    }
    for (int cc = 0; cc < C; ++cc) {
        temporal_cross_gram(c, hp, cc, dts, nt, t, nt, Kts + (size_t)cc * nt * nt, s);
        k_sym_fold_rect(c, Kts + (size_t)cc * nt * nt, nt, fm.sym_t, fm.sym_t, Ktf + cc * ktf_sz,
                        Ktf + cc * ktf_sz + (size_t)nts * nts, s);
    }
    // then everything that needs only the spatial eigenvectors, beside the temporal eigensolver
    join_spatial(c, e);
    // gpcsd_loglik_predict_async: the log-likelihood's whole tail goes here, in front of everything of predict that needs a
    // decomposition -- it is what the caller waits for
    if (after_spatial_join) (*after_spatial_join)();
    fold_proj_spatial(c, fm.fs, Yf, W, RT, s);                      // W~ = diag(U)^T Y~
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        const double *kf = Kcf + (size_t)(which - 1) * kcf_sz;
        for (int p = 0; p < 2; ++p) {
            const int np = p ? na : ns, nzp = p ? nza : nzs;
            if (np == 0 || nzp == 0) continue;
            GemmDesc gm;                  // M1_p[zq][x'] = sum_xq Kc~_pp[xq][zq] U_p[xq][x']
            gm.M = nzp; gm.N = np; gm.K = np;
            gm.A = kf + (p ? (size_t)ns * nzs : 0); gm.lda = nzp; gm.transA = true;
            gm.B = fm.fs.U + (p ? (size_t)ns * ns : 0); gm.ldb = np;
            gm.C = M1 + (size_t)(which - 1) * m1_sz + (p ? (size_t)nzs * ns : 0); gm.ldc = np;
            gm.prof_name = "gemm_pred_M1";
            gemm_f64(c, gm, s);
        }
    }
    join_temporal(c, e, &fm, false);      // predict never reads sum(log D)
    GemmDesc g2[2];                       // Bm~[:, p block] = (W~[:, p block] V_p) / D~
    for (int p = 0; p < 2; ++p) {
        const int np = p ? nta : nts, c0 = p ? nts : 0;
        g2[p].M = nx * R; g2[p].N = np; g2[p].K = np;
        g2[p].A = W + c0; g2[p].lda = nt;
        g2[p].B = fm.ft.U + (p ? (size_t)nts * nts : 0); g2[p].ldb = np;
        g2[p].C = Bm + c0; g2[p].ldc = nt;
        g2[p].epi = EPI_DIV_D; g2[p].D = e.Dinv + c0; g2[p].rdiv = R; g2[p].ldd = nt;
        g2[p].prof_name = "gemm_pred_temporal_div";
    }
    // Pcat = V^T Kt*~ needs the temporal eigenvectors and the folded prediction-time Grams only: it runs on a stream of its
    // own beside the large Bm~ / S~ products of the main stream instead of in front of them (two small launches off the
    // serial tail) -- not on stream2, where it would sit between this call's temporal chain and the next call's
    GP_HIP(hipEventRecord(c->ev_aux, s));                            // Kt*~ and the temporal eigenvectors are complete here
    GP_HIP(hipStreamWaitEvent(c->stream4, c->ev_aux, 0));
    for (int p = 0; p < 2; ++p) {
        const int np = p ? nta : nts, npP = p ? ntaP : ntsP;
        if (np == 0) continue;
        GemmDesc gp;                      // Pcat_p[i'][cc*npP + b] = sum_j V_p[j][i'] Kt*~_cc,pp[j][b], all components batched
        gp.M = np; gp.N = np; gp.K = np;
        gp.A = fm.ft.U + (p ? (size_t)nts * nts : 0); gp.lda = np; gp.transA = true;
        gp.B = Ktf + (p ? (size_t)nts * nts : 0); gp.ldb = np;
        gp.C = Pc + (p ? pc_s : 0); gp.ldc = (long)C * npP;
        gp.batch = C; gp.sA = 0; gp.sB = (long)ktf_sz; gp.sC = npP;
        gp.prof_name = "gemm_pred_Pc";
        gemm_f64(c, gp, c->stream4);
    }
    GP_HIP(hipEventRecord(c->ev_pc, c->stream4));
    c->tl("Pc end (s4)", c->stream4);
    gemm_pair(c, g2[0], g2[1], s);
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        double *o_sum = c->buf<double>(which == 1 ? "pred_out_csd" : "pred_out_lfp", out_elems);
        double *o_list = want_lists ? c->buf<double>(which == 1 ? "pred_out_csd_list" : "pred_out_lfp_list", out_elems * C)
                                    : nullptr;
        GemmDesc g5[2], g6[2];
        for (int p = 0; p < 2; ++p) {     // S~[p rows] = M1_p Bm~[p rows]
            const int np = p ? na : ns, nzp = p ? nza : nzs;
            g5[p].M = nzp; g5[p].N = (int)RT; g5[p].K = np;
            g5[p].A = M1 + (size_t)(which - 1) * m1_sz + (p ? (size_t)nzs * ns : 0); g5[p].lda = np;
            g5[p].B = Bm + (p ? (size_t)ns * RT : 0); g5[p].ldb = RT;
            g5[p].C = S + (p ? (size_t)nzs * RT : 0); g5[p].ldc = RT;
            g5[p].prof_name = "gemm_pred_cross";
        }
        gemm_pair(c, g5[0], g5[1], s);
        if (which == 1 || !(type & 1)) GP_HIP(hipStreamWaitEvent(s, c->ev_pc, 0));     // Pcat (stream4) before its first use
        for (int p = 0; p < 2; ++p) {     // comp~[(zq, r)][p][cc][b] = sum_i' S~[(zq, r)][p block i'] Pcat_p[i'][cc*npP + b]
            const int np = p ? nta : nts, npP = p ? ntaP : ntsP, c0 = p ? nts : 0;
            // (the padding columns between two components are computed along -- whatever Pcat holds there only reaches comp~'s
            // own padding columns, which nobody reads; the last component's padding is left out)
            g6[p].M = nz * R; g6[p].N = (C - 1) * npP + np; g6[p].K = np;
            g6[p].A = S + c0; g6[p].lda = nt;
            g6[p].B = Pc + (p ? pc_s : 0); g6[p].ldb = (long)C * npP;
            g6[p].C = comp + (p ? (size_t)C * ntsP : 0); g6[p].ldc = ldcomp;
            g6[p].prof_name = "gemm_pred_tstar";
        }
        if (gemm_pred_unfold_supported(C, (long)nz * R, nt)) {
            // ... as ONE launch whose epilogue unfolds in site and time, turns (r, t) into (t, r) and sums the components:
            // comp~ is never written (gemm_f64.hip: gemm_pred_unfold_kernel)
            PredUnfoldDesc pu{};
            pu.S = S; pu.lds = nt;
            pu.Pc[0] = Pc; pu.Pc[1] = Pc + pc_s;
            pu.ldp[0] = (long)C * ntsP; pu.ldp[1] = (long)C * ntaP;
            pu.npP[0] = ntsP; pu.npP[1] = ntaP;
            pu.K[0] = nts; pu.K[1] = nta;
            pu.kcol0[0] = 0; pu.kcol0[1] = nts;
            pu.nb = nts; pu.nba = nta;
            pu.ncolS = (long)nzs * R; pu.ncolA = (long)nza * R; pu.anti_row0 = (long)nzs * R;
            pu.R = R; pu.nt = nt; pu.C = C;
            pu.sz = sz; pu.st = fm.sym_t;
            pu.list = o_list; pu.list_stride = (long)out_elems; pu.sum = o_sum;
            gemm_pred_unfold(c, pu, s);
        } else {
            gemm_pair(c, g6[0], g6[1], s);
            k_unfold_swap_sum(c, comp, C, o_list, (long)out_elems, o_sum, R, nt, sz, fm.sym_t, s, ntsP, ntaP, ldcomp);
        }
    }
    c->tl("predict end (main)", s);
    if (async && c->prof_mode != 1) {     // results stay on the device: return with the tail still in flight
        c->async_pending = true;
        c->status_zeroed = false;
        return 0;
    }
    return finish_call(c, e, nullptr, 0);
}

// Posterior mean into ctx-owned device buffers, already in the reference's output layout (z, t, trial):
//   pred_out_csd / pred_out_lfp            (nz, ntstar, R)
//   pred_out_csd_list / pred_out_lfp_list  (C, nz, ntstar, R)     when want_lists
static int predict_impl(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                        int type, bool want_lists, bool async = false) {
    GP_REQUIRE(z && tstar && nz > 0 && ntstar > 0, -3, "predict: bad arguments");
    GP_REQUIRE(type >= 1 && type <= 3, -3, "predict: type must be CSD(1), LFP(2) or BOTH(3)");
    GP_REQUIRE(c->nt > 0 && ntstar == c->nt, -22,
               "predict: len(t)=%d must equal the training nt=%d (the reference's reshape raises ValueError, gpcsd1d.py:279)",
               ntstar, c->nt);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    // folded basis when the grids, the prediction sites and the prediction times all share the reflection symmetries
    const FoldMode fm0 = fold_mode(c, hp);         // the decision only: views are taken after the front half
    // a folded side needs its outputs on a grid with the same symmetry (t* = t; mirror-symmetric sites); an unfolded side
    // takes any sites / times
    const bool t_ok = fm0.on && ntstar == c->nt &&
                      (!fm0.ft.on || ((int)c->time_host.size() == c->nt &&
                                      memcmp(c->time_host.data(), tstar, (size_t)ntstar * sizeof(double)) == 0));
    if (t_ok) {
        const SymDev sz = fm0.fs.on ? site_symmetry(c, z, nz, c->dim) : identity_sym(c, nz);
        if (sz.ns > 0 && sz.ns + sz.na == nz) {
            // the chains go first (they need no upload of this call), then the host-side uploads
            EigState ef = front_half(c, hp, 0.0, false, /*join_s=*/false);  // no jitter in predict (gpcsd1d.py:258)
            const FoldMode fm = fold_mode(c, hp);
            const double *Yf = folded_lfp(c, fm);
            double *dzf = c->upload_cached<double>("pred_z", z, (size_t)nz * c->dim);
            double *dtf = c->upload_cached<double>("pred_tstar", tstar, ntstar);
            return predict_fold(c, hp, ef, fm, Yf, sz, dzf, nz, dtf, type, want_lists, async);
        }
    }
    EigState e = front_half(c, hp, 0.0);           // no jitter in predict (gpcsd1d.py:258)
    const Geo g = resident_geo(c);
    const int nx = c->nx, nt = c->nt, R = c->ntrials, C = hp->n_temporal;
    const long RT = (long)R * nt;
    hipStream_t s = c->stream;
    double *W = c->buf<double>("proj_W", (size_t)nx * RT);
    double *Bm = c->buf<double>("pred_B", (size_t)nx * RT);
    GemmDesc g1;                          // W = Qs^T Y
    g1.M = nx; g1.N = (int)RT; g1.K = nx;
    g1.A = e.Qs; g1.lda = nx; g1.transA = true;
    g1.B = c->d_lfp; g1.ldb = RT; g1.C = W; g1.ldc = RT;
    g1.prof_name = "gemm_proj_spatial";
    gemm_f64(c, g1, s);
    // invy = (Qs (x) Qt) vec(Bm) (gpcsd1d.py:262-265) is never formed: the cross-covariance contraction
    //   out_c = Kc^T Qs Bm Qt^T Kt*_c  is re-associated as  (Kc^T Qs) Bm (Qt^T Kt*_c),
    // i.e. two small (n^3) products M1, Pc and two flat GEMMs, instead of back-projecting to the original bases first
    // (saves 2 nx^2 nt + 2 nx nt^2 flops per trial; identical up to rounding).
    double *dz = c->upload_cached<double>("pred_z", z, (size_t)nz * g.dim);
    double *dts = c->upload_cached<double>("pred_tstar", tstar, ntstar);
    const double *t = (const double *)c->bufs["time_t"].p;
    double *Kc = c->buf<double>("pred_Kcross", (size_t)nx * nz);
    double *S = c->buf<double>("pred_S", (size_t)nz * RT);
    double *comp = c->buf<double>("pred_comp", (size_t)C * nz * RT);
    double *Kts = c->buf<double>("pred_Ktstar", (size_t)C * ntstar * nt);
    double *M1 = c->buf<double>("pred_M1", (size_t)2 * nz * nx);
    double *Pc = c->buf<double>("pred_Pc", (size_t)C * nt * nt);
    const size_t out_elems = (size_t)nz * RT;
    // Everything that needs only Qs is queued before the join, i.e. it runs beside the temporal eigensolver:
    // cross-covariances Kc, M1 = Kc^T Qs for the requested outputs, and the prediction-time temporal Grams.
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        if (which == 1) build_kphig(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, Kc, s);        // gpcsd1d.py:273
        else build_kphi(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, 0.0, Kc, s);               // gpcsd1d.py:275
        GemmDesc gm;                      // M1[z][x'] = sum_x Kc[x][z] Qs[x][x']
        gm.M = nz; gm.N = nx; gm.K = nx;
        gm.A = Kc; gm.lda = nz; gm.transA = true; gm.B = e.Qs; gm.ldb = nx; gm.C = M1 + (size_t)(which - 1) * nz * nx; gm.ldc = nx;
        gm.prof_name = "gemm_pred_M1";
        gemm_f64(c, gm, s);
    }
    for (int cc = 0; cc < C; ++cc) {
        // Ktstar_c = cov_c.compute_Kt(tstar): (ntstar, nt); its FIRST axis is contracted with the training
        // time index (reference quirk when tstar != t)      gpcsd1d.py:277-279
        temporal_cross_gram(c, hp, cc, dts, ntstar, t, nt, Kts + (size_t)cc * ntstar * nt, s);
    }
    join_temporal(c, e, nullptr, false);      // predict never reads sum(log D)
    GemmDesc g2;                          // Bm = (W Qt) / D
    g2.M = nx * R; g2.N = nt; g2.K = nt;
    g2.A = W; g2.lda = nt; g2.B = e.Qt; g2.ldb = nt; g2.C = Bm; g2.ldc = nt;
    g2.epi = EPI_DIV_D; g2.D = e.Dinv; g2.rdiv = R; g2.ldd = nt;
    g2.prof_name = "gemm_pred_temporal_div";
    gemm_f64(c, g2, s);
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        double *o_sum = c->buf<double>(which == 1 ? "pred_out_csd" : "pred_out_lfp", out_elems);
        double *o_list = want_lists ? c->buf<double>(which == 1 ? "pred_out_csd_list" : "pred_out_lfp_list", out_elems * C)
                                    : nullptr;
        GemmDesc g5;                      // S[z][(r,i')] = sum_x' M1[z][x'] Bm[x'][(r,i')]
        g5.M = nz; g5.N = (int)RT; g5.K = nx;
        g5.A = M1 + (size_t)(which - 1) * nz * nx; g5.lda = nx; g5.B = Bm; g5.ldb = RT; g5.C = S; g5.ldc = RT;
        g5.prof_name = "gemm_pred_cross";
        gemm_f64(c, g5, s);
        for (int cc = 0; cc < C; ++cc) {
            GemmDesc gp;                  // Pcat[i'][cc*nt + t'] = sum_j Qt[j][i'] Ktstar_cc[j][t']
            gp.M = nt; gp.N = nt; gp.K = ntstar;
            gp.A = e.Qt; gp.lda = nt; gp.transA = true; gp.B = Kts + (size_t)cc * ntstar * nt; gp.ldb = nt;
            gp.C = Pc + (size_t)cc * nt; gp.ldc = (long)C * nt;
            gp.prof_name = "gemm_pred_Pc";
            gemm_f64(c, gp, s);
        }
        // All temporal components in ONE flat GEMM: out[(z,r)][cc*nt + t'] = sum_i' S[(z,r)][i'] Pcat[i'][cc*nt + t'],
        // then one pass writes every component in the reference's (z, t, r) layout plus their sum (no read-modify-write
        // epilogue, one launch instead of C, a single relayout pass instead of C + 1).
        GemmDesc g6;
        g6.M = nz * R; g6.N = C * nt; g6.K = nt;
        g6.A = S; g6.lda = nt; g6.B = Pc; g6.ldb = (long)C * nt; g6.C = comp; g6.ldc = (long)C * nt;
        g6.prof_name = "gemm_pred_tstar";
        gemm_f64(c, g6, s);
        k_swap_last2_sum(c, comp, C, o_list, (long)out_elems, o_sum, nz, R, nt, s);     // (z,r,c,t) -> (c,z,t,r), sum over c
    }
    return finish_call(c, e, nullptr, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// loglik + predict as ONE queued call with the four decompositions batched two by two.
//
// Chains of small dependent launches do not overlap on this part (1.4x at best, however many queues), but
// replicas inside one chain are nearly free (gpcsd_eigh_batch: 8 problems in 1.09 ms against 0.93 ms for one).  So when a
// caller wants the log-likelihood at one hyper-parameter set and the prediction at another (the same set without jitter, in
// practice), the two temporal problems go through ONE chain as two replicas and the two spatial problems through another:
// two chains per pair of calls instead of four.  Every problem is still solved (nothing is reused between the two unless the
// decomposition cache is on and the temporal hyper-parameters coincide: then that side is solved once, as the cache would).
// Results: the bits of the two calls made separately.
struct PairFront {
    EigState e[2];
    FoldMode fm[2];
};

static bool same_temporal(const gpcsd_hparams *a, const gpcsd_hparams *b) {
    if (a->n_temporal != b->n_temporal) return false;
    for (int i = 0; i < a->n_temporal; ++i)
        if (a->kind[i] != b->kind[i] || a->ell_t[i] != b->ell_t[i] || a->sigma2_t[i] != b->sigma2_t[i]) return false;
    return true;
}

// Both sets decomposed, set b's results at replica b of the generation just started (folded-basis callers only).
static void front_half_pair(gpcsd_ctx *c, const gpcsd_hparams *const hp[2], const double jitter[2], PairFront &out) {
    const Geo g = resident_geo(c);
    const int nx = c->nx, nt = c->nt;
    const long nxx = (long)nx * nx, ntt = (long)nt * nt;
    hipStream_t s = c->stream, s2 = c->stream2, s3 = c->stream3;
    const SymDev *sym_s = c->sym_s.ns > 0 ? &c->sym_s : nullptr, *sym_t = c->sym_t.ns > 0 ? &c->sym_t : nullptr;
    const double *t = (const double *)c->bufs["time_t"].p;
    const int nT = (c->decomp_cache_on && same_temporal(hp[0], hp[1])) ? 1 : 2;      // replicas of the temporal problem
    double *scal = c->buf<double>("scal_status", 64 + 2);
    int *status = reinterpret_cast<int *>(scal + 64);
    const bool clear_now = !c->status_zeroed && !c->async_pending;
    if (clear_now) GP_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int), s));
    c->status_zeroed = false;
    begin_generation(c, 1, s2, clear_now);
    begin_generation(c, 0, s3, clear_now);
    // inputs and outputs (of the generations just started), two replicas each.  The inputs have names of their own: the
    // one-chain form below reads them on stream2, the separate calls' spatial chain writes "Ks" on stream3.
    double *Ks = c->buf<double>("Ks_pair", (size_t)nxx * 2), *Kt = c->buf<double>("Kt_pair", (size_t)ntt * 2);
    double *Qs = c->buf<double>(gen_name(c, 0, "Qs"), (size_t)nxx * 2), *es = c->buf<double>(gen_name(c, 0, "es"), (size_t)nx * 2);
    double *Qt = c->buf<double>(gen_name(c, 1, "Qt"), (size_t)ntt * 2), *et = c->buf<double>(gen_name(c, 1, "et"), (size_t)nt * 2);
    const FoldView vs = sym_s ? eigh_fold_view(c, 0, sym_s, nx, 2) : FoldView(), vt = sym_t ? eigh_fold_view(c, 1, sym_t, nt, 2) : FoldView();
    c->tl("call start (main)", s);
    // Gram matrices.  Temporal (stream2): replica b = Kt(hp[b]).  Spatial (stream3): replica b = Ks(hp[b]) + jitter[b] I; with
    // equal spatial hyper-parameters -- the usual pair -- the two differ by the diagonal shift only, so the matrix is
    // assembled once and copied (the same GEMM output plus the same diagonal add: the same bits).
    c->tl("T chain start (s2)", s2);
    const bool tfill = temporal_fill_applies(c, sym_t, nt, false);       // (the paired call is refused for host temporal Grams)
    if (tfill) temporal_fill(c, hp, nT, t, nt, *sym_t, status + 1, 2, s2);
    else for (int b = 0; b < nT; ++b) build_kt(c, hp[b], t, nt, t, nt, Kt + b * ntt, s2);
    // the temporal chain is the critical path of the call: it is queued before the host spends its time on the launches of the
    // spatial Gram assembly (status words [1], [3]; one replica when the problem is shared -- decomposition cache on, equal
    // temporal hyper-parameters).  (All four problems in ONE chain was measured slower, 1.38 against 1.18 ms per cfg3 step: with
    // two chains the log-likelihood's spatial projection runs under the end of the temporal one.)
    {
        ProfScope ps(c, "eigh_temporal", 9.0 * (double)nt * nt * nt * nT, s2);
        eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, status + 1, s2, false, nT, 2, -1,
                         tfill ? 2 : 0);
    }
    GP_HIP(hipEventRecord(c->ev_join, s2));
    c->tl("T chain end (s2)", s2);
    c->tl("S chain start (s3)", s3);
    const bool same_ks = hp[0]->R == hp[1]->R && hp[0]->ell_s[0] == hp[1]->ell_s[0] &&
                         (g.dim == 1 || (hp[0]->eps == hp[1]->eps && hp[0]->ell_s[1] == hp[1]->ell_s[1]));
    const bool sfill = spatial_fill_applies(c, sym_s, nx);
    if (sfill) {
        // the fill folds Ks and adds each replica's jitter to the folded diagonals: one assembly, no copy, no diagonal pass
        if (same_ks) build_kphi(c, g, hp[0]->R, hp[0]->eps, hp[0]->ell_s, nullptr, 0, 0.0, Ks, s3, "ks_");
        else for (int b = 0; b < 2; ++b) build_kphi(c, g, hp[b]->R, hp[b]->eps, hp[b]->ell_s, nullptr, 0, 0.0, Ks + b * nxx, s3, "ks_");
        spatial_fill(c, Ks, nx, same_ks ? 0 : nxx, 2, jitter, *sym_s, status, 2, s3);
    } else if (same_ks) {
        const int lo = jitter[0] == 0.0 ? 0 : 1, hi = 1 - lo;           // assemble the one without a shift (if any) first
        build_kphi(c, g, hp[lo]->R, hp[lo]->eps, hp[lo]->ell_s, nullptr, 0, 0.0, Ks + lo * nxx, s3, "ks_");
        GP_HIP(hipMemcpyAsync(Ks + hi * nxx, Ks + lo * nxx, (size_t)nxx * sizeof(double), hipMemcpyDeviceToDevice, s3));
        if (jitter[lo] != 0.0) k_add_diag(c, Ks + lo * nxx, nx, jitter[lo], s3);
        if (jitter[hi] != 0.0) k_add_diag(c, Ks + hi * nxx, nx, jitter[hi], s3);
    } else {
        for (int b = 0; b < 2; ++b) build_kphi(c, g, hp[b]->R, hp[b]->eps, hp[b]->ell_s, nullptr, 0, jitter[b], Ks + b * nxx, s3, "ks_");
    }
    // two replicas of the spatial problem on stream3 (status words [0], [2])
    {
        ProfScope ps(c, "eigh_spatial", 9.0 * (double)nx * nx * nx * 2, s3);
        eigh_pair_device(c, Ks, nx, es, Qs, sym_s, nullptr, 0, nullptr, nullptr, nullptr, status, s3, false, 2, 2, -1,
                         sfill ? 1 : 0);
    }
    GP_HIP(hipEventRecord(c->ev_sjoin, s3));
    c->tl("S chain end (s3)", s3);
    c->decomp_gen[0] = c->decomp_gen[1] = -1;          // replicas are not what the separate calls' cache looks for
    const double *d_sig[2] = {c->upload_cached<double>("sig2n", hp[0]->sig2n, 1), c->upload_cached<double>("sig2n_pair", hp[1]->sig2n, 1)};
    for (int b = 0; b < 2; ++b) {
        const int bt = nT == 2 ? b : 0;
        EigState &e = out.e[b];
        e.Qs = Qs + b * nxx; e.es = es + (long)b * nx; e.Qt = Qt + bt * ntt; e.et = et + (long)bt * nt;
        e.D = c->buf<double>("D", (size_t)nx * nt);
        e.Dinv = c->buf<double>("Dinv", (size_t)nx * nt);
        e.scal = scal;
        e.status = status;
        e.pending = true;
        e.wait_temporal = e.wait_spatial = true;
        e.d_sig = d_sig[b];
        e.nsig = 1;
        FoldMode &fm = out.fm[b];
        fm = fold_mode(c, hp[b]);                       // replica 0 of the generations just started ...
        if (fm.fs.on) { fm.fs.w += (long)b * vs.sw; fm.fs.U += (long)b * vs.sU; }       // ... moved to replica b
        else { fm.fs.w += (long)b * nx; fm.fs.U += b * nxx; }
        if (fm.ft.on) { fm.ft.w += (long)bt * vt.sw; fm.ft.U += (long)bt * vt.sU; }
        else { fm.ft.w += (long)bt * nt; fm.ft.U += bt * ntt; }
    }
}

// Collect the status words of an asynchronous predict (see gpcsd_ctx::async_pending): drains the streams.
static int drain_async(gpcsd_ctx *c) {
    if (!c->async_pending) return 0;
    c->async_pending = false;
    int *st = reinterpret_cast<int *>(c->buf<double>("scal_status", 64 + 2) + 64);
    GP_HIP(hipStreamSynchronize(c->stream2));
    GP_HIP(hipStreamSynchronize(c->stream3));
    return finish_status(c, st);          // downloads + synchronises; the words are cleared by the next call's front half
}

extern "C" int gpcsd_predict_resident(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar,
                                      int ntstar, int type, int want_lists) {
    GP_API_BEGIN(c)
    return predict_impl(c, hp, z, nz, tstar, ntstar, type, want_lists != 0, /*async=*/true);
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_predict_async(gpcsd_ctx *c, const gpcsd_hparams *hp_ll, const gpcsd_hparams *hp_pr, const double *z, int nz,
                                          const double *tstar, int ntstar, int type, int want_lists) {
    if (c && c->ll_count >= gpcsd_ctx::LL_SLOTS)
        return fail(c, HipError{-3, "loglik_predict_async: too many asynchronous evaluations outstanding (collect with "
                                    "gpcsd_loglik_parts_wait)"});
    GP_API_BEGIN(c)
    GP_REQUIRE(hp_ll && hp_pr, -3, "loglik_predict_async: null hparams");
    GP_REQUIRE(z && tstar && nz > 0 && ntstar > 0, -3, "predict: bad arguments");
    GP_REQUIRE(type >= 1 && type <= 3, -3, "predict: type must be CSD(1), LFP(2) or BOTH(3)");
    GP_REQUIRE(c->nt > 0 && ntstar == c->nt, -22,
               "predict: len(t)=%d must equal the training nt=%d (the reference's reshape raises ValueError, gpcsd1d.py:279)",
               ntstar, c->nt);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    // the paired front half serves the folded-basis tails only; anything else is the two calls one after the other
    bool pair = two_stream_front() && c->prof_mode != 1 && !uses_host_kt(hp_ll) && !uses_host_kt(hp_pr) &&
                hp_ll->n_sig2n == 1 && hp_pr->n_sig2n == 1;
    FoldMode fm0;
    SymDev sz;
    if (pair) {
        fm0 = fold_mode(c, hp_ll);
        pair = fm0.on && fold_mode(c, hp_pr).on &&
               (!fm0.ft.on || ((int)c->time_host.size() == c->nt &&
                               memcmp(c->time_host.data(), tstar, (size_t)ntstar * sizeof(double)) == 0));
    }
    if (pair) {
        sz = fm0.fs.on ? site_symmetry(c, z, nz, c->dim) : identity_sym(c, nz);
        pair = sz.ns > 0 && sz.ns + sz.na == nz;
    }
    if (!pair) {
        const int rc = loglik_parts_impl(c, hp_ll, nullptr, true);
        if (rc != 0) return rc;
        return predict_impl(c, hp_pr, z, nz, tstar, ntstar, type, want_lists != 0, true);
    }
    GP_REQUIRE(c->time_nt == c->nt, -4, "time grid has %d points but lfp has nt=%d", c->time_nt, c->nt);
    GP_REQUIRE(resident_geo(c).nx == c->nx, -4, "geometry has %d electrodes but lfp has nx=%d", resident_geo(c).nx, c->nx);
    check_hp(c, hp_ll, c->nx);
    check_hp(c, hp_pr, c->nx);
    const gpcsd_hparams *hps[2] = {hp_ll, hp_pr};
    const double jit[2] = {hp_ll->jitter, 0.0};          // no jitter in predict (gpcsd1d.py:258)
    PairFront pf;
    front_half_pair(c, hps, jit, pf);
    const double *Yf = folded_lfp(c, pf.fm[1]);
    double *dzf = c->upload_cached<double>("pred_z", z, (size_t)nz * c->dim);
    double *dtf = c->upload_cached<double>("pred_tstar", tstar, ntstar);
    const std::function<void()> ll_tail = [&]() {
        double *Wll = c->buf<double>("proj_W_ll", (size_t)c->nx * c->ntrials * c->nt);
        const bool batched = loglik_fold_tail(c, pf.e[0], pf.fm[0], Yf, Wll);
        (void)finish_loglik_async(c, pf.e[0], !batched);
    };
    return predict_fold(c, hp_pr, pf.e[1], pf.fm[1], Yf, sz, dzf, nz, dtf, type, want_lists != 0, true, &ll_tail);
    GP_API_END(c)
}

extern "C" int gpcsd_fetch(gpcsd_ctx *c, const char *name, double *host, long count) {
    GP_API_BEGIN(c)
    GP_REQUIRE(name && host && count > 0, -3, "fetch: bad arguments");
    auto it = c->bufs.find(name);
    GP_REQUIRE(it != c->bufs.end() && it->second.p, -2, "fetch: no device buffer named '%s'", name);
    GP_REQUIRE((size_t)count * sizeof(double) <= it->second.bytes, -3, "fetch: '%s' holds %zu bytes, asked for %ld doubles", name,
               it->second.bytes, count);
    c->download(host, it->second.p, (size_t)count * sizeof(double));
    c->sync();
    return drain_async(c);                // a numerical failure of a preceding asynchronous predict surfaces here
    GP_API_END(c)
}

extern "C" int gpcsd_predict(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                             int type, double *csd_list, double *csd, double *lfp_list, double *lfp) {
    GP_API_BEGIN(c)
    const bool want_lists = (csd_list != nullptr) || (lfp_list != nullptr);
    int rc = predict_impl(c, hp, z, nz, tstar, ntstar, type, want_lists);
    if (rc < 0) return rc;
    const size_t out_elems = (size_t)nz * ntstar * c->ntrials;
    const int C = hp->n_temporal;
    if ((type & 1) && csd) c->download(csd, c->bufs["pred_out_csd"].p, out_elems * sizeof(double));
    if ((type & 1) && csd_list) c->download(csd_list, c->bufs["pred_out_csd_list"].p, out_elems * C * sizeof(double));
    if ((type & 2) && lfp) c->download(lfp, c->bufs["pred_out_lfp"].p, out_elems * sizeof(double));
    if ((type & 2) && lfp_list) c->download(lfp_list, c->bufs["pred_out_lfp_list"].p, out_elems * C * sizeof(double));
    c->sync();
    return rc;
    GP_API_END(c)
}

extern "C" int gpcsd_sample_prior(gpcsd_ctx *c, const gpcsd_hparams *hp, int which, const double *normals, int ntrials,
                                  double *out) {
    GP_API_BEGIN(c)
    const Geo g = resident_geo(c);
    GP_REQUIRE(normals && out && ntrials > 0, -3, "sample_prior: bad arguments");
    GP_REQUIRE(which == GPCSD_PRED_CSD || which == GPCSD_PRED_LFP, -3, "sample_prior: which must be CSD(1) or LFP(2)");
    GP_REQUIRE(c->time_nt > 0, -4, "time grid not set");
    check_hp(c, hp, g.nx);
    const int nx = g.nx, nt = c->time_nt, R = ntrials;
    const long RT = (long)R * nt;
    hipStream_t s = c->stream;
    double *Ks = c->buf<double>("Ks", (size_t)nx * nx);
    double *Kt = c->buf<double>("Kt", (size_t)nt * nt);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), s));
    const double *t = (const double *)c->bufs["time_t"].p;
    if (which == GPCSD_PRED_CSD) {
        build_ks_csd(c, g, hp->ell_s, Ks, s);                                      // gpcsd1d.py:298
        k_add_diag(c, Ks, nx, hp->jitter, s);
    } else {
        build_kphi(c, g, hp->R, hp->eps, hp->ell_s, nullptr, 0, hp->jitter, Ks, s);   // gpcsd2d.py:346-347
    }
    if (uses_host_kt(hp)) {
        GP_REQUIRE(c->host_kt_nt == nt && (int)c->host_kt.size() == nt * nt, -3, "host temporal Gram does not match nt=%d", nt);
        GP_HIP(hipMemcpyAsync(Kt, c->host_kt.data(), (size_t)nt * nt * sizeof(double), hipMemcpyHostToDevice, s));
    } else {
        build_kt(c, hp, t, nt, t, nt, Kt, s);
    }
    potrf_device(c, Kt, nt, st, s);                                                 // Lt
    potrf_device(c, Ks, nx, st, s);                                                 // Ls
    double *stage = c->upload<double>("sp_stage", normals, (size_t)nx * RT);
    double *Z = c->buf<double>("sp_Z", (size_t)nx * RT);
    double *T1 = c->buf<double>("sp_T1", (size_t)nx * RT);
    k_swap_last2(c, stage, Z, nx, nt, R, s);                                        // (x,t,r) -> (x,r,t)
    GemmDesc g1;                          // T1[x'][(r,t)] = sum_x Ls[x'][x] Z[x][(r,t)]
    g1.M = nx; g1.N = (int)RT; g1.K = nx;
    g1.A = Ks; g1.lda = nx; g1.B = Z; g1.ldb = RT; g1.C = T1; g1.ldc = RT;
    g1.prof_name = "gemm_sample_spatial";
    gemm_f64(c, g1, s);
    GemmDesc g2;                          // out[(x',r)][t'] = sum_t T1[(x',r)][t] Lt[t'][t]
    g2.M = nx * R; g2.N = nt; g2.K = nt;
    g2.A = T1; g2.lda = nt; g2.B = Kt; g2.ldb = nt; g2.transB = true; g2.C = Z; g2.ldc = nt;
    g2.prof_name = "gemm_sample_temporal";
    gemm_f64(c, g2, s);
    k_swap_last2(c, Z, stage, nx, R, nt, s);                                        // (x,r,t) -> (x,t,r)
    c->download(out, stage, (size_t)nx * RT * sizeof(double));
    return finish_status(c, st);
    GP_API_END(c)
}

// ---- log-likelihood + analytic gradient for B hyper-parameter sets in ONE chain of launches ---------------------------------
// fit() restarts are independent optimiser chains (gpcsd1d.py:193-220) whose evaluations are latency-bound: ~100 dependent
// launches in which the longest kernel occupies one workgroup per eigenproblem.  B sets evaluated together share every launch:
// the Gram builders and derivative kernels take the set index as a grid dimension (scalars from a device table of
// hyper-parameters), the eigensolver runs B replicas of each problem class, every GEMM gets an outer batch level.  Each set
// executes exactly the arithmetic of an evaluation on its own (same kernels, same tile configurations, same reduction
// order), so its results do not depend on B.
static HpDev hp_image(const gpcsd_hparams *hp) {
    HpDev h{};
    h.R = hp->R; h.eps = hp->eps; h.ell_s[0] = hp->ell_s[0]; h.ell_s[1] = hp->ell_s[1];
    h.ncomp = hp->n_temporal;
    for (int i = 0; i < hp->n_temporal; ++i) {
        h.kind[i] = hp->kind[i];
        h.ell_t[i] = hp->ell_t[i];
        h.sigma2_t[i] = hp->sigma2_t[i];
    }
    h.sig2n = hp->sig2n[0];
    h.jitter = hp->jitter;
    return h;
}

// out2: (B, 2) = (sum log D, quad) per set; grad: (B, ngrad); status: (B) -- 0 ok, > 0 numerical failure of that set alone.
static int loglik_grad_impl(gpcsd_ctx *c, const gpcsd_hparams *hps, int B, double *out2, double *grad, int ngrad, int *status) {
    GP_REQUIRE(out2 && grad && hps && B >= 1, -3, "loglik_grad: null argument");
    // every argument check comes before any work is queued (the front half launches on two streams)
    const Geo g = resident_geo(c);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    GP_REQUIRE(c->time_nt == c->nt, -4, "time grid has %d points but lfp has nt=%d", c->time_nt, c->nt);
    GP_REQUIRE(g.nx == c->nx, -4, "geometry has %d electrodes but lfp has nx=%d", g.nx, c->nx);
    const int nx = c->nx, nt = c->nt, R = c->ntrials, C = hps[0].n_temporal, G = g.G();
    const int nsig = hps[0].n_sig2n;
    for (int b = 0; b < B; ++b) {
        check_hp(c, &hps[b], nx);
        // user-defined temporal covariances: the caller supplies d Kt / d theta_k (gpcsd_set_host_temporal_dgram), one set at a time
        GP_REQUIRE(!uses_host_kt(&hps[b]) || (B == 1 && c->host_kt_on && c->host_kt_nt == nt && c->host_dkt_n == 2 * C &&
                                              c->host_dkt.size() == (size_t)2 * C * nt * nt), -3,
                   "loglik_grad: user-defined temporal covariances need their Gram matrix and the %d derivative matrices "
                   "d Kt / d (ell_c, sigma2_c) (gpcsd_set_host_temporal_gram + gpcsd_set_host_temporal_dgram), one set per call", 2 * C);
        GP_REQUIRE(hps[b].n_temporal == C && hps[b].n_sig2n == nsig, -3,
                   "loglik_grad_batch: every hyper-parameter set must have the same number of temporal components and noise entries");
        for (int i = 0; i < C; ++i)
            GP_REQUIRE(hps[b].kind[i] == hps[0].kind[i], -3, "loglik_grad_batch: temporal kernel kinds differ between sets");
    }
    // scalar sig2n: one trailing entry; per-electrode list (indexed by eigen-row like the reference's D): nx entries
    GP_REQUIRE(nsig == 1 || nsig == nx, -3, "loglik_grad: sig2n must be a scalar or a list of nx=%d values (got %d)", nx, nsig);
    GP_REQUIRE(nsig == 1 || B == 1, -3, "loglik_grad_batch: per-electrode noise lists are evaluated one set at a time");
    const int nhead = 1 + g.dim + 2 * C;
    GP_REQUIRE(ngrad == nhead + nsig, -3, "loglik_grad: ngrad=%d, expected %d", ngrad, nhead + nsig);
    const long RT = (long)R * nt, nxx = (long)nx * nx, ntt = (long)nt * nt, nD = (long)nx * nt, nxRT = (long)nx * RT;
    const long nxG = (long)nx * G, GG = (long)G * G;
    hipStream_t s = c->stream, s2 = c->stream2;

    // ---- device table of the hyper-parameter sets
    std::vector<HpDev> himg(B);
    for (int b = 0; b < B; ++b) himg[b] = hp_image(&hps[b]);
    const HpDev *tab = c->upload_cached<HpDev>("b_hp_tab", himg.data(), B);
    const double *d_siglist = nsig > 1 ? c->upload_cached<double>("sig2n", hps[0].sig2n, nsig) : nullptr;

    double *Ks = c->buf<double>("b_Ks", nxx * B), *Kt = c->buf<double>("b_Kt", ntt * B);
    double *Qs = c->buf<double>("b_Qs", nxx * B), *Qt = c->buf<double>("b_Qt", ntt * B);
    double *es = c->buf<double>("b_es", (size_t)nx * B), *et = c->buf<double>("b_et", (size_t)nt * B);
    double *D = c->buf<double>("b_D", nD * B), *Dinv = c->buf<double>("b_Dinv", nD * B);
    constexpr int NS = 8;                                     // scalars per set: sumlog, quad, sum B^2, sum 1/D
    double *scal = c->buf<double>("b_scal", (size_t)NS * B);
    int *st = c->buf<int>("b_status", (size_t)2 * B);        // [0, B): spatial chains, [B, 2B): temporal chains
    double *A = c->buf<double>("b_ks_A", nxG * B), *Kgl = c->buf<double>("b_ks_Kgl", GG * B), *T = c->buf<double>("b_ks_T", nxG * B);
    double *W = c->buf<double>("b_W", nxRT * B), *Bm = c->buf<double>("b_Bm", nxRT * B);
    double *Bet = c->buf<double>("b_Bet", nxRT * B), *Bes = c->buf<double>("b_Bes", nxRT * B);
    double *gdev = c->buf<double>("b_grad_out", (size_t)64 * B);
    const double *t = (const double *)c->bufs["time_t"].p;
    const bool host_kt = uses_host_kt(&hps[0]);
    // (a caller-supplied Gram need not commute with the reflection of the time grid: that side is not folded, cf. front_half)
    const SymDev *sym_s = c->sym_s.ns > 0 ? &c->sym_s : nullptr, *sym_t = (c->sym_t.ns > 0 && !host_kt) ? &c->sym_t : nullptr;
    // Folded basis (see FoldMode): with a scalar noise variance the whole evaluation runs on the half-size eigenvector blocks
    // of the symmetry-folded eigensolver -- projections, the Ghat_s / Ghat_t sums and the back-rotations are each two
    // half-size products.  The cross-parity blocks of Ghat are never needed: dKs and dKt commute with the reflections, so
    // <G, dK> only sees the parity-diagonal blocks.  Half the GEMM flops of the full-size path below.
    // ---- front half: temporal chain on stream2 (queued first: the critical path), spatial chain on the main stream.  Both
    // read the hyper-parameter table uploaded above and report into the status words cleared here: they start behind the
    // main stream's current position (this call returns values, so nothing of it outlives it anyway).
    GP_HIP(hipMemsetAsync(st, 0, (size_t)2 * B * sizeof(int), s));
    begin_generation(c, 1, s2, true);
    begin_generation(c, 0, s, true);
    const FoldMode fm = fold_mode(c, &hps[0]);
    const bool fold = fm.on;
    const double *Yf = fold ? folded_lfp(c, fm) : nullptr;
    if (host_kt) GP_HIP(hipMemcpyAsync(Kt, c->host_kt.data(), (size_t)ntt * sizeof(double), hipMemcpyHostToDevice, s2));
    else k_temporal_gram(c, C, nullptr, nullptr, nullptr, t, nt, t, nt, Kt, s2, tab, B, ntt);
    {
        ProfScope ps(c, "eigh_temporal", 9.0 * (double)nt * nt * nt * B, s2);
        eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, st + B, s2, !fold, B, 1);
    }
    GP_HIP(hipEventRecord(c->ev_join, s2));
    // Ks_b = A_b Kgl_b A_b^T + jitter_b I                     covariances.py:74-96 / :204-232
    if (g.dim == 1) {
        k_fwd_weights_1d(c, g.x, nx, g.gx1, g.gw1, g.ngl1, 0.0, A, s, tab, B, nxG);
        k_se_1d(c, g.gx1, G, g.gx1, G, 0.0, Kgl, s, tab, B, GG);
    } else {
        k_fwd_weights_2d(c, g.x, nx, g.gx1, g.gw1, g.ngl1, g.gx2, g.gw2, g.ngl2, 0.0, 0.0, A, s, tab, B, nxG);
        k_se_2d(c, g.gx1, g.gx2, G, g.ngl2, g.gx1, g.gx2, G, g.ngl2, 0.0, 0.0, Kgl, s, tab, B, GG);
    }
    {
        GemmDesc d1;                                   // T = A Kgl
        d1.M = nx; d1.N = G; d1.K = G;
        d1.A = A; d1.lda = G; d1.B = Kgl; d1.ldb = G; d1.C = T; d1.ldc = G;
        d1.batch2 = B; d1.sA2 = nxG; d1.sB2 = GG; d1.sC2 = nxG;
        d1.prof_name = "gemm_Ks_AKgl";
        gemm_f64(c, d1, s);
        GemmDesc d2;                                   // Ks = T A^T
        d2.M = nx; d2.N = nx; d2.K = G;
        d2.A = T; d2.lda = G; d2.B = A; d2.ldb = G; d2.transB = true; d2.C = Ks; d2.ldc = nx;
        d2.batch2 = B; d2.sA2 = nxG; d2.sB2 = nxG; d2.sC2 = nxx;
        d2.prof_name = "gemm_Ks_TAt";
        gemm_f64(c, d2, s);
        k_add_diag(c, Ks, nx, 0.0, s, tab, B, nxx);
    }
    {
        ProfScope ps(c, "eigh_spatial", 9.0 * (double)nx * nx * nx * B, s);
        eigh_pair_device(c, Ks, nx, es, Qs, sym_s, nullptr, 0, nullptr, nullptr, nullptr, st, s, !fold, B, 1);
    }
    double *av = c->buf<double>("b_grad_a", (size_t)nx * B), *bv = c->buf<double>("b_grad_b", (size_t)nt * B);
    double *Gs = c->buf<double>("b_grad_Gs", nxx * B), *Gt = c->buf<double>("b_grad_Gt", ntt * B);
    const long nmx = (long)std::max(nx, nt) * std::max(nx, nt);
    double *T1 = c->buf<double>("b_grad_T1", (size_t)nmx * B);
    const int CH = 512;                   // row chunk of the Ghat_t sums
    bool quad_in_two = false;             // the quadratic form came out as two partial sums (parity blocks of unequal shape)
    if (fold) {
        ++c->fold_gemm_calls;
        // a side that is not folded takes part as one "symmetric" block of full size (identity fold, U = Q, w = eigenvalues)
        struct Side {
            int n, ns, na;
            const double *U, *w;
            long sU, sw;
            SymDev sym;
        } S_, T_;
        auto side = [&](int slot, const FoldView &fv1, const SymDev &sym, int n, const double *Q, const double *ev) {
            Side sd;
            sd.n = n;
            if (fv1.on) {
                const FoldView fv = eigh_fold_view(c, slot, slot ? &c->sym_t : &c->sym_s, n, B);
                sd.ns = fv.ns; sd.na = fv.na; sd.U = fv.U; sd.w = fv.w; sd.sU = fv.sU; sd.sw = fv.sw;
            } else {
                sd.ns = n; sd.na = 0; sd.U = Q; sd.w = ev; sd.sU = (long)n * n; sd.sw = n;
            }
            sd.sym = sym;
            return sd;
        };
        S_ = side(0, fm.fs, fm.sym_s, nx, Qs, es);
        T_ = side(1, fm.ft, fm.sym_t, nt, Qt, et);
        const long sUs = (long)S_.ns * S_.ns + (long)S_.na * S_.na, sUt = (long)T_.ns * T_.ns + (long)T_.na * T_.na;
        // W~_b = diag(U_b)^T Y~ : the data is folded once per geometry and shared by all sets
        for (int p = 0; p < 2; ++p) {
            const int np = p ? S_.na : S_.ns;
            const long r0 = p ? S_.ns : 0;
            if (np == 0) continue;
            GemmDesc gw;
            gw.M = np; gw.N = (int)RT; gw.K = np;
            gw.A = S_.U + (p ? (long)S_.ns * S_.ns : 0); gw.lda = np; gw.transA = true;
            gw.B = Yf + r0 * RT; gw.ldb = RT; gw.C = W + r0 * RT; gw.ldc = RT;
            gw.batch2 = B; gw.sA2 = S_.sU; gw.sB2 = 0; gw.sC2 = nxRT;
            gw.prof_name = "gemm_proj_spatial";
            gemm_f64(c, gw, s);
        }
        GP_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
        // D~_b = ws_b (x) wt_b + sig2n_b in fold order, sum log D_b -> scal[b][0]
        k_build_D(c, S_.w, nx, T_.w, nt, nullptr, 1, D, Dinv, scal, s, tab, B, NS);
        {   // alpha~ = W~ V (per temporal parity block);  B~ = alpha~ / D~, B~ wt, B~ ws;  sums of alpha~ B~ and B~^2
            GemmDesc gq[2];
            for (int q = 0; q < 2; ++q) {
                const int nq = q ? T_.na : T_.ns, c0 = q ? T_.ns : 0;
                gq[q].M = nx * R; gq[q].N = nq; gq[q].K = nq;
                gq[q].A = W + c0; gq[q].lda = nt; gq[q].B = T_.U + (q ? (long)T_.ns * T_.ns : 0); gq[q].ldb = nq;
                gq[q].C = Bm + c0; gq[q].ldc = nt; gq[q].C2 = Bet + c0; gq[q].C3 = Bes + c0;
                gq[q].epi = EPI_GRAD; gq[q].D = Dinv + c0; gq[q].rdiv = R; gq[q].ldd = nt;
                gq[q].colscale = T_.w + c0; gq[q].rowscale = S_.w;
                gq[q].quad_out = scal + 1 + 3 * q;        // scal[b][1], [2] (first block or both), scal[b][4], [5] (second block)
                gq[q].batch2 = B; gq[q].sA2 = nxRT; gq[q].sB2 = T_.sU; gq[q].sC2 = nxRT; gq[q].sD2 = nD; gq[q].sColscale2 = nt;
                gq[q].sRowscale2 = nx; gq[q].sQuad2 = NS;
                gq[q].prof_name = "gemm_grad_temporal";
            }
            if (T_.na > 0 && T_.na == T_.ns) {           // equal parity blocks: one launch, one sum over both
                gq[0].batch = 2;
                gq[0].sA = gq[1].A - gq[0].A; gq[0].sB = gq[1].B - gq[0].B; gq[0].sC = gq[1].C - gq[0].C;
                gq[0].sD = gq[1].D - gq[0].D; gq[0].sColscale = gq[1].colscale - gq[0].colscale;
                gemm_f64(c, gq[0], s);
            } else {
                gemm_f64(c, gq[0], s);
                if (T_.na > 0) {
                    gemm_f64(c, gq[1], s);
                    quad_in_two = true;
                }
            }
        }
        k_D_sums(c, D, S_.w, T_.w, nx, nt, av, bv, scal + 3, s, B, NS);   // a, b in fold order; scal[b][3] = sum 1/D
        // Ghat_s~ parity blocks: 1/2 sum_r (B~ wt)[p rows] B~[p rows]^T - R/2 diag(a[p rows])
        const long sCs = (long)R * sUs;
        double *Cs = c->buf<double>("b_grad_Cs", (size_t)sCs * B);
        double *Ghs = c->buf<double>("b_grad_Ghs", (size_t)sUs * B);
        for (int p = 0; p < 2; ++p) {
            const int np = p ? S_.na : S_.ns;
            const long r0 = p ? S_.ns : 0, o_in = p ? (long)R * S_.ns * S_.ns : 0, o_out = p ? (long)S_.ns * S_.ns : 0;
            if (np == 0) continue;
            GemmDesc gs;
            gs.M = np; gs.N = np; gs.K = nt;
            gs.A = Bet + r0 * RT; gs.lda = RT; gs.B = Bm + r0 * RT; gs.ldb = RT; gs.transB = true; gs.C = Cs + o_in; gs.ldc = np;
            gs.batch = R; gs.sA = nt; gs.sB = nt; gs.sC = (long)np * np;
            gs.batch2 = B; gs.sA2 = nxRT; gs.sB2 = nxRT; gs.sC2 = sCs;
            // the tile configuration must not depend on B (a set has to run the same tiles alone or in a batch): these
            // half-size products have few tiles per set, which the automatic choice would read as "latency-bound"
            if (np >= 64) gs.cfg = 3;
            gs.prof_name = "gemm_grad_Gs";
            gemm_f64(c, gs, s);
            k_batch_reduce(c, Cs + o_in, R, (long)np * np, np, 0.5, av + r0, -0.5 * R, Ghs + o_out, s, B, sCs, nx, sUs);
        }
        // Ghat_t~ parity blocks: 1/2 sum_{(x,r)} (B~ ws)[:, q]^T B~[:, q] - R/2 diag(b[q block])   (row chunks, then a fixed-order sum)
        const long rows = (long)nx * R;
        const int nfull = (int)(rows / CH), rem = (int)(rows % CH), nchunk = nfull + (rem > 0 ? 1 : 0);
        const long sCt = (long)nchunk * sUt;
        double *Ct = c->buf<double>("b_grad_Ct", (size_t)sCt * B);
        double *Ght = c->buf<double>("b_grad_Ght", (size_t)sUt * B);
        for (int q = 0; q < 2; ++q) {
            const int nq = q ? T_.na : T_.ns, c0 = q ? T_.ns : 0;
            const long o_in = q ? (long)nchunk * T_.ns * T_.ns : 0, o_out = q ? (long)T_.ns * T_.ns : 0, nqq = (long)nq * nq;
            if (nq == 0) continue;
            if (nfull > 0) {
                GemmDesc gt;
                gt.M = nq; gt.N = nq; gt.K = CH;
                gt.A = Bes + c0; gt.lda = nt; gt.transA = true; gt.B = Bm + c0; gt.ldb = nt; gt.C = Ct + o_in; gt.ldc = nq;
                gt.batch = nfull; gt.sA = (long)CH * nt; gt.sB = (long)CH * nt; gt.sC = nqq;
                gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
                if (nq >= 64) gt.cfg = 3;
                gt.prof_name = "gemm_grad_Gt";
                gemm_f64(c, gt, s);
            }
            if (rem > 0) {
                GemmDesc gt;
                gt.M = nq; gt.N = nq; gt.K = rem;
                if (nq >= 64) gt.cfg = 3;
                gt.A = Bes + (long)nfull * CH * nt + c0; gt.lda = nt; gt.transA = true; gt.B = Bm + (long)nfull * CH * nt + c0; gt.ldb = nt;
                gt.C = Ct + o_in + (long)nfull * nqq; gt.ldc = nq;
                gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
                gt.prof_name = "gemm_grad_Gt";
                gemm_f64(c, gt, s);
            }
            k_batch_reduce(c, Ct + o_in, nchunk, nqq, nq, 0.5, bv + c0, -0.5 * R, Ght + o_out, s, B, sCt, nt, sUt);
        }
        // back to the original bases, block by block: G~_pp = U_p Ghat_pp U_p^T, then G = F^T diag(G~_ss, G~_aa) F
        double *Gsf = c->buf<double>("b_grad_Gsf", (size_t)sUs * B), *Gtf = c->buf<double>("b_grad_Gtf", (size_t)sUt * B);
        auto sandwich_blocks = [&](const Side &sd, const double *H, long sH, double *outf) {
            for (int p = 0; p < 2; ++p) {
                const int np = p ? sd.na : sd.ns;
                const long o = p ? (long)sd.ns * sd.ns : 0;
                if (np == 0) continue;
                GemmDesc a;
                a.M = np; a.N = np; a.K = np; a.A = sd.U + o; a.lda = np; a.B = H + o; a.ldb = np; a.C = T1; a.ldc = np;
                a.batch2 = B; a.sA2 = sd.sU; a.sB2 = sH; a.sC2 = nmx;
                a.prof_name = "gemm_grad_sandwich";
                gemm_f64(c, a, s);
                GemmDesc bq;
                bq.M = np; bq.N = np; bq.K = np; bq.A = T1; bq.lda = np; bq.B = sd.U + o; bq.ldb = np; bq.transB = true;
                bq.C = outf + o; bq.ldc = np;
                bq.batch2 = B; bq.sA2 = nmx; bq.sB2 = sd.sU; bq.sC2 = sH;
                bq.prof_name = "gemm_grad_sandwich";
                gemm_f64(c, bq, s);
            }
        };
        sandwich_blocks(S_, Ghs, sUs, Gsf);
        sandwich_blocks(T_, Ght, sUt, Gtf);
        k_sym_unfold_mat(c, Gsf, sUs, S_.sym, nx, Gs, s, B);
        k_sym_unfold_mat(c, Gtf, sUt, T_.sym, nt, Gt, s, B);
    } else {
        GemmDesc g1;                          // W_b = Qs_b^T Y          (gpcsd1d.py:125 inner dot; the data is shared)
        g1.M = nx; g1.N = (int)RT; g1.K = nx;
        g1.A = Qs; g1.lda = nx; g1.transA = true; g1.B = c->d_lfp; g1.ldb = RT; g1.C = W; g1.ldc = RT;
        g1.batch2 = B; g1.sA2 = nxx; g1.sB2 = 0; g1.sC2 = nxRT;
        g1.prof_name = "gemm_proj_spatial";
        gemm_f64(c, g1, s);
        GP_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
        // D_b = es_b (x) et_b + sig2n_b, sum log D_b -> scal[b][0]
        if (nsig == 1) k_build_D(c, es, nx, et, nt, nullptr, 1, D, Dinv, scal, s, tab, B, NS);
        else k_build_D(c, es, nx, et, nt, d_siglist, nsig, D, Dinv, scal, s);
        GemmDesc g2;                          // alpha = W Qt;  B = alpha / D, B*et, B*es;  sum alpha*B, sum B^2
        g2.M = nx * R; g2.N = nt; g2.K = nt;
        g2.A = W; g2.lda = nt; g2.B = Qt; g2.ldb = nt; g2.C = Bm; g2.ldc = nt; g2.C2 = Bet; g2.C3 = Bes;
        g2.epi = EPI_GRAD; g2.D = Dinv; g2.rdiv = R; g2.ldd = nt; g2.colscale = et; g2.rowscale = es;
        g2.quad_out = scal + 1;               // scal[b][1] = quad, scal[b][2] = sum B^2
        g2.batch2 = B; g2.sA2 = nxRT; g2.sB2 = ntt; g2.sC2 = nxRT; g2.sD2 = nD; g2.sColscale2 = nt; g2.sRowscale2 = nx; g2.sQuad2 = NS;
        g2.prof_name = "gemm_grad_temporal";
        gemm_f64(c, g2, s);
        k_D_sums(c, D, es, et, nx, nt, av, bv, scal + 3, s, B, NS);           // scal[b][3] = sum 1/D

        // Ghat_s = 1/2 sum_r (B_r et) B_r^T - R/2 diag(a)      (one GEMM per trial, batched; then a fixed-order sum)
        const long sCs = (long)R * nxx;
        double *Cs = c->buf<double>("b_grad_Cs", (size_t)sCs * B);
        GemmDesc gs;
        gs.M = nx; gs.N = nx; gs.K = nt;
        gs.A = Bet; gs.lda = RT; gs.B = Bm; gs.ldb = RT; gs.transB = true; gs.C = Cs; gs.ldc = nx;
        gs.batch = R; gs.sA = nt; gs.sB = nt; gs.sC = nxx;
        gs.batch2 = B; gs.sA2 = nxRT; gs.sB2 = nxRT; gs.sC2 = sCs;
        gs.prof_name = "gemm_grad_Gs";
        gemm_f64(c, gs, s);
        double *Ghs = c->buf<double>("b_grad_Ghs", nxx * B);
        k_batch_reduce(c, Cs, R, nxx, nx, 0.5, av, -0.5 * R, Ghs, s, B, sCs);
        if (nsig > 1) {
            // noise tied to the eigen-index: eigenvector-rotation term, S = sum_r B_r B_r^T (see grad.hip)
            GemmDesc g3 = gs;
            g3.A = Bm;
            g3.prof_name = "gemm_grad_BBt";
            gemm_f64(c, g3, s);
            double *Ssum = c->buf<double>("grad_Ssum", (size_t)nxx);
            double *zero = c->buf<double>("grad_zero", nx);
            k_fill(c, zero, nx, 0.0, s);
            k_batch_reduce(c, Cs, R, nxx, nx, 1.0, zero, 0.0, Ssum, s);
            k_siglist_eigvec_term(c, Ghs, Ssum, es, d_siglist, nx, 0.0, s);
        }
        // Ghat_t = 1/2 sum_{(x,r)} (B es)^T B - R/2 diag(b)    (row chunks of 512, batched; remainder separately)
        const long rows = (long)nx * R;
        const int nfull = (int)(rows / CH), rem = (int)(rows % CH);
        const long sCt = (long)(nfull + 1) * ntt;
        double *Ct = c->buf<double>("b_grad_Ct", (size_t)sCt * B);
        if (nfull > 0) {
            GemmDesc gt;
            gt.M = nt; gt.N = nt; gt.K = CH;
            gt.A = Bes; gt.lda = nt; gt.transA = true; gt.B = Bm; gt.ldb = nt; gt.C = Ct; gt.ldc = nt;
            gt.batch = nfull; gt.sA = (long)CH * nt; gt.sB = (long)CH * nt; gt.sC = ntt;
            gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
            gt.prof_name = "gemm_grad_Gt";
            gemm_f64(c, gt, s);
        }
        if (rem > 0) {
            GemmDesc gt;
            gt.M = nt; gt.N = nt; gt.K = rem;
            gt.A = Bes + (long)nfull * CH * nt; gt.lda = nt; gt.transA = true; gt.B = Bm + (long)nfull * CH * nt; gt.ldb = nt;
            gt.C = Ct + (long)nfull * ntt; gt.ldc = nt;
            gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
            gt.prof_name = "gemm_grad_Gt";
            gemm_f64(c, gt, s);
        }
        double *Ght = c->buf<double>("b_grad_Ght", ntt * B);
        k_batch_reduce(c, Ct, nfull + (rem > 0 ? 1 : 0), ntt, nt, 0.5, bv, -0.5 * R, Ght, s, B, sCt);
        // back to the original bases: Gs = Qs Ghat_s Qs^T, Gt = Qt Ghat_t Qt^T
        auto sandwich = [&](const double *Q, const double *H, int n, double *out) {
            const long nn = (long)n * n;
            GemmDesc a;
            a.M = n; a.N = n; a.K = n; a.A = Q; a.lda = n; a.B = H; a.ldb = n; a.C = T1; a.ldc = n;
            a.batch2 = B; a.sA2 = nn; a.sB2 = nn; a.sC2 = nmx;
            a.prof_name = "gemm_grad_sandwich";
            gemm_f64(c, a, s);
            GemmDesc bq;
            bq.M = n; bq.N = n; bq.K = n; bq.A = T1; bq.lda = n; bq.B = Q; bq.ldb = n; bq.transB = true; bq.C = out; bq.ldc = n;
            bq.batch2 = B; bq.sA2 = nmx; bq.sB2 = nn; bq.sC2 = nn;
            bq.prof_name = "gemm_grad_sandwich";
            gemm_f64(c, bq, s);
        };
        sandwich(Qs, Ghs, nx, Gs);
        sandwich(Qt, Ght, nt, Gt);
    }
    // natural-parameter order: [R, ell_s (dim), (ell_t, sigma2_t) per component, sig2n]; 64 slots per set
    if (host_kt) {                        // <Gt, d Kt / d theta_k> with the caller's derivative matrices
        double *dK = c->upload<double>("b_host_dkt", c->host_dkt.data(), (size_t)2 * C * ntt);
        k_frob_inner(c, Gt, dK, ntt, 2 * C, gdev + 1 + g.dim, s);
    } else {
        k_temporal_grad(c, &hps[0], Gt, t, nt, gdev + 1 + g.dim, s, tab, B, 64);
    }
    double *P = c->buf<double>("b_grad_P", nxG * B);
    double *Mg = c->buf<double>("b_grad_M", GG * B);
    GemmDesc gp;                          // P = Gs A
    gp.M = nx; gp.N = G; gp.K = nx; gp.A = Gs; gp.lda = nx; gp.B = A; gp.ldb = G; gp.C = P; gp.ldc = G;
    gp.batch2 = B; gp.sA2 = nxx; gp.sB2 = nxG; gp.sC2 = nxG;
    gp.prof_name = "gemm_grad_GsA";
    gemm_f64(c, gp, s);
    GemmDesc gm;                          // M = A^T P
    gm.M = G; gm.N = G; gm.K = nx; gm.A = A; gm.lda = G; gm.transA = true; gm.B = P; gm.ldb = G; gm.C = Mg; gm.ldc = G;
    gm.batch2 = B; gm.sA2 = nxG; gm.sB2 = nxG; gm.sC2 = GG;
    gm.prof_name = "gemm_grad_AtP";
    gemm_f64(c, gm, s);
    k_kgl_grad(c, Mg, Kgl, g.gx1, g.gx2, G, g.dim == 2 ? g.ngl2 : 0, 0.0, 0.0, gdev + 1, s, tab, B, 64);
    GemmDesc gr;                          // S = Gs T  (T = A Kgl from the forward pass)
    gr.M = nx; gr.N = G; gr.K = nx; gr.A = Gs; gr.lda = nx; gr.B = T; gr.ldb = G; gr.C = P; gr.ldc = G;
    gr.batch2 = B; gr.sA2 = nxx; gr.sB2 = nxG; gr.sC2 = nxG;
    gr.prof_name = "gemm_grad_GsT";
    gemm_f64(c, gr, s);
    k_fwdR_grad(c, P, g.x, nx, g.gx1, g.gw1, g.gx2, g.gw2, G, g.dim == 2 ? g.ngl2 : 0, 0.0, 0.0, gdev, s, tab, B, 64);
    std::vector<double> hb2, hinv;
    if (nsig > 1) {                       // d/d sig2n_x = -R/2 sum_i 1/D_xi + 1/2 sum_{r,i} B_{(x,r),i}^2
        double *b2row = c->buf<double>("grad_b2row", nx);
        k_rowgroup_sumsq(c, Bm, nx, RT, b2row, s);
        hb2.resize(nx);
        hinv.resize(nx);
        c->download(hb2.data(), b2row, nx * sizeof(double));
        c->download(hinv.data(), c->bufs["grad_s1row"].p, nx * sizeof(double));   // written by k_D_sums
    }
    std::vector<double> hs((size_t)NS * B), hg((size_t)64 * B);
    std::vector<int> hst((size_t)2 * B);
    c->download(hs.data(), scal, hs.size() * sizeof(double));
    c->download(hg.data(), gdev, hg.size() * sizeof(double));
    c->download(hst.data(), st, hst.size() * sizeof(int));
    GP_HIP(hipStreamSynchronize(s2));
    c->sync();
    if (c->prof_mode == 1) c->prof_collect();
    int worst = 0;
    for (int b = 0; b < B; ++b) {
        out2[2 * b] = hs[(size_t)NS * b];
        out2[2 * b + 1] = hs[(size_t)NS * b + 1] + (quad_in_two ? hs[(size_t)NS * b + 4] : 0.0);
        double *gb = grad + (size_t)b * ngrad;
        for (int k = 0; k < nhead; ++k) gb[k] = hg[(size_t)64 * b + k];
        if (nsig == 1)
            gb[nhead] = -0.5 * R * hs[(size_t)NS * b + 3] + 0.5 * (hs[(size_t)NS * b + 2] + (quad_in_two ? hs[(size_t)NS * b + 5] : 0.0));
        else
            for (int x = 0; x < nx; ++x) gb[nhead + x] = -0.5 * R * hinv[x] + 0.5 * hb2[x];
        int stb = hst[b] != 0 ? hst[b] : hst[B + b];
        if (stb < 0) stb = 1;
        if (status) status[b] = stb;
        worst = std::max(worst, stb);
    }
    if (worst != 0) {
        char msg[160];
        snprintf(msg, sizeof(msg), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", worst);
        c->last_error = msg;
    }
    return worst;
}

extern "C" int gpcsd_loglik_grad(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out2, double *grad, int ngrad) {
    GP_API_BEGIN(c)
    GP_REQUIRE(hp != nullptr, -3, "loglik_grad: null hparams");
    if (int rc = drain_async(c)) return rc;      // (this path keeps its own status words)
    return loglik_grad_impl(c, hp, 1, out2, grad, ngrad, nullptr);
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_grad_batch(gpcsd_ctx *c, const gpcsd_hparams *hps, int nsets, double *out2, double *grad, int ngrad,
                                       int *status) {
    GP_API_BEGIN(c)
    GP_REQUIRE(hps && nsets >= 1 && status, -3, "loglik_grad_batch: bad arguments");
    if (int rc = drain_async(c)) return rc;
    (void)loglik_grad_impl(c, hps, nsets, out2, grad, ngrad, status);   // per-set failures are reported in status[], not as rc
    return 0;
    GP_API_END(c)
}

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
