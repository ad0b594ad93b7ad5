// Elementwise / pairwise Gram builders of the GPCSD hot path.
//
// These are tiny (<= 1200^2 outputs) and transcendental-bound, never HBM-bound; the rules that matter are
// coalesced stores, coordinates staged through LDS once per tile, and operation order identical to the
// reference expressions so results agree to the last few ulps (no fast-math, IEEE div/sqrt).
#include "devutil.hpp"
#include "kernels.hpp"

namespace gpcsd {

// ------------------------------------------------------------------------------------------------
// forward-model weights
// ------------------------------------------------------------------------------------------------
// T = double everywhere on the default path.  T = float is the "fp32 kernel build" of BASELINE cfg5: the Gram builders
// evaluate in single precision (coordinates and hyper-parameters rounded to float, float exp/log/sqrt) and widen the result,
// everything downstream (GEMMs, eigensolver, likelihood) stays fp64.  gpcsd_set_gram_precision() selects it per context.
template <typename T>
__device__ __forceinline__ T dev_b_fwd_1d(T r, T R) {
    // sqrt((r/R)^2 + 1) - sqrt((r/R)^2)            forward_models.py:16
    const T q = (r / R) * (r / R);
    return sqrt(q + T(1)) - sqrt(q);
}

template <typename T>
__device__ __forceinline__ T dev_b_fwd_2d_w(T w, T R, T eps) {
    // log(R+eps+sqrt((R+eps)^2+w^2)) - log(eps+sqrt(eps^2+w^2))      forward_models.py:53
    const T re = R + eps;
    return log(re + sqrt(re * re + w * w)) - log(eps + sqrt(eps * eps + w * w));
}

__global__ void b_fwd_1d_kernel(const double *__restrict__ r, long n, double R, double *__restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = dev_b_fwd_1d<double>(r[i], R);
}

__global__ void b_fwd_2d_kernel(const double *__restrict__ d1, const double *__restrict__ d2, const double *__restrict__ w,
                                long n, double R, double eps, double *__restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double ww = w ? w[i] : sqrt(d1[i] * d1[i] + d2[i] * d2[i]);   // forward_models.py:52
        out[i] = dev_b_fwd_2d_w<double>(ww, R, eps);
    }
}

static inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// Traditional CSD estimator (predict_csd.py:3-31): minus the second difference along the electrode axis of data laid out as
// (outer, axis, inner); the first and last position of the axis get `edge` (-0.0 in 1-D, NaN in 2-D).  HBM-bound: one read
// of three neighbouring planes (two of them cache hits) and one write per element.
__global__ void second_diff_kernel(const double *__restrict__ in, long n_axis, long n_inner, long total, double edge,
                                   double *__restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long a = (i / n_inner) % n_axis;
        out[i] = (a == 0 || a == n_axis - 1) ? edge : -(in[i + n_inner] + in[i - n_inner] - 2.0 * in[i]);
    }
}

void k_second_diff(gpcsd_ctx *c, const double *in, long n_outer, long n_axis, long n_inner, double edge, double *out, hipStream_t s) {
    const long total = n_outer * n_axis * n_inner;
    hipLaunchKernelGGL(second_diff_kernel, dim3(ew_grid(total)), dim3(256), 0, s, in, n_axis, n_inner, total, edge, out);
    GP_HIP(hipGetLastError());
}

void k_b_fwd_1d(gpcsd_ctx *c, const double *r, long n, double R, double *out, hipStream_t s) {
    hipLaunchKernelGGL(b_fwd_1d_kernel, dim3(ew_grid(n)), dim3(256), 0, s, r, n, R, out);
    GP_HIP(hipGetLastError());
}

void k_b_fwd_2d(gpcsd_ctx *c, const double *d1, const double *d2, const double *w, long n, double R, double eps,
                double *out, hipStream_t s) {
    hipLaunchKernelGGL(b_fwd_2d_kernel, dim3(ew_grid(n)), dim3(256), 0, s, d1, d2, w, n, R, eps, out);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// pairwise kernels: 64 x 16 output tile per 256-thread block, coordinates staged in LDS
// ------------------------------------------------------------------------------------------------
constexpr int PT_COLS = 64, PT_ROWS = 16;

struct TemporalParams {
    int ncomp;
    int kind[GPCSD_MAX_TEMPORAL];
    double ell[GPCSD_MAX_TEMPORAL];
    double sigma2[GPCSD_MAX_TEMPORAL];
};

// tab != nullptr: blockIdx.z = hyper-parameter set of a batched evaluation: the kernel parameters come from tab[z] and the
// output is the z-th matrix (s_out apart).  Same arithmetic, so each set gets the bits of a launch of its own.
template <typename T>
__global__ __launch_bounds__(256) void temporal_gram_kernel(TemporalParams p, const double *__restrict__ t, int n,
                                                            const double *__restrict__ tp, int m, double *__restrict__ out,
                                                            const HpDev *__restrict__ tab, long s_out) {
    if (tab) {
        const HpDev &h = tab[blockIdx.z];
        p.ncomp = h.ncomp;
#pragma unroll
        for (int cc = 0; cc < GPCSD_MAX_TEMPORAL; ++cc) {
            p.kind[cc] = h.kind[cc];
            p.ell[cc] = h.ell_t[cc];
            p.sigma2[cc] = h.sigma2_t[cc];
        }
        out += blockIdx.z * s_out;
    }
    __shared__ T st[PT_ROWS], stp[PT_COLS];
    const int c0 = blockIdx.x * PT_COLS, r0 = blockIdx.y * PT_ROWS;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x < PT_COLS) stp[threadIdx.x] = (c0 + threadIdx.x < m) ? T(tp[c0 + threadIdx.x]) : T(0);
    else if (threadIdx.x < PT_COLS + PT_ROWS) {
        int i = threadIdx.x - PT_COLS;
        st[i] = (r0 + i < n) ? T(t[r0 + i]) : T(0);
    }
    __syncthreads();
    const int col = c0 + tx;
    if (col >= m) return;
#pragma unroll
    for (int k = 0; k < PT_ROWS / 4; ++k) {
        const int rr = ty + 4 * k;
        const int row = r0 + rr;
        if (row >= n) continue;
        const T d = st[rr] - stp[tx];
        T acc = T(0);                                                  // Kt = zeros; Kt = Kt + K_c (gpcsd1d.py:118-120)
        for (int cc = 0; cc < p.ncomp; ++cc) {
            const T ell = T(p.ell[cc]), s2 = T(p.sigma2[cc]);
            T v;
            if (p.kind[cc] == GPCSD_KIND_SE)
                v = s2 * exp(T(-0.5) * (d * d) / (ell * ell));         // covariances.py:270
            else
                v = s2 * exp(-sqrt(d * d) / ell);                      // covariances.py:304
            acc = acc + v;
        }
        out[(long)row * m + col] = (double)acc;
    }
}

void k_temporal_gram(gpcsd_ctx *c, int ncomp, const int *kind, const double *ell, const double *sigma2, const double *t, int n,
                     const double *tp, int m, double *out, hipStream_t s, const HpDev *tab, int B, long s_out) {
    TemporalParams p{};
    if (!tab) {
        GP_REQUIRE(ncomp >= 1 && ncomp <= GPCSD_MAX_TEMPORAL, -3, "temporal gram: %d components (max %d)", ncomp,
                   GPCSD_MAX_TEMPORAL);
        p.ncomp = ncomp;
        for (int i = 0; i < ncomp; ++i) {
            p.kind[i] = kind[i];
            p.ell[i] = ell[i];
            p.sigma2[i] = sigma2[i];
        }
    }
    dim3 grid(ceil_div(m, PT_COLS), ceil_div(n, PT_ROWS), tab ? B : 1);
    ProfScope ps(c, "gram_temporal", 0.0, s);
    if (c->gram_fp32) hipLaunchKernelGGL(temporal_gram_kernel<float>, grid, dim3(256), 0, s, p, t, n, tp, m, out, tab, s_out);
    else hipLaunchKernelGGL(temporal_gram_kernel<double>, grid, dim3(256), 0, s, p, t, n, tp, m, out, tab, s_out);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// temporal chain input in one launch (see kernels.hpp: k_temporal_fold_fill)
// ------------------------------------------------------------------------------------------------
struct TFoldFillArgs {
    TemporalSet set[2];          // hyper-parameters of replica 0 / 1
    double inv_m[2], m[2];       // scale of replica r: entries are multiplied by inv_m = 1 / m (m a power of two)
    SymDev sy;
    const double *t;
    int n;
    double *A0s, *A0a, *Vs, *Va, *taus, *taua, *amaxs, *amaxa;
    long blks, blka;
    int *status;
    int status_stride;
    int elem_blocks;
    const HpDev *tab;            // non-null: replica r takes its hyper-parameters from tab[r] (any number of replicas: the batched
                                 // evaluations of fit) and forms its scale on the device by the same rule
};

// power of two >= 2 sum |sigma2_c|: every entry of either folded block is a signed combination of four kernel values with
// weights summing to at most 2 (host and device form it by the same rule: a set gets the same scale either way)
__host__ __device__ inline double tfold_scale(const double *sigma2, int ncomp) {
    double bound = 0.0;
    for (int cc = 0; cc < ncomp; ++cc) bound += fabs(sigma2[cc]);
    bound *= 2.0;
    double m = 1.0;
    if (bound > 0.0 && bound <= 1.7e308) {
        int ex = 0;
        (void)frexp(bound, &ex);                // bound = f * 2^ex, f in [0.5, 1)
        m = ldexp(1.0, ex);
        if (!(m <= 1.7e308)) m = bound;         // 2^1024 overflows: fall back to the bound itself
    }
    return m;
}

template <typename T, class P>
__device__ __forceinline__ double tset_eval(const P &p, const T ti, const T tj) {
    const T d = ti - tj;
    T acc = T(0);                                                      // Kt = zeros; Kt = Kt + K_c (gpcsd1d.py:118-120)
    for (int cc = 0; cc < p.ncomp; ++cc) {
        const T ell = T(p.ell_of(cc)), s2 = T(p.sigma2_of(cc));
        T v;
        if (p.kind[cc] == GPCSD_KIND_SE)
            v = s2 * exp(T(-0.5) * (d * d) / (ell * ell));             // covariances.py:270
        else
            v = s2 * exp(-sqrt(d * d) / ell);                          // covariances.py:304
        acc = acc + v;
    }
    return (double)acc;
}

template <typename T, class P>
__device__ __forceinline__ void fold_fill_entry(const TFoldFillArgs &g, const P &p, const double m, const int rep, const long e);

// blockIdx.y = replica.  Blocks [0, elem_blocks): one entry (a, b) of the symmetric block per thread (and of the
// antisymmetric block when a, b < na), the four Kt entries of the two orbits evaluated in place and combined exactly as
// sym_fold_kernel combines them.  Blocks beyond: zero-fill of the reflector storage of both classes.
template <typename T>
__global__ __launch_bounds__(256) void temporal_fold_fill_kernel(TFoldFillArgs g) {
    const int rep = blockIdx.y;
    const SymDev sy = g.sy;
    const int ns = sy.ns, na = sy.na;
    if ((int)blockIdx.x >= g.elem_blocks) {
        const long i0 = ((long)blockIdx.x - g.elem_blocks) * 256 + threadIdx.x, stride = ((long)gridDim.x - g.elem_blocks) * 256;
        double *Vs = g.Vs + rep * g.blks, *Va = g.Va + rep * g.blka, *ts = g.taus + rep * g.blks, *ta = g.taua + rep * g.blka;
        const long nvs = (long)(ns + 64) * ns, nva = (long)(na + 64) * na;
        for (long i = i0; i < nvs; i += stride) Vs[i] = 0.0;
        for (long i = i0; i < nva; i += stride) Va[i] = 0.0;
        for (long i = i0; i < ns + 66; i += stride) ts[i] = 0.0;           // (+2: the tail's progress word, eigh_dc.hip)
        for (long i = i0; i < na + 66; i += stride) ta[i] = 0.0;
        if (blockIdx.x == g.elem_blocks && threadIdx.x == 0) {
            const double mr = g.tab ? tfold_scale(g.tab[rep].sigma2_t, g.tab[rep].ncomp) : g.m[rep];
            g.amaxs[rep * g.blks] = mr;
            g.amaxa[rep * g.blka] = mr;
        }
        return;
    }
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= (long)ns * ns) return;
    if (g.tab) fold_fill_entry<T>(g, g.tab[rep], tfold_scale(g.tab[rep].sigma2_t, g.tab[rep].ncomp), rep, e);
    else fold_fill_entry<T>(g, g.set[rep], g.m[rep], rep, e);
}

// one entry (a, b) of the symmetric block (and of the antisymmetric one) of replica rep, hyper-parameters from p, scale m
template <typename T, class P>
__device__ __forceinline__ void fold_fill_entry(const TFoldFillArgs &g, const P &p, const double m, const int rep, const long e) {
    const SymDev &sy = g.sy;
    const int ns = sy.ns, na = sy.na;
    const int a = (int)(e / ns), b = (int)(e % ns);
    const int i = sy.rep_i[a], j = sy.rep_j[a], k = sy.rep_i[b], l = sy.rep_j[b];
    const T ti = T(g.t[i]), tj = T(g.t[j]), tk = T(g.t[k]), tl = T(g.t[l]);
    const double kik = tset_eval<T>(p, ti, tk);
    double ssum = kik, asum = kik;
    if (l != k) {
        const double v = tset_eval<T>(p, ti, tl);
        ssum += v;
        asum -= v;
    }
    if (j != i) {
        const double v = tset_eval<T>(p, tj, tk);
        ssum += v;
        asum -= v;
        if (l != k) {
            const double v2 = tset_eval<T>(p, tj, tl);
            ssum += v2;
            asum += v2;
        }
    }
    const double isq2 = 0.70710678118654752440;
    const double wa = (i == j) ? 1.0 : isq2, wb = (k == l) ? 1.0 : isq2;
    const double inv = 1.0 / m;                      // (m is a power of two: exact)
    bool bad = false;
    auto scaled = [&](double x) {
        x = (m > 1e300) ? x / m : x * inv;          // 1 / m is subnormal beyond 1e300
        if (!(fabs(x) <= 2.0)) {                    // non-finite (a NaN / inf hyper-parameter): never reaches the solver
            x = 0.0;
            bad = true;
        }
        return x;
    };
    g.A0s[rep * g.blks + e] = scaled(wa * wb * ssum);
    if (a < na && b < na) g.A0a[rep * g.blka + (long)a * na + b] = scaled(0.5 * asum);
    if (bad && g.status) atomicMax(g.status + (long)rep * g.status_stride, 4);
}

void k_temporal_fold_fill(gpcsd_ctx *c, const TemporalSet *sets, int nrep, const double *t, int n, const SymDev &sy,
                          const EigArenaView &as, const EigArenaView &aa, int *status, int status_stride, hipStream_t s) {
    GP_REQUIRE(nrep >= 1 && nrep <= 2, -3, "temporal fold fill: %d replicas (1 or 2)", nrep);
    GP_REQUIRE(sy.ns > 0 && sy.ns + sy.na == n, -3, "temporal fold fill: the symmetry does not cover the %d time points", n);
    TFoldFillArgs g{};
    for (int r = 0; r < nrep; ++r) {
        g.set[r] = sets[r];
        GP_REQUIRE(sets[r].ncomp >= 1 && sets[r].ncomp <= GPCSD_MAX_TEMPORAL, -3, "temporal gram: %d components (max %d)",
                   sets[r].ncomp, GPCSD_MAX_TEMPORAL);
        g.m[r] = tfold_scale(sets[r].sigma2, sets[r].ncomp);
        g.inv_m[r] = 1.0 / g.m[r];
    }
    g.sy = sy;
    g.t = t;
    g.n = n;
    g.A0s = as.A0; g.A0a = aa.A0; g.Vs = as.V; g.Va = aa.V; g.taus = as.tau; g.taua = aa.tau; g.amaxs = as.amax; g.amaxa = aa.amax;
    g.blks = as.blk; g.blka = aa.blk;
    g.status = status;
    g.status_stride = status_stride;
    g.elem_blocks = ceil_div((long)sy.ns * sy.ns, 256);
    const int zero_blocks = 64;
    // sums of SE / Matern-1/2 kernels with non-negative variances: positive semi-definite (EigArenaView::psd)
    bool nonneg = true;
    for (int r = 0; r < nrep; ++r)
        for (int cc = 0; cc < sets[r].ncomp; ++cc) nonneg = nonneg && sets[r].sigma2[cc] >= 0.0;
    *as.psd = *aa.psd = nonneg;
    ProfScope ps(c, "gram_temporal_fold_fill", 0.0, s);
    if (c->gram_fp32)
        hipLaunchKernelGGL(temporal_fold_fill_kernel<float>, dim3(g.elem_blocks + zero_blocks, nrep), dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL(temporal_fold_fill_kernel<double>, dim3(g.elem_blocks + zero_blocks, nrep), dim3(256), 0, s, g);
    GP_HIP(hipGetLastError());
}

// the same for B sets whose hyper-parameters sit in a device table (the lock-step batches of fit)
void k_temporal_fold_fill_tab(gpcsd_ctx *c, const HpDev *tab, int B, const double *t, int n, const SymDev &sy,
                              const EigArenaView &as, const EigArenaView &aa, int *status, int status_stride, hipStream_t s,
                              bool variances_nonneg) {
    GP_REQUIRE(tab && B >= 1, -3, "temporal fold fill: no hyper-parameter table");
    *as.psd = *aa.psd = variances_nonneg;          // (the table is on the device: its owner vouches for the signs)
    GP_REQUIRE(sy.ns > 0 && sy.ns + sy.na == n, -3, "temporal fold fill: the symmetry does not cover the %d time points", n);
    TFoldFillArgs g{};
    g.tab = tab;
    g.sy = sy;
    g.t = t;
    g.n = n;
    g.A0s = as.A0; g.A0a = aa.A0; g.Vs = as.V; g.Va = aa.V; g.taus = as.tau; g.taua = aa.tau; g.amaxs = as.amax; g.amaxa = aa.amax;
    g.blks = as.blk; g.blka = aa.blk;
    g.status = status;
    g.status_stride = status_stride;
    g.elem_blocks = ceil_div((long)sy.ns * sy.ns, 256);
    const int zero_blocks = 64;
    ProfScope ps(c, "gram_temporal_fold_fill", 0.0, s);
    if (c->gram_fp32)
        hipLaunchKernelGGL(temporal_fold_fill_kernel<float>, dim3(g.elem_blocks + zero_blocks, B), dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL(temporal_fold_fill_kernel<double>, dim3(g.elem_blocks + zero_blocks, B), dim3(256), 0, s, g);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// log-likelihood in the basis U (x) Q: shifted tridiagonal systems (see kernels.hpp: k_ll_tridiag)
// ------------------------------------------------------------------------------------------------
// One wave per item (x', p) = (spatial eigen-row in fold order, temporal parity block): with Kt_p = m_p Q_p T_p Q_p^T and
// lam = es[x'], the block of Ks (x) Kt + sig2 I belonging to the item is Q_p (lam m_p T_p + sig2 I) Q_p^T, a shifted
// symmetric tridiagonal matrix A = L D L^T (no pivoting: positive definite for lam >= 0, sig2 > 0).  Its log-determinant is
// sum log D_k, and for a row w = (U^T Y Q)[x', r, p block] the quadratic form w^T A^-1 w = sum z_k^2 / D_k with z = L^-1 w,
// i.e. ONE forward recurrence per (item, trial): lane = trial, every lane runs the (uniform) pivot recurrence beside its own.
// The rows are staged through LDS in chunks of LT_CK columns (coalesced 256-byte pieces), the logs of a chunk's pivots are
// taken by 32 lanes in parallel.  Partials per item: [0, nitems) quadratic forms, [nitems, 2 nitems) log-determinants.
constexpr int LT_CK = 32, LT_WAVES = 4, LT_RPL = 32;     // columns per chunk; waves per workgroup; staged rows per lane (64 trials)
struct LlTridiagArgs {
    const double *W;             // (U^T Y Q) in the layout [x'][r][t~], rows of nt doubles
    const double *es;            // spatial eigenvalues, fold order (nx)
    const double *d[2], *e[2];   // tridiagonal of the scaled temporal blocks (np entries each)
    const double *amax[2];       // their scales m_p
    const double *sig;           // scalar noise variance (device)
    int nx, R, nt, np[2], c0[2];
    double *partials;
};
__global__ __launch_bounds__(64 * LT_WAVES) void ll_tridiag_kernel(LlTridiagArgs g) {
    __shared__ double tile[LT_WAVES][2][64][LT_CK + 1];            // double-buffered chunk of the wave's rows
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int item = blockIdx.x * LT_WAVES + wid, nitems = 2 * g.nx;
    if (item >= nitems) return;                                   // (whole waves leave: no workgroup barrier below)
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    if (np <= 0) {
        if (lane == 0) {
            g.partials[item] = 0.0;
            g.partials[nitems + item] = 0.0;
        }
        return;
    }
    const double lam_m = g.es[xr] * g.amax[p][0], sig = g.sig[0];
    const double *__restrict__ dd = g.d[p], *__restrict__ ee = g.e[p];
    const double *__restrict__ Wx = g.W + (long)xr * g.R * g.nt + g.c0[p];
    const int half = lane >> 5, kk_l = lane & 31;                 // staging: two rows of 32 columns per load instruction
    double quad = 0.0, logsum = 0.0;
    for (int r0 = 0; r0 < g.R; r0 += 64) {
        const int nr = min(64, g.R - r0);
        double stg[LT_RPL];
        // all loads of a chunk are issued before any of them is used: one L2 / HBM round trip per chunk, not one per row pair
        auto load_chunk = [&](int k0) {
            const int nk = min(LT_CK, np - k0);
#pragma unroll
            for (int i = 0; i < LT_RPL; ++i) {
                const int rr = half + 2 * i;
                stg[i] = (rr < nr && kk_l < nk) ? Wx[(long)(r0 + rr) * g.nt + k0 + kk_l] : 0.0;
            }
        };
        auto store_chunk = [&](int buf) {
#pragma unroll
            for (int i = 0; i < LT_RPL; ++i) tile[wid][buf][half + 2 * i][kk_l] = stg[i];
        };
        // the tridiagonal's entries of a chunk travel one chunk ahead as well (lane kk < 32 holds d / e of column k0 + kk)
        double cd = 0.0, ce = 0.0;
        auto load_coef = [&](int k0) {
            const int k = k0 + (lane & 31);
            cd = (k < np) ? dd[k] : 0.0;
            ce = (k < np) ? ee[k] : 0.0;
        };
        load_coef(0);
        load_chunk(0);
        store_chunk(0);
        double piv_inv = 0.0, bprev = 0.0, z = 0.0, q = 0.0;      // 1 / D_{k-1}, offdiagonal b_{k-1}, z_{k-1}
        int buf = 0;
        for (int k0 = 0; k0 < np; k0 += LT_CK, buf ^= 1) {
            const int nk = min(LT_CK, np - k0);
            // a_k = lam m d_k + sig2, b_k = lam m e_k of this chunk: lane kk holds column k0 + kk's, the recurrence reads them
            // with v_readlane (a scalar per step, no LDS on the serial path)
            const double ca = lam_m * cd + sig, cb = lam_m * ce;
            if (k0 + LT_CK < np) {                                // the next chunk's loads fly during this chunk's recurrence
                load_coef(k0 + LT_CK);
                load_chunk(k0 + LT_CK);
            }
            __builtin_amdgcn_wave_barrier();
            double wv[LT_CK];                                     // this lane's row of the chunk: one batch of LDS reads
#pragma unroll
            for (int kk = 0; kk < LT_CK; ++kk) wv[kk] = tile[wid][buf][lane][kk];
            // the serial part, all in registers: per step one multiply, one fma and a reciprocal on the pivot chain (v_rcp_f64 +
            // two Newton steps, ~2 ulp: an IEEE division is 3x the latency and every one of the 250 steps waits for it)
            double mypiv = 1.0;
#pragma unroll
            for (int kk = 0; kk < LT_CK; ++kk) {
                if (kk < nk) {                                    // wave-uniform
                    const double l = bprev * piv_inv;             // L_{k,k-1} = b_{k-1} / D_{k-1}  (0 for k = 0)
                    const double piv = lane_get(ca, kk) - l * bprev;
                    z = wv[kk] - l * z;
                    piv_inv = fast_rcp(piv);
                    q += z * z * piv_inv;
                    bprev = lane_get(cb, kk);
                    if (lane == kk) mypiv = piv;
                }
            }
            if (r0 == 0) {                                        // log-determinant: once per item, lane kk takes log D_{k0+kk}
                const double lg = (lane < nk) ? log(mypiv) : 0.0;
                logsum += wave_sum(lg);
            }
            __builtin_amdgcn_wave_barrier();
            if (k0 + LT_CK < np) store_chunk(buf ^ 1);
        }
        quad += wave_sum(lane < nr ? q : 0.0);
    }
    if (lane == 0) {
        g.partials[item] = quad;
        g.partials[nitems + item] = logsum;
    }
}

// The same with the pivots off the serial path (blocks of at most 64 * LS_CPL columns).  In the kernel above every column step
// waits for a chain of ~8 dependent fp64 operations -- multiplier, pivot, reciprocal with its Newton steps -- that all lanes
// walk in lock-step: 250 x ~200 cycles = 21 us of its 38.  The pivots D_k = a_k - b_{k-1}^2 / D_{k-1} are ratios theta_k /
// theta_{k-1} of leading principal minors, (theta_k, theta_{k-1})^T = M_k (theta_{k-1}, theta_{k-2})^T with M_k = [[a_k,
// -b_{k-1}^2], [1, 0]]: lane j multiplies the M_k of its LS_CPL consecutive columns, a wave scan (six shuffle steps, every
// product renormalised by a power of two: positive definite, all minors positive) gives each lane the product of everything
// in front of it, and a second walk over its own columns the pivots themselves.  One step of the exact recurrence from the
// neighbour's pivot (D_k = a_k - b_{k-1}^2 / D_{k-1}, all columns at once) then removes what the re-associated products
// lost: a relative error delta in D_{k-1} reaches D_k multiplied by b^2 / (D_{k-1} D_k) < 1.  Multipliers and reciprocal
// pivots go to LDS as pairs; a column step of the sweep is ONE dependent FMA (z) plus one on a rotating accumulator (q),
// coefficients by LDS broadcast reads a chunk ahead of their use.
constexpr int LS_CPL = 4;                                           // columns per lane in the scan: blocks of up to 256 columns
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64 * LT_WAVES) void ll_tridiag_scan_kernel(LlTridiagArgs g) {
    __shared__ double tile[LT_WAVES][2][64][LT_CK + 1];            // double-buffered chunk of the wave's rows
    __shared__ dbl2_t coef[LT_WAVES][64 * LS_CPL + LT_CK];          // (l_k, 1 / D_k); zero beyond the block
    __shared__ double pexc[LT_WAVES][64 * LS_CPL + 1];              // pivot reciprocals on their way between lanes
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * LT_WAVES + wid, nitems = 2 * g.nx;
    if (item >= nitems) return;                                   // (whole waves leave: no workgroup barrier below)
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    if (np <= 0) {
        if (lane == 0) {
            g.partials[item] = 0.0;
            g.partials[nitems + item] = 0.0;
        }
        return;
    }
    const double lam_m = g.es[xr] * g.amax[p][0], sig = g.sig[0];
    const double *__restrict__ dd = g.d[p], *__restrict__ ee = g.e[p];
    const double *__restrict__ Wx = g.W + (long)xr * g.R * g.nt + g.c0[p];
    const int half = lane >> 5, kk_l = lane & 31;                 // staging: two rows of 32 columns per load instruction
    double stg[LT_RPL];
    auto load_chunk = [&](int r0, int nr, int k0) {               // all loads of a chunk are issued before any of them is used
        const int nk = min(LT_CK, np - k0);
#pragma unroll
        for (int i = 0; i < LT_RPL; ++i) {
            const int rr = half + 2 * i;
            stg[i] = (rr < nr && kk_l < nk) ? Wx[(long)(r0 + rr) * g.nt + k0 + kk_l] : 0.0;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LT_RPL; ++i) tile[wid][buf][half + 2 * i][kk_l] = stg[i];
    };
    load_chunk(0, min(64, g.R), 0);                               // the first chunk's rows fly while the pivots are formed
    // ---- pivots of the item
    double logsum;
    {
        double a[LS_CPL], b2[LS_CPL], bb[LS_CPL];
#pragma unroll
        for (int cc = 0; cc < LS_CPL; ++cc) {
            const int k = lane * LS_CPL + cc;
            a[cc] = (k < np) ? lam_m * dd[k] + sig : 1.0;
            bb[cc] = (k >= 1 && k < np) ? lam_m * ee[k - 1] : 0.0;
            b2[cc] = bb[cc] * bb[cc];
        }
        auto renorm = [](double &x0, double &x1, double &x2, double &x3) {
            const double big = fmax(fmax(fabs(x0), fabs(x1)), fmax(fabs(x2), fabs(x3)));
            int ex;
            (void)frexp(big, &ex);
            x0 = ldexp(x0, -ex); x1 = ldexp(x1, -ex); x2 = ldexp(x2, -ex); x3 = ldexp(x3, -ex);
        };
        // the lane's own product M_last .. M_first
        double m00 = a[0], m01 = -b2[0], m10 = 1.0, m11 = 0.0;
#pragma unroll
        for (int cc = 1; cc < LS_CPL; ++cc) {                      // M_k P: rows (a P0 - b2 P1, P0)
            const double n00 = a[cc] * m00 - b2[cc] * m10, n01 = a[cc] * m01 - b2[cc] * m11;
            m10 = m00; m11 = m01; m00 = n00; m01 = n01;
        }
        renorm(m00, m01, m10, m11);
        // inclusive scan over the lanes: P_j <- P_j P_{j - dlt}
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const double e00 = __shfl_up(m00, dlt), e01 = __shfl_up(m01, dlt), e10 = __shfl_up(m10, dlt), e11 = __shfl_up(m11, dlt);
            if (lane >= dlt) {
                double n00 = m00 * e00 + m01 * e10, n01 = m00 * e01 + m01 * e11;
                double n10 = m10 * e00 + m11 * e10, n11 = m10 * e01 + m11 * e11;
                renorm(n00, n01, n10, n11);
                m00 = n00; m01 = n01; m10 = n10; m11 = n11;
            }
        }
        // (theta, theta') in front of this lane's columns: first column of the product of the lanes before it
        double t0 = __shfl_up(m00, 1), t1 = __shfl_up(m10, 1);
        if (lane == 0) { t0 = 1.0; t1 = 0.0; }
        double pv[LS_CPL];                                         // 1 / D_k from the products
#pragma unroll
        for (int cc = 0; cc < LS_CPL; ++cc) {
            const double tn = a[cc] * t0 - b2[cc] * t1;
            pv[cc] = t0 / tn;
            t1 = t0; t0 = tn;
            if (cc == 1) {                                         // keep the pair in range (a ratio is all that is used)
                int ex;
                (void)frexp(t0, &ex);
                t0 = ldexp(t0, -ex); t1 = ldexp(t1, -ex);
            }
        }
#pragma unroll
        for (int cc = 0; cc < LS_CPL; ++cc) pexc[wid][lane * LS_CPL + cc + 1] = pv[cc];
        if (lane == 0) pexc[wid][0] = 0.0;
        __builtin_amdgcn_wave_barrier();
        // one step of the exact recurrence, every column from its left neighbour's reciprocal pivot
        double dk[LS_CPL], lg = 0.0;
#pragma unroll
        for (int cc = 0; cc < LS_CPL; ++cc) {
            const int k = lane * LS_CPL + cc;
            dk[cc] = a[cc] - b2[cc] * pexc[wid][k];
            pv[cc] = 1.0 / dk[cc];
            if (k < np) lg += log(dk[cc]);
        }
        logsum = wave_sum(lg);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cc = 0; cc < LS_CPL; ++cc) pexc[wid][lane * LS_CPL + cc + 1] = pv[cc];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cc = 0; cc < LS_CPL; ++cc) {
            const int k = lane * LS_CPL + cc;
            dbl2_t cf;
            cf.x = (k < np) ? bb[cc] * pexc[wid][k] : 0.0;         // l_k = b_{k-1} / D_{k-1}
            cf.y = (k < np) ? pv[cc] : 0.0;
            coef[wid][k] = cf;
        }
        if (lane < LT_CK) coef[wid][64 * LS_CPL + lane] = dbl2_t{0.0, 0.0};
    }
    // ---- the sweeps: the item's rows in passes of 64, chunks of LT_CK columns
    double quad = 0.0;
    for (int r0 = 0; r0 < g.R; r0 += 64) {
        const int nr = min(64, g.R - r0);
        if (r0 > 0) load_chunk(r0, nr, 0);
        __builtin_amdgcn_wave_barrier();
        store_chunk(0);
        double z = 0.0, q[4] = {0.0, 0.0, 0.0, 0.0};
        int buf = 0;
        for (int k0 = 0; k0 < np; k0 += LT_CK, buf ^= 1) {
            if (k0 + LT_CK < np) load_chunk(r0, nr, k0 + LT_CK);  // the next chunk's loads fly during this chunk's recurrence
            __builtin_amdgcn_wave_barrier();
            double wv[LT_CK];                                     // this lane's row of the chunk: one batch of LDS reads
            dbl2_t cf[LT_CK];
#pragma unroll
            for (int kk = 0; kk < LT_CK; ++kk) {
                wv[kk] = tile[wid][buf][lane][kk];
                cf[kk] = coef[wid][k0 + kk];
            }
#pragma unroll
            for (int kk = 0; kk < LT_CK; ++kk) {                  // (columns beyond the block: w = 0, l = 0, 1 / D = 0)
                z = fma(-cf[kk].x, z, wv[kk]);
                q[kk & 3] = fma(z * z, cf[kk].y, q[kk & 3]);
            }
            __builtin_amdgcn_wave_barrier();
            if (k0 + LT_CK < np) store_chunk(buf ^ 1);
        }
        quad += wave_sum(lane < nr ? (q[0] + q[1]) + (q[2] + q[3]) : 0.0);
    }
    if (lane == 0) {
        g.partials[item] = quad;
        g.partials[nitems + item] = logsum;
    }
}

// host_slot != null: the two sums also go straight into the caller's pinned result slot (device-accessible host memory:
// host_slot[0] = sum log D, host_slot[1] = quadratic form) together with the status words (status_doubles doubles from
// status_src to host_slot + status_at) -- the launch is the last of the evaluation, its results are visible to the host when
// the event behind it completes, and the separate device-to-host copy (a 4 us blit kernel behind a 6 us gap) is not queued.
__global__ __launch_bounds__(256) void ll_tridiag_reduce_kernel(const double *__restrict__ partials, int nitems, double *out_sumlog,
                                                                double *out_quad, double *host_slot, const double *status_src,
                                                                int status_at, int status_doubles) {
    __shared__ double sh[256];
    const double *p = partials + (blockIdx.x == 0 ? nitems : 0);
    double s = 0.0;
    for (int i = threadIdx.x; i < nitems; i += 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        (blockIdx.x == 0 ? out_sumlog : out_quad)[0] = sh[0];
        if (host_slot) host_slot[blockIdx.x == 0 ? 0 : 1] = sh[0];
    }
    if (host_slot && blockIdx.x == 0 && (int)threadIdx.x < status_doubles) host_slot[status_at + threadIdx.x] = status_src[threadIdx.x];
}

// the final sums of the per-item partials
void ll_tridiag_reduce_launch(gpcsd_ctx *c, const double *partials, int nitems, double *out_sumlog, double *out_quad, double *host_slot,
                              const double *status_src, int status_at, int status_doubles, hipStream_t s, bool *wrote) {
    (void)c;
    static const bool direct = !(getenv("GPCSD_LL_HOST_WRITE") && getenv("GPCSD_LL_HOST_WRITE")[0] == '0');
    if (!direct || status_doubles > 256) host_slot = nullptr;
    hipLaunchKernelGGL(ll_tridiag_reduce_kernel, dim3(2), dim3(256), 0, s, partials, nitems, out_sumlog, out_quad, host_slot, status_src,
                       status_at, host_slot ? status_doubles : 0);
    *wrote = host_slot != nullptr;
}

bool k_ll_tridiag(gpcsd_ctx *c, const double *W, const double *es, const double *const d[2], const double *const e[2],
                  const double *const amax[2], const double *sig, int nx, int R, int nt, const int np[2], const int c0[2],
                  double *out_sumlog, double *out_quad, hipStream_t s, double *host_slot, const double *status_src, int status_at,
                  int status_doubles) {
    LlTridiagArgs g{};
    g.W = W; g.es = es; g.sig = sig; g.nx = nx; g.R = R; g.nt = nt;
    for (int p = 0; p < 2; ++p) {
        g.d[p] = d[p]; g.e[p] = e[p]; g.amax[p] = amax[p]; g.np[p] = np[p]; g.c0[p] = c0[p];
    }
    const int nitems = 2 * nx;
    g.partials = c->buf<double>("ll_tridiag_partials", (size_t)2 * nitems);
    ProfScope ps(c, "ll_tridiag", 0.0, s);
    static const bool serial = getenv("GPCSD_LL_PIVOT_SCAN") && getenv("GPCSD_LL_PIVOT_SCAN")[0] == '0';
    if (!serial && std::max(np[0], np[1]) <= 64 * LS_CPL)
        hipLaunchKernelGGL(ll_tridiag_scan_kernel, dim3(ceil_div(nitems, LT_WAVES)), dim3(64 * LT_WAVES), 0, s, g);
    else
        hipLaunchKernelGGL(ll_tridiag_kernel, dim3(ceil_div(nitems, LT_WAVES)), dim3(64 * LT_WAVES), 0, s, g);
    bool wrote = false;
    ll_tridiag_reduce_launch(c, g.partials, nitems, out_sumlog, out_quad, host_slot, status_src, status_at, status_doubles, s, &wrote);
    GP_HIP(hipGetLastError());
    return wrote;
}

// ------------------------------------------------------------------------------------------------
// posterior mean in the basis U (x) Q: the same shifted tridiagonal systems, SOLVED (kernels.hpp: k_tridiag_solve)
// ------------------------------------------------------------------------------------------------
// predict needs B = (Ks (x) Kt + sig2 I)^-1 y in a basis in which the remaining products are plain GEMMs.  In the basis U (x) Q
// that is, per item (x', p) and trial, one solve with A = lam m T_p + sig2 I = L D L^T: forward z_k = w_k - l_k z_{k-1}, backward
// x_k = z_k / D_k - l_{k+1} x_{k+1}.  One wave per item and workgroup; lane = trial; the item's z (np x trials) stays in LDS between
// the two sweeps (k-major: conflict-free), so the rows travel through HBM once in and once out -- the traffic of the GEMM
// (W V) / D this replaces, without its operand V: the temporal divide & conquer and back-transformation are then on nobody's
// path.  Pivots and multipliers are formed once per item (first pass over the trials) and kept in LDS, a lane per column of a chunk
// holds them during a sweep and the serial chain reads them with v_readlane.
// Per item: the rows come into LDS in one burst (64 loads per lane in flight at a time; 16-column pieces of a row = one 128-byte
// line, written transposed), and the item's 250 pivots -- a serial chain of ~56 cycles per column that every lane would otherwise
// drag through its own sweep -- are formed in the shadow of those loads.  The sweeps then run in place in LDS, 64 columns per
// batch: 64 reads up front, one dependent FMA per column with the multiplier from an SGPR (v_readlane), 64 writes.  (A first
// version streamed 16-column chunks one ahead through a staging tile and formed the pivots inside the forward sweep, as the
// log-likelihood's kernel does: 83 us per item against ~15 -- and with 122 KB of LDS per item only one item runs per CU, three
// rounds per launch.)
constexpr int TS_BATCH = 64;                    // columns per batch of the sweeps / of the load burst
constexpr int TS_P = 64;                        // LDS row of a column: one entry per lane (trial) of the pass (the widest form)
constexpr int TS_P_DEFAULT = 64;
struct TriSolveArgs {
    const double *W;             // (U^T Y Q) in the layout [x'][r][t~], rows of nt doubles
    double *B;                   // the solutions, same layout (may be W)
    const double *es;
    const double *d[2], *e[2], *amax[2];
    const double *sig;
    int nx, R, nt, np[2], c0[2];
    int npad;                    // column capacity of the LDS block: max(np) rounded up to TS_BATCH
    unsigned long long *clk;     // measurement aid (GPCSD_TS_CLK=1): wall-clock stamps of workgroup 0 at the phase boundaries
};
// Written for the scalar unit as much as for the vector one: NO per-element predicate inside the unrolled batches (a first version
// guarded every LDS access with `column < np && lane < P`: the compiler turned each into an exec-mask save / restore with scalar
// compares, ran out of SGPRs and spilled them to VGPR lanes -- 200 cycles per column step).  Instead the LDS block is padded to
// whole batches with zero columns whose multipliers are zero (the recurrences pass through them unchanged), every lane owns a
// row of the block (trials beyond the pass are duplicate rows nobody stores), and the global accesses of a piece are classified
// wave-uniformly as whole, partial or absent.
// Four waves per item: each brings in and takes out one batch of 64 columns (all of an item's ~200 loads in flight at once: with
// one wave the four batches queued behind each other, 26 us), wave 0 forms the pivots meanwhile and runs both sweeps.  The
// coefficients of a sweep are LDS broadcast reads into VGPRs, 32 columns ahead -- as v_readlane operands (SGPRs) they cost
// ~40 cycles per column step against the 8 of the dependent FMA itself.
constexpr int TS_HB = 32;                       // columns per half batch of a sweep (registers: 32 values + 2 x 32 coefficients)
constexpr int TS_KMAX = 256;                    // columns of a block at most: one per thread of the pivot scan, one batch per wave
template <int P>
__global__ __launch_bounds__(256, P == 32 ? 2 : 1) void tridiag_solve_kernel(TriSolveArgs g) {
    constexpr int HB = (P == 32) ? 16 : TS_HB;     // columns per half batch of a sweep: the two-per-CU form has half the registers
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zbuf = smem;                               // [npad / 2][64][2]: w, then z, then x, in place -- two consecutive columns of
                                                       // a lane side by side (one 16-byte LDS access per two column steps)
    double *pinv = zbuf + (long)g.npad * P;         // [TS_KMAX + TS_BATCH]: 1 / D_k
    double *lmul = pinv + TS_KMAX + TS_BATCH;          // [TS_KMAX + TS_BATCH]: l_k = b_{k-1} / D_{k-1} (0 for k = 0 and k >= np)
    double *scan = lmul + TS_KMAX + TS_BATCH;          // [2][256][4]: the 2 x 2 prefix products of the pivot scan (pass 0)
    auto stamp = [&](int k) { if (g.clk && blockIdx.x == 0 && threadIdx.x == 0) g.clk[k] = wall_clock64(); };
    stamp(0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // A workgroup takes the items blockIdx.x, blockIdx.x + gridDim.x, ..: the grid is capped below the CU count (k_tridiag_solve) so
    // that this kernel -- one workgroup per CU, 140 KB of LDS and 280 VGPRs each, several rounds back to back -- never holds every
    // CU: the next call's eigen-chains start beside it, and their one-workgroup tridiagonalisation needs a whole CU's registers
    // (with 768 workgroups on all 256 CUs the spatial chain of the next step waited ~0.1 ms for one).
  for (int item = blockIdx.x; item < 2 * g.nx; item += gridDim.x) {
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    if (np <= 0) continue;                             // (the whole workgroup: no barrier is skipped by part of it)
    const int nbatch = (np + TS_BATCH - 1) / TS_BATCH, npad = nbatch * TS_BATCH;
    const double lam_m = g.es[xr] * g.amax[p][0], sig = g.sig[0];
    const double *__restrict__ dd = g.d[p], *__restrict__ ee = g.e[p];
    const long rowbase = (long)xr * g.R * g.nt + g.c0[p];
    const int quarter = lane >> 4, kk_l = lane & 15;   // a global access: four rows of 16 columns
    for (int r0 = 0; r0 < g.R; r0 += P) {
        const int nr = min(P, g.R - r0);
        const bool first = r0 == 0;
        const double *const wl = g.W + rowbase + (long)(r0 + quarter) * g.nt + kk_l;     // this lane's corner of a piece
        // ---- the rows of this pass into LDS: wave w takes batch w (npad <= 256: at most four batches, one per wave); the loads
        // are issued first and land while the pivots are formed
        const int b0 = wid * TS_BATCH;
        double stg[4][P / 4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kq = b0 + 16 * q;                              // wave-uniform classification of the piece's columns
            // rows beyond the pass and columns beyond the block read a clamped (legal) address and are NOT zeroed: a duplicate
            // row is a lane whose results are never stored, a duplicate column feeds a padding column that nothing reads back
            // (its multipliers are zero) -- and 64 loop-invariant lane masks would live in SGPRs the kernel does not have
            const int kc = min(kk_l, max(np - 1 - kq, 0));
#pragma unroll
            for (int i = 0; i < P / 4; ++i) {
                double v = 0.0;
                if (4 * i < nr && kq < np) {                         // wave-uniform: some row / column of the piece exists
                    const int rc = min(quarter + 4 * i, nr - 1) - quarter;
                    v = wl[(long)rc * g.nt + kq + (kc - kk_l)];
                }
                stg[q][i] = v;
            }
        }
        if (first) {
            // The pivots D_k = a_k - b_{k-1}^2 / D_{k-1} of the item, once: as a serial chain they are 250 x 7 dependent fp64
            // operations (~175 cycles per column: 18 us, with three of the four waves waiting).  D_k is the ratio theta_k /
            // theta_{k-1} of leading principal minors, and (theta_k, theta_{k-1})^T = M_k (theta_{k-1}, theta_{k-2})^T with
            // M_k = [[a_k, -b_{k-1}^2], [1, 0]]: an inclusive scan over 2 x 2 matrix products, thread = column, eight doubling
            // steps.  A prefix product may be scaled by any factor without changing the ratio: every product is renormalised by a
            // power of two (positive definite blocks: all minors positive).
            const int k = tid;
            const bool real = k < np;
            const double ak = real ? lam_m * dd[k] + sig : 1.0;
            const double bkm = (k >= 1 && k < np) ? lam_m * ee[k - 1] : 0.0;
            double m00 = ak, m01 = -bkm * bkm, m10 = 1.0, m11 = 0.0;
            double *buf = scan;
#pragma unroll 1
            for (int dlt = 1; dlt < 256; dlt <<= 1, buf = (buf == scan) ? scan + 1024 : scan) {
                buf[4 * k + 0] = m00; buf[4 * k + 1] = m01; buf[4 * k + 2] = m10; buf[4 * k + 3] = m11;
                __syncthreads();
                if (k >= dlt) {                                      // P <- P (later) x P_{k - dlt} (earlier)
                    const double e00 = buf[4 * (k - dlt) + 0], e01 = buf[4 * (k - dlt) + 1], e10 = buf[4 * (k - dlt) + 2],
                                 e11 = buf[4 * (k - dlt) + 3];
                    const double n00 = m00 * e00 + m01 * e10, n01 = m00 * e01 + m01 * e11;
                    const double n10 = m10 * e00 + m11 * e10, n11 = m10 * e01 + m11 * e11;
                    const double big = fmax(fmax(fabs(n00), fabs(n01)), fmax(fabs(n10), fabs(n11)));
                    int ex;
                    (void)frexp(big, &ex);
                    m00 = ldexp(n00, -ex); m01 = ldexp(n01, -ex); m10 = ldexp(n10, -ex); m11 = ldexp(n11, -ex);
                }
            }
            // first column of the prefix product = (theta_k, theta_{k-1}) up to a common factor
            const double pk = m10 / m00;                             // 1 / D_k
            pinv[k] = real ? pk : 1.0;
            buf[k] = pk;
            __syncthreads();
            lmul[k] = (k >= 1 && real) ? bkm * buf[k - 1] : 0.0;     // l_k = b_{k-1} / D_{k-1}; zero through the padding columns
            if (k < TS_BATCH) {
                pinv[TS_KMAX + k] = 1.0;
                lmul[TS_KMAX + k] = 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < P / 4; ++i)
                if (b0 < npad) zbuf[((b0 + 16 * q + kk_l) >> 1) * (2 * P) + (quarter + 4 * i) * 2 + (kk_l & 1)] = stg[q][i];
        __syncthreads();
        stamp(1);
        if (wid == 0) {
            typedef double dbl2 __attribute__((ext_vector_type(2)));
            dbl2 *const zl = reinterpret_cast<dbl2 *>(zbuf) + (lane & (P - 1));      // this lane's row of the block: columns 2j, 2j + 1 at
                                                                               // zl[j * P] (P = 32: the upper half wave doubles the lower)
            // ---- forward sweep in place: z_k = w_k - l_k z_{k-1}
            double z = 0.0;
            for (int h0 = 0; h0 < npad; h0 += HB) {
                dbl2 *const zb = zl + (h0 >> 1) * P;
                const dbl2 *const lc = reinterpret_cast<const dbl2 *>(lmul + h0);
                dbl2 v[HB / 2], cl[HB / 2];
#pragma unroll
                for (int j = 0; j < HB / 2; ++j) {
                    v[j] = zb[j * P];
                    cl[j] = lc[j];
                }
#pragma unroll
                for (int j = 0; j < HB / 2; ++j) {
                    z = fma(-cl[j].x, z, v[j].x);
                    v[j].x = z;
                    z = fma(-cl[j].y, z, v[j].y);
                    v[j].y = z;
                }
#pragma unroll
                for (int j = 0; j < HB / 2; ++j) zb[j * P] = v[j];
            }
            stamp(2);
            // ---- backward sweep in place: x_k = z_k / D_k - l_{k+1} x_{k+1}
            double x = 0.0;
            for (int h0 = npad - HB; h0 >= 0; h0 -= HB) {
                dbl2 *const zb = zl + (h0 >> 1) * P;
                const dbl2 *const pc = reinterpret_cast<const dbl2 *>(pinv + h0);
                double cl[HB + 1];                                // l_{h0 + 1} .. l_{h0 + 32}: an odd offset, read singly
                dbl2 v[HB / 2], cp[HB / 2];
#pragma unroll
                for (int j = 0; j < HB / 2; ++j) {
                    v[j] = zb[j * P];
                    cp[j] = pc[j];
                }
#pragma unroll
                for (int kk = 0; kk < HB; ++kk) cl[kk] = lmul[h0 + kk + 1];
#pragma unroll
                for (int j = HB / 2 - 1; j >= 0; --j) {
                    x = fma(v[j].y, cp[j].y, -cl[2 * j + 1] * x);
                    v[j].y = x;
                    x = fma(v[j].x, cp[j].x, -cl[2 * j] * x);
                    v[j].x = x;
                }
#pragma unroll
                for (int j = 0; j < HB / 2; ++j) zb[j * P] = v[j];
            }
        }
        __syncthreads();
        stamp(3);
        // ---- the solutions out: 16-column pieces of a row, read transposed; wave w its batches
        double *const bl = g.B + rowbase + (long)(r0 + quarter) * g.nt + kk_l;
        for (int bi = wid; bi < nbatch; bi += 4) {
            const int b0 = bi * TS_BATCH;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kq = b0 + 16 * q;
                if (kq >= np) continue;                              // wave-uniform
                const bool kfull = kq + 15 < np, kok = kq + kk_l < np;
#pragma unroll
                for (int i = 0; i < P / 4; ++i) {
                    if (4 * i >= nr) continue;                       // wave-uniform
                    const double v = zbuf[((kq + kk_l) >> 1) * (2 * P) + (quarter + 4 * i) * 2 + (kk_l & 1)];
                    if (kfull && 4 * i + 3 < nr) bl[(long)(4 * i) * g.nt + kq] = v;      // whole piece: no lane mask
                    else if (kok && quarter + 4 * i < nr) bl[(long)(4 * i) * g.nt + kq] = v;
                }
            }
        }
        __syncthreads();                                             // the block is free for the next pass
        stamp(4);
    }
  }
}

static int tridiag_solve_npad(int npmax) { return (npmax + TS_BATCH - 1) / TS_BATCH * TS_BATCH; }
static size_t tridiag_solve_lds(int npmax, int P = TS_P) {
    const int npad = tridiag_solve_npad(npmax);
    return ((size_t)npad * P + 2 * (size_t)(256 + TS_BATCH) + 2 * 256 * 4) * sizeof(double);
}
// trials per pass (64: a lane each) if column blocks of up to npmax fit the kernel's LDS block, else 0 (not supported: the caller keeps
// the eigenvector form).  R < 16 leaves most of the wave's lanes idle: the GEMM form is the better one there.
int k_tridiag_solve_pass(int npmax, int R) {
    return (R >= 16 && npmax <= 256 && tridiag_solve_lds(npmax) <= (size_t)150 * 1024) ? TS_P : 0;
}

void k_tridiag_solve(gpcsd_ctx *c, const double *W, double *B, const double *es, const double *const d[2], const double *const e[2],
                     const double *const amax[2], const double *sig, int nx, int R, int nt, const int np[2], const int c0[2],
                     hipStream_t s) {
    TriSolveArgs g{};
    g.W = W; g.B = B; g.es = es; g.sig = sig; g.nx = nx; g.R = R; g.nt = nt;
    for (int p = 0; p < 2; ++p) {
        g.d[p] = d[p]; g.e[p] = e[p]; g.amax[p] = amax[p]; g.np[p] = np[p]; g.c0[p] = c0[p];
    }
    const int npmax = std::max(np[0], np[1]);
    GP_REQUIRE(k_tridiag_solve_pass(npmax, R) > 0, -3, "tridiag_solve: temporal blocks of %d columns do not fit the solve kernel", npmax);
    g.npad = tridiag_solve_npad(npmax);
    // Trials per pass: 64 (a lane each: one pass for up to 64 trials, 131 KB of LDS and 336 registers per lane: nothing else lives
    // on a CU beside such a workgroup) or 32 (two passes for 50 trials; 66 KB, 193 registers: two per CU, and other streams'
    // workgroups beside them).  Same recurrences per trial, same bits.  The caller says which (gpcsd_ctx::solve_pass): the paired
    // call of a step loop WITHOUT announcements asks for 32 -- the next step's eigen-chains start while this solve runs, and their
    // spatial Gram assembly took 204 us beside the wide form against 91 alone (0.901 against 0.941 ms per cfg3 step; with
    // announcements the chains are not there to be disturbed and the narrow form's second pass costs 2.4 %: 0.811 against 0.792).
    // Up to 32 trials the narrow form is one pass as well and always taken.  GPCSD_TS_P=32|64 forces one (A/B).
    static const int forced = getenv("GPCSD_TS_P") ? (atoi(getenv("GPCSD_TS_P")) == 32 ? 32 : 64) : 0;
    const int P = forced ? forced : (R <= 32 || c->solve_pass == 32) ? 32 : TS_P_DEFAULT;
    const size_t lds = tridiag_solve_lds(npmax, P);
    static size_t attr_dev[64][2] = {};            // (per device: the attribute belongs to the device's copy of the kernel)
    size_t *attr = attr_dev[c->device & 63];
    if (lds > attr[P == 32]) {
        if (P == 32) GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_solve_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_solve_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[P == 32] = lds;
    }
    static const bool clk_on = getenv("GPCSD_TS_CLK") && getenv("GPCSD_TS_CLK")[0] == '1';
    g.clk = clk_on ? c->buf<unsigned long long>("ts_clk", 8) : nullptr;
    ProfScope ps(c, "tridiag_solve", 6.0 * nx * (double)R * nt, s);
    static const int grid_cap = getenv("GPCSD_TS_GRID") ? atoi(getenv("GPCSD_TS_GRID")) : 192;     // (A/B: 0 = one workgroup per item)
    const int grid = grid_cap > 0 ? std::min(2 * nx, grid_cap) : 2 * nx;
    if (P == 32) hipLaunchKernelGGL(tridiag_solve_kernel<32>, dim3(grid), dim3(256), lds, s, g);
    else hipLaunchKernelGGL(tridiag_solve_kernel<64>, dim3(grid), dim3(256), lds, s, g);
    GP_HIP(hipGetLastError());
    if (clk_on) {
        unsigned long long h[8];
        c->copy_out(h, g.clk, sizeof(h), s);
        GP_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[tridiag_solve] item 0 (10 ns ticks): load+pivots %llu  forward %llu  backward %llu  store %llu\n", h[1] - h[0],
                h[2] - h[1], h[3] - h[2], h[4] - h[3]);
    }
}

// A(nx, ngl) = gl_w[g] * b_fwd_1d(gl_x[g] - x[i], R)                      covariances.py:86-88
template <typename T>
__global__ __launch_bounds__(256) void fwd_weights_1d_kernel(const double *__restrict__ x, int nx,
                                                             const double *__restrict__ gl_x,
                                                             const double *__restrict__ gl_w, int ngl, double R,
                                                             double *__restrict__ A, const HpDev *__restrict__ tab, long s_out) {
    if (tab) {
        R = tab[blockIdx.z].R;
        A += blockIdx.z * s_out;
    }
    const long n = (long)nx * ngl;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / ngl), g = (int)(e % ngl);
        A[e] = (double)(T(gl_w[g]) * dev_b_fwd_1d<T>(T(gl_x[g]) - T(x[i]), T(R)));
    }
}

void k_fwd_weights_1d(gpcsd_ctx *c, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl, double R,
                      double *A, hipStream_t s, const HpDev *tab, int B, long s_out) {
    const dim3 grid(ew_grid((long)nx * ngl), 1, tab ? B : 1);
    if (c->gram_fp32)
        hipLaunchKernelGGL(fwd_weights_1d_kernel<float>, grid, dim3(256), 0, s, x, nx, gl_x, gl_w, ngl, R, A, tab, s_out);
    else
        hipLaunchKernelGGL(fwd_weights_1d_kernel<double>, grid, dim3(256), 0, s, x, nx, gl_x, gl_w, ngl, R, A, tab, s_out);
    GP_HIP(hipGetLastError());
}

// A(nx, G) with g = g1*ngl2 + g2 (expand_grid order, utility_functions.py:22):
//   gl_w_prod[g] * b_fwd_2d(w = |gl_g - x_i|)                             covariances.py:125-131, :220-221
template <typename T>
__global__ __launch_bounds__(256) void fwd_weights_2d_kernel(const double *__restrict__ xy, int nx,
                                                             const double *__restrict__ gx1,
                                                             const double *__restrict__ gw1, int ngl1,
                                                             const double *__restrict__ gx2,
                                                             const double *__restrict__ gw2, int ngl2, double R, double eps,
                                                             double *__restrict__ A, const HpDev *__restrict__ tab, long s_out) {
    if (tab) {
        R = tab[blockIdx.z].R;
        eps = tab[blockIdx.z].eps;
        A += blockIdx.z * s_out;
    }
    const int G = ngl1 * ngl2;
    const int i = blockIdx.y;
    const T x1 = T(xy[2 * i]), x2 = T(xy[2 * i + 1]);
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
        const int g1 = g / ngl2, g2 = g % ngl2;
        const T d1 = T(gx1[g1]) - x1, d2 = T(gx2[g2]) - x2;
        const T w = sqrt(d1 * d1 + d2 * d2);
        A[(long)i * G + g] = (double)((T(gw1[g1]) * T(gw2[g2])) * dev_b_fwd_2d_w<T>(w, T(R), T(eps)));
    }
}

void k_fwd_weights_2d(gpcsd_ctx *c, const double *xy, int nx, const double *gx1, const double *gw1, int ngl1, const double *gx2,
                      const double *gw2, int ngl2, double R, double eps, double *A, hipStream_t s, const HpDev *tab, int B,
                      long s_out) {
    const int G = ngl1 * ngl2;
    dim3 grid(ceil_div(G, 256), nx, tab ? B : 1);
    ProfScope ps(c, "fwd_weights_2d", 0.0, s);
    if (c->gram_fp32)
        hipLaunchKernelGGL(fwd_weights_2d_kernel<float>, grid, dim3(256), 0, s, xy, nx, gx1, gw1, ngl1, gx2, gw2, ngl2, R, eps, A,
                           tab, s_out);
    else
        hipLaunchKernelGGL(fwd_weights_2d_kernel<double>, grid, dim3(256), 0, s, xy, nx, gx1, gw1, ngl1, gx2, gw2, ngl2, R, eps, A,
                           tab, s_out);
    GP_HIP(hipGetLastError());
}

// out(n,m) = exp(-0.5 ((a_i - b_j)/ell)^2)      covariances.py:55, :67, :89
template <typename T>
__global__ __launch_bounds__(256) void se_1d_kernel(const double *__restrict__ a, int n, const double *__restrict__ b, int m,
                                                    double ell, double *__restrict__ out, const HpDev *__restrict__ tab,
                                                    long s_out) {
    if (tab) {
        ell = tab[blockIdx.z].ell_s[0];
        out += blockIdx.z * s_out;
    }
    __shared__ T sa[PT_ROWS], sb[PT_COLS];
    const int c0 = blockIdx.x * PT_COLS, r0 = blockIdx.y * PT_ROWS;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x < PT_COLS) sb[threadIdx.x] = (c0 + threadIdx.x < m) ? T(b[c0 + threadIdx.x]) : T(0);
    else if (threadIdx.x < PT_COLS + PT_ROWS) {
        int i = threadIdx.x - PT_COLS;
        sa[i] = (r0 + i < n) ? T(a[r0 + i]) : T(0);
    }
    __syncthreads();
    const int col = c0 + tx;
    if (col >= m) return;
#pragma unroll
    for (int k = 0; k < PT_ROWS / 4; ++k) {
        const int rr = ty + 4 * k, row = r0 + rr;
        if (row >= n) continue;
        const T q = (sa[rr] - sb[tx]) / T(ell);
        out[(long)row * m + col] = (double)exp(T(-0.5) * (q * q));
    }
}

void k_se_1d(gpcsd_ctx *c, const double *a, int n, const double *b, int m, double ell, double *out, hipStream_t s,
             const HpDev *tab, int B, long s_out) {
    dim3 grid(ceil_div(m, PT_COLS), ceil_div(n, PT_ROWS), tab ? B : 1);
    if (c->gram_fp32) hipLaunchKernelGGL(se_1d_kernel<float>, grid, dim3(256), 0, s, a, n, b, m, ell, out, tab, s_out);
    else hipLaunchKernelGGL(se_1d_kernel<double>, grid, dim3(256), 0, s, a, n, b, m, ell, out, tab, s_out);
    GP_HIP(hipGetLastError());
}

// Anisotropic SE between 2D point sets.  A set is either a tensor grid (n2 > 0: point i = (p1[i / n2], p2[i % n2]))
// or an explicit (n,2) list (n2 == 0: p1 = base, point i = (p1[2i], p1[2i+1])).
// Both sets grids -> exp(-0.5 d1^2 / ell1^2) * exp(-0.5 d2^2 / ell2^2)          covariances.py:216
// otherwise       -> exp(-0.5 (d1/ell1)^2)   * exp(-0.5 (d2/ell2)^2)            covariances.py:186, :199
template <typename T>
__global__ __launch_bounds__(256) void se_2d_kernel(const double *__restrict__ a1, const double *__restrict__ a2, int na,
                                                    int na2, const double *__restrict__ b1, const double *__restrict__ b2,
                                                    int nb, int nb2, double ell1d, double ell2d, double *__restrict__ out,
                                                    const HpDev *__restrict__ tab, long s_out) {
    if (tab) {
        ell1d = tab[blockIdx.z].ell_s[0];
        ell2d = tab[blockIdx.z].ell_s[1];
        out += blockIdx.z * s_out;
    }
    __shared__ T sa[2][PT_ROWS], sb[2][PT_COLS];
    const T ell1 = T(ell1d), ell2 = T(ell2d);
    const int c0 = blockIdx.x * PT_COLS, r0 = blockIdx.y * PT_ROWS;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x < PT_COLS) {
        const int j = c0 + threadIdx.x;
        double v1 = 0.0, v2 = 0.0;
        if (j < nb) {
            if (nb2 > 0) { v1 = b1[j / nb2]; v2 = b2[j % nb2]; }
            else { v1 = b1[2 * j]; v2 = b1[2 * j + 1]; }
        }
        sb[0][threadIdx.x] = T(v1);
        sb[1][threadIdx.x] = T(v2);
    } else if (threadIdx.x < PT_COLS + PT_ROWS) {
        const int ii = threadIdx.x - PT_COLS, i = r0 + ii;
        double v1 = 0.0, v2 = 0.0;
        if (i < na) {
            if (na2 > 0) { v1 = a1[i / na2]; v2 = a2[i % na2]; }
            else { v1 = a1[2 * i]; v2 = a1[2 * i + 1]; }
        }
        sa[0][ii] = T(v1);
        sa[1][ii] = T(v2);
    }
    __syncthreads();
    const int col = c0 + tx;
    if (col >= nb) return;
    const bool grid_form = (na2 > 0) && (nb2 > 0);
#pragma unroll
    for (int k = 0; k < PT_ROWS / 4; ++k) {
        const int rr = ty + 4 * k, row = r0 + rr;
        if (row >= na) continue;
        const T d1 = sa[0][rr] - sb[0][tx], d2 = sa[1][rr] - sb[1][tx];
        T v;
        if (grid_form) v = exp(T(-0.5) * (d1 * d1) / (ell1 * ell1)) * exp(T(-0.5) * (d2 * d2) / (ell2 * ell2));
        else {
            const T q1 = d1 / ell1, q2 = d2 / ell2;
            v = exp(T(-0.5) * (q1 * q1)) * exp(T(-0.5) * (q2 * q2));
        }
        out[(long)row * nb + col] = (double)v;
    }
}

void k_se_2d(gpcsd_ctx *c, const double *a1, const double *a2, int na, int na2, const double *b1, const double *b2, int nb,
             int nb2, double ell1, double ell2, double *out, hipStream_t s, const HpDev *tab, int B, long s_out) {
    dim3 grid(ceil_div(nb, PT_COLS), ceil_div(na, PT_ROWS), tab ? B : 1);
    ProfScope ps(c, "gram_se_2d", 0.0, s);
    if (c->gram_fp32)
        hipLaunchKernelGGL(se_2d_kernel<float>, grid, dim3(256), 0, s, a1, a2, na, na2, b1, b2, nb, nb2, ell1, ell2, out, tab, s_out);
    else
        hipLaunchKernelGGL(se_2d_kernel<double>, grid, dim3(256), 0, s, a1, a2, na, na2, b1, b2, nb, nb2, ell1, ell2, out, tab, s_out);
    GP_HIP(hipGetLastError());
}

// One factor of the tensor-grid SE kernel: out(n,n) = exp(-0.5 (a_i - a_j)^2 / ell^2), the per-axis term of the grid form of
// se_2d_kernel (covariances.py:216: Kgl = exp(-0.5 d1^2/ell1^2) * exp(-0.5 d2^2/ell2^2) = K1 (x) K2 on the GL tensor grid).
template <typename T>
__global__ __launch_bounds__(256) void se_axis_kernel(const double *__restrict__ a, int n, double elld, double *__restrict__ out) {
    const T ell = T(elld);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
        const T d = T(a[e / n]) - T(a[e % n]);
        out[e] = (double)exp(T(-0.5) * (d * d) / (ell * ell));
    }
}
void k_se_axis(gpcsd_ctx *c, const double *a, int n, double ell, double *out, hipStream_t s) {
    const int grid = ceil_div((long)n * n, 256);
    if (c->gram_fp32) hipLaunchKernelGGL(se_axis_kernel<float>, dim3(grid), dim3(256), 0, s, a, n, ell, out);
    else hipLaunchKernelGGL(se_axis_kernel<double>, dim3(grid), dim3(256), 0, s, a, n, ell, out);
    GP_HIP(hipGetLastError());
}

// The same factor and its derivative w.r.t. the length scale for every hyper-parameter set of a batched evaluation
// (blockIdx.y = set, ell = tab[set].ell_s[axis]): K = exp(-0.5 d^2 / ell^2), dK = K d^2 / ell^3 -- the per-axis pieces of
// dKgl/dell1 = dK1 (x) K2 and dKgl/dell2 = K1 (x) dK2 (the gradient's Kronecker form, grad.hip: k_frob_pair).
template <typename T>
__global__ __launch_bounds__(256) void se_axis_tab_kernel(const double *__restrict__ a, int n, int axis, const HpDev *__restrict__ tab,
                                                          double *__restrict__ K, double *__restrict__ dK) {
    const T ell = T(tab[blockIdx.y].ell_s[axis]);
    K += (long)blockIdx.y * n * n;
    dK += (long)blockIdx.y * n * n;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
        const T d = T(a[e / n]) - T(a[e % n]);
        const T k = exp(T(-0.5) * (d * d) / (ell * ell));
        K[e] = (double)k;
        dK[e] = (double)(k * (d * d) / (ell * ell * ell));
    }
}
void k_se_axis_tab(gpcsd_ctx *c, const double *a, int n, int axis, const HpDev *tab, int B, double *K, double *dK, hipStream_t s) {
    const int grid = ceil_div((long)n * n, 256);
    if (c->gram_fp32) hipLaunchKernelGGL(se_axis_tab_kernel<float>, dim3(grid, B), dim3(256), 0, s, a, n, axis, tab, K, dK);
    else hipLaunchKernelGGL(se_axis_tab_kernel<double>, dim3(grid, B), dim3(256), 0, s, a, n, axis, tab, K, dK);
    GP_HIP(hipGetLastError());
}

__global__ void add_diag_kernel(double *A, int n, double v, const HpDev *__restrict__ tab, long s_out) {
    if (tab) {
        v = tab[blockIdx.z].jitter;
        A += blockIdx.z * s_out;
        if (v == 0.0) return;
    }
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(long)i * n + i] += v;
}
// dst = src + v (a spectrum shifted by a multiple of the identity: the paired call's shared spatial side)
__global__ void shift_copy_kernel(const double *__restrict__ src, double *__restrict__ dst, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + v;
}
void k_shift_copy(gpcsd_ctx *c, const double *src, double *dst, int n, double v, hipStream_t s) {
    hipLaunchKernelGGL(shift_copy_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, src, dst, n, v);
    GP_HIP(hipGetLastError());
}

void k_add_diag(gpcsd_ctx *c, double *A, int n, double v, hipStream_t s, const HpDev *tab, int B, long s_out) {
    hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(n, 256), 1, tab ? B : 1), dim3(256), 0, s, A, n, v, tab, s_out);
    GP_HIP(hipGetLastError());
}

// out[i] = ((P_0[i] + P_1[i]) + P_2[i]) + ..: the partial products of a GEMM whose K range was split over workgroups (fixed order)
__global__ __launch_bounds__(256) void sum_partials_kernel(double *__restrict__ out, const double *__restrict__ P, long n, int parts) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        double acc = P[i];
        for (int q = 1; q < parts; ++q) acc += P[q * n + i];
        out[i] = acc;
    }
}
void k_sum_partials(gpcsd_ctx *c, double *out, const double *P, long n, int parts, hipStream_t s) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3((int)std::min<long>(ceil_div(n, 256L), 1024)), dim3(256), 0, s, out, P, n, parts);
    GP_HIP(hipGetLastError());
}

__global__ void fill_kernel(double *p, long n, double v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
void k_fill(gpcsd_ctx *c, double *p, long n, double v, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, s, p, n, v);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// D = repeat(es, nt) * tile(et, nx) + sig2n_vec  and  sum(log D)          utility_functions.py:54-63, gpcsd1d.py:122
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_D_kernel(const double *__restrict__ es, int nx, const double *__restrict__ et,
                                                      int nt, const double *__restrict__ sig, int nsig,
                                                      double *__restrict__ D, double *__restrict__ Dinv,
                                                      double *__restrict__ partials, const HpDev *__restrict__ tab) {
    const long n = (long)nx * nt;
    if (tab) {                       // blockIdx.y = hyper-parameter set: spectra nx / nt apart, D n apart, scalar noise from tab
        const long b = blockIdx.y;   // (or, with a per-electrode list: set b's nx entries of `sig`)
        es += b * nx;
        et += b * nt;
        if (nsig == 1) sig = &tab[b].sig2n;
        else sig += b * nx;
        D += b * n;
        if (Dinv) Dinv += b * n;
        partials += b * 256;
    }
    double s = 0.0;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e / nt), i = (int)(e % nt);
        const double d = es[x] * et[i] + (nsig == 1 ? sig[0] : sig[x]);
        D[e] = d;
        if (Dinv) Dinv[e] = 1.0 / d;             // the GEMM epilogues multiply by this instead of dividing 16x per lane and tile
        s += log(d);
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// blockIdx.x = segment: p 256 apart, out ostride apart
__global__ __launch_bounds__(256) void sum_small_kernel(const double *__restrict__ p, int n, double *out, long ostride = 0) {
    p += blockIdx.x * 256L;
    out += blockIdx.x * ostride;
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

int k_build_D(gpcsd_ctx *c, const double *es, int nx, const double *et, int nt, const double *sig, int nsig, double *D,
              double *Dinv, double *sumlog_out, hipStream_t s, const HpDev *tab, int B, long s_sumlog) {
    const long n = (long)nx * nt;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int nb = tab ? B : 1;
    double *part = c->buf<double>("buildD_partials", (size_t)256 * nb);
    ProfScope ps(c, "build_D_logdet", 0.0, s);
    hipLaunchKernelGGL(build_D_kernel, dim3(blocks, nb), dim3(256), 0, s, es, nx, et, nt, sig, nsig, D, Dinv, part, tab);
    if (sumlog_out)
        hipLaunchKernelGGL(sum_small_kernel, dim3(nb), dim3(256), 0, s, (const double *)part, blocks, sumlog_out, s_sumlog);
    GP_HIP(hipGetLastError());
    return blocks;
}

// ------------------------------------------------------------------------------------------------
// [n0][n1][n2] -> [n0][n2][n1]   (lfp (x,t,r) -> (x,r,t) on upload; predictions (z,r,t) -> (z,t,r) on download)
// LDS-tiled 32x32 transpose, both sides coalesced.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swap_last2_kernel(const double *__restrict__ in, double *__restrict__ out, int n1,
                                                         int n2) {
    __shared__ double tile[32][33];
    const long base = (long)blockIdx.z * n1 * n2;
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;       // i over n1, j over n2
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + ty + 8 * k, j = j0 + tx;
        if (i < n1 && j < n2) tile[ty + 8 * k][tx] = in[base + (long)i * n2 + j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + ty + 8 * k, i = i0 + tx;
        if (i < n1 && j < n2) out[base + (long)j * n1 + i] = tile[tx][ty + 8 * k];
    }
}

void k_swap_last2(gpcsd_ctx *c, const double *in, double *out, int n0, int n1, int n2, hipStream_t s) {
    dim3 grid(ceil_div(n2, 32), ceil_div(n1, 32), n0);
    ProfScope ps(c, "relayout", 0.0, s);
    hipLaunchKernelGGL(swap_last2_kernel, grid, dim3(256), 0, s, in, out, n1, n2);
    GP_HIP(hipGetLastError());
}

// Predictions of all C temporal components arrive from ONE GEMM as in[(z, r)][c*n2 + t] (row stride C*n2).  One pass
// writes every component in the reference's (z, t, r) layout and their sum: per (z, c) a 32x32 LDS-tiled transpose of the
// (r, t) block, the sum accumulated in registers over c.  list may be null (sum only).
__global__ __launch_bounds__(256) void swap_last2_sum_kernel(const double *__restrict__ in, int C, double *__restrict__ list,
                                                             long list_stride, double *__restrict__ sum, int n1, int n2) {
    __shared__ double tile[32][33];
    const long zin = (long)blockIdx.z * n1 * C * n2, zout = (long)blockIdx.z * n1 * n2;
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;       // i over n1 (trials), j over n2 (times)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < C; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + ty + 8 * k, j = j0 + tx;
            if (i < n1 && j < n2) tile[ty + 8 * k][tx] = in[zin + (long)i * C * n2 + (long)c * n2 + j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + ty + 8 * k, i = i0 + tx;
            if (i < n1 && j < n2) {
                const double v = tile[tx][ty + 8 * k];
                if (list) list[(long)c * list_stride + zout + (long)j * n1 + i] = v;
                acc[k] += v;                                    // components summed in index order, as the reference does
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + ty + 8 * k, i = i0 + tx;
        if (i < n1 && j < n2) sum[zout + (long)j * n1 + i] = acc[k];
    }
}

void k_swap_last2_sum(gpcsd_ctx *c, const double *in, int C, double *list, long list_stride, double *sum, int n0, int n1, int n2,
                      hipStream_t s) {
    dim3 grid(ceil_div(n2, 32), ceil_div(n1, 32), n0);
    ProfScope ps(c, "relayout", 0.0, s);
    hipLaunchKernelGGL(swap_last2_sum_kernel, grid, dim3(256), 0, s, in, C, list, list_stride, sum, n1, n2);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Folded-basis helpers.  With mirror-symmetric electrode and time grids every covariance of the model commutes with the
// reflections, so in the basis of symmetric / antisymmetric combinations (F = fold operator, orthogonal) all of them are
// block diagonal and the flat GEMMs of loglik / predict split into two half-size products each (half the flops).  Fold
// order of an index: the ns symmetric orbits first (pairs, then fixed points), then the na antisymmetric pairs.
// ------------------------------------------------------------------------------------------------
// out_ss[a][b] = wa wb sum_{r in orbit a} sum_{c in orbit b} K[r][c];  out_aa[a][b] = 1/2 (K_ik - K_il - K_jk + K_jl)
__global__ __launch_bounds__(256) void sym_fold_rect_kernel(const double *__restrict__ K, long ldk, SymDev rs, SymDev cs,
                                                            double *__restrict__ out_ss, double *__restrict__ out_aa) {
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= (long)rs.ns * cs.ns) return;
    const int a = (int)(e / cs.ns), b = (int)(e % cs.ns);
    const int i = rs.rep_i[a], j = rs.rep_j[a], k = cs.rep_i[b], l = cs.rep_j[b];
    const double kik = K[(long)i * ldk + k];
    double ssum = kik, asum = kik;
    if (l != k) {
        const double v = K[(long)i * ldk + l];
        ssum += v;
        asum -= v;
    }
    if (j != i) {
        const double v = K[(long)j * ldk + k];
        ssum += v;
        asum -= v;
        if (l != k) {
            const double v2 = K[(long)j * ldk + l];
            ssum += v2;
            asum += v2;
        }
    }
    const double isq2 = 0.70710678118654752440;
    const double wa = (i == j) ? 1.0 : isq2, wb = (k == l) ? 1.0 : isq2;
    out_ss[e] = wa * wb * ssum;
    if (a < rs.na && b < cs.na) out_aa[(long)a * cs.na + b] = 0.5 * asum;
}

void k_sym_fold_rect(gpcsd_ctx *c, const double *K, long ldk, const SymDev &rs, const SymDev &cs, double *out_ss, double *out_aa,
                     hipStream_t s) {
    (void)c;
    hipLaunchKernelGGL(sym_fold_rect_kernel, dim3(ceil_div((long)rs.ns * cs.ns, 256)), dim3(256), 0, s, K, ldk, rs, cs, out_ss,
                       out_aa);
    GP_HIP(hipGetLastError());
}

// Inverse of the fold for a matrix that is block diagonal in the folded basis: G = F^T diag(Gss, Gaa) F, i.e.
//   G[i][j] = w_i w_j Gss[orb_i][orb_j] + (sgn_i sgn_j / 2) Gaa[orb_i][orb_j]      (w = 1/sqrt2 for pair members, 1 for fixed points)
// blockIdx.y = hyper-parameter set: inputs s_in apart (Gaa follows Gss), outputs n*n apart.
__global__ __launch_bounds__(256) void sym_unfold_mat_kernel(const double *__restrict__ Gf, long s_in, SymDev sy, int n,
                                                             double *__restrict__ out) {
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= (long)n * n) return;
    const double *__restrict__ Gss = Gf + blockIdx.y * s_in, *__restrict__ Gaa = Gss + (long)sy.ns * sy.ns;
    const int i = (int)(e / n), j = (int)(e % n);
    const int a = sy.orb[i], b = sy.orb[j], si = sy.sgn[i], sj = sy.sgn[j];
    const double isq2 = 0.70710678118654752440;
    double v = ((si == 0) ? 1.0 : isq2) * ((sj == 0) ? 1.0 : isq2) * Gss[(long)a * sy.ns + b];
    if (si != 0 && sj != 0) v += 0.5 * (double)(si * sj) * Gaa[(long)a * sy.na + b];
    out[blockIdx.y * (long)n * n + e] = v;
}
void k_sym_unfold_mat(gpcsd_ctx *c, const double *Gf, long s_in, const SymDev &sy, int n, double *out, hipStream_t s, int B) {
    (void)c;
    hipLaunchKernelGGL(sym_unfold_mat_kernel, dim3(ceil_div((long)n * n, 256), B), dim3(256), 0, s, Gf, s_in, sy, n, out);
    GP_HIP(hipGetLastError());
}

// Y[x][r][t] -> out[q][r][b], q / b = fold index of the electrode / the time point
__global__ __launch_bounds__(256) void fold_lfp_kernel(const double *__restrict__ Y, int nx, int R, int nt, SymDev ss, SymDev st,
                                                       double *__restrict__ out) {
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= (long)nx * R * nt) return;
    const int b = (int)(e % nt);
    const long qr = e / nt;
    const int r = (int)(qr % R), q = (int)(qr / R);
    const bool qa = q >= ss.ns, ba = b >= st.ns;
    const int a = qa ? q - ss.ns : q, bb = ba ? b - st.ns : b;
    const int i = ss.rep_i[a], j = ss.rep_j[a], k = st.rep_i[bb], l = st.rep_j[bb];
    const double isq2 = 0.70710678118654752440;
    const double wq = (i == j) ? 1.0 : isq2, wb = (k == l) ? 1.0 : isq2;
    const double sj = qa ? -1.0 : 1.0, sl = ba ? -1.0 : 1.0;
    const double *__restrict__ yi = Y + ((long)i * R + r) * nt, *__restrict__ yj = Y + ((long)j * R + r) * nt;
    double v = yi[k];
    if (l != k) v += sl * yi[l];
    if (j != i) {
        double u = yj[k];
        if (l != k) u += sl * yj[l];
        v += sj * u;
    }
    out[e] = wq * wb * v;
}

void k_fold_lfp(gpcsd_ctx *c, const double *Y, int nx, int R, int nt, const SymDev &ss, const SymDev &st, double *out,
                hipStream_t s) {
    ProfScope ps(c, "fold_lfp", 0.0, s);
    hipLaunchKernelGGL(fold_lfp_kernel, dim3(ceil_div((long)nx * R * nt, 256)), dim3(256), 0, s, Y, nx, R, nt, ss, st, out);
    GP_HIP(hipGetLastError());
}

// Predictions in the folded basis, in[(zq, r)][pt][c][b] (zq = fold index of the site; pt = 0: the C symmetric time blocks of
// width st.ns, pt = 1: the C antisymmetric blocks of width st.na; row stride C*nt) -> list[c][z][t][r] (optional) and
// sum[z][t][r].  One workgroup owns a site orbit x 16 time orbits x 64 trials: it reads the four parity tiles once and
// writes the (up to) four mirror images, transposing (r, t) -> (t, r) through LDS.  The outputs have the trial index
// innermost (R = 50: 400-byte rows, consecutive time points back to back), so a wave writes ONE WHOLE ROW per store and
// the workgroup's sixteen consecutive time points form one contiguous run per mirror image -- the 32-trial tiles of the
// first version cut every row in two pieces owned by different workgroups (partial cache lines: 4.2 TB/s; now 4.9 TB/s).
constexpr int UF_TB = 16, UF_TR = 64;
__global__ __launch_bounds__(256) void unfold_swap_sum_kernel(const double *__restrict__ in, int C, double *__restrict__ list,
                                                              long list_stride, double *__restrict__ sum, int R, int nt,
                                                              SymDev sz, SymDev st, int nsP, int naP, long ldin) {
    __shared__ double tile[2][2][UF_TR][UF_TB + 1];
    const int az = blockIdx.z;                                  // site orbit
    const int b0 = blockIdx.x * UF_TB, i0 = blockIdx.y * UF_TR; // time orbits, trials
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;     // load phase: 16 time orbits x 16 trials per pass
    const int wi = threadIdx.x & 63, wb = threadIdx.x >> 6;     // write phase: 64 trials x 4 time orbits per pass
    const int zi = sz.rep_i[az], zj = sz.rep_j[az];
    const double isq2 = 0.70710678118654752440;
    const double wz = (zi == zj) ? 1.0 : isq2;
    // (column blocks padded to nsP / naP columns -- 128-byte aligned blocks --, rows ldin apart)
    double acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[m][k] = 0.0;
    for (int c = 0; c < C; ++c) {
        __syncthreads();
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const bool zok = pz == 0 || az < sz.na;
                const long row = pz ? (long)sz.ns + az : az;
                const int wdt = pt ? st.na : st.ns;
                const long col0 = (pt ? (long)C * nsP : 0) + (long)c * (pt ? naP : nsP);
#pragma unroll
                for (int k = 0; k < UF_TR / 16; ++k) {
                    const int i = i0 + ly + 16 * k, bb = b0 + lx;
                    double v = 0.0;
                    if (zok && i < R && bb < wdt) v = in[(row * R + i) * ldin + col0 + bb];
                    tile[pz][pt][ly + 16 * k][lx] = v;
                }
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bl = wb + 4 * k, bb = b0 + bl, i = i0 + wi;
            if (i < R && bb < st.ns) {
                const int tk = st.rep_i[bb], tl = st.rep_j[bb];
                const double wt = (tk == tl) ? 1.0 : isq2;
                const double ss = wz * wt * tile[0][0][wi][bl], sa = wz * isq2 * tile[0][1][wi][bl];
                const double as = isq2 * wt * tile[1][0][wi][bl], aa = 0.5 * tile[1][1][wi][bl];
                const double v0 = (ss + sa) + (as + aa);        // (zi, tk)
                const double v1 = (ss - sa) + (as - aa);        // (zi, tl)
                const double v2 = (ss + sa) - (as + aa);        // (zj, tk)
                const double v3 = (ss - sa) - (as - aa);        // (zj, tl)
                acc[0][k] += v0;                                // components summed in index order, as the reference does
                acc[1][k] += v1;
                acc[2][k] += v2;
                acc[3][k] += v3;
                if (list) {
                    double *__restrict__ lc = list + (long)c * list_stride;
                    lc[((long)zi * nt + tk) * R + i] = v0;
                    if (tl != tk) lc[((long)zi * nt + tl) * R + i] = v1;
                    if (zj != zi) {
                        lc[((long)zj * nt + tk) * R + i] = v2;
                        if (tl != tk) lc[((long)zj * nt + tl) * R + i] = v3;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int bb = b0 + wb + 4 * k, i = i0 + wi;
        if (i < R && bb < st.ns) {
            const int tk = st.rep_i[bb], tl = st.rep_j[bb];
            sum[((long)zi * nt + tk) * R + i] = acc[0][k];
            if (tl != tk) sum[((long)zi * nt + tl) * R + i] = acc[1][k];
            if (zj != zi) {
                sum[((long)zj * nt + tk) * R + i] = acc[2][k];
                if (tl != tk) sum[((long)zj * nt + tl) * R + i] = acc[3][k];
            }
        }
    }
}

void k_unfold_swap_sum(gpcsd_ctx *c, const double *in, int C, double *list, long list_stride, double *sum, int R, int nt,
                       const SymDev &sz, const SymDev &st, hipStream_t s, int nsP, int naP, long ldin) {
    if (nsP <= 0) nsP = st.ns;                                  // unpadded input: blocks of st.ns / st.na columns
    if (naP <= 0) naP = st.na;
    if (ldin <= 0) ldin = (long)C * (nsP + naP);
    dim3 grid(ceil_div(st.ns, UF_TB), ceil_div(R, UF_TR), sz.ns);
    ProfScope ps(c, "relayout", 0.0, s);
    hipLaunchKernelGGL(unfold_swap_sum_kernel, grid, dim3(256), 0, s, in, C, list, list_stride, sum, R, nt, sz, st, nsP, naP, ldin);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
