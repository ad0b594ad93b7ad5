// Symmetric eigensolver front end -- replaces numpy.linalg.eigh at utility_functions.py:58-59.
//
//   n <= 64 : two-sided cyclic Jacobi, one workgroup, A and V resident in LDS (the nx = 24 spatial problems of
//             GPCSD1D; also the leaf solver of the divide-and-conquer path).
//   n  > 64 : Householder tridiagonalisation + divide & conquer + compact-WY back-transformation (eigh_dc.hip,
//             stedc.hip), all requested problems sharing every launch, replayed as a hipGraph.
//   symmetry: when the caller knows an involutive permutation P with P K P = K (a uniform time grid makes Kt
//             centro-symmetric; a probe that is symmetric about the centre of the integration box does the same for
//             Ks), K is block-diagonalised by the (e_i +- e_Pi)/sqrt(2) basis into a symmetric and an antisymmetric
//             half-size problem.  The tridiagonalisation is launch-latency bound (one dependent launch per column), so
//             halving n halves its time; the eigenvalues are the union of the two spectra, merged in ascending order.
// GPCSD_EIGH=jacobi forces the single-workgroup Jacobi on global memory for any n (slow; independent cross-check).
// GPCSD_NO_GRAPH=1 disables graph replay, GPCSD_NO_SYMFOLD=1 disables the symmetry folding.
#include <algorithm>
#include <cstdlib>

#include "devutil.hpp"
#include "jacobi.hpp"
#include "kernels.hpp"

namespace gpcsd {

void eigh_large_multi(gpcsd_ctx *c, const EigReq *reqs, int nclass, int *d_status, int status_stride, hipStream_t s,
                      int stage = 0);   // eigh_dc.hip

// blockIdx.x = replica: inputs sA apart, eigenvalues sw apart, eigenvectors sZ apart, status words status_stride apart
template <int NT>
__global__ __launch_bounds__(NT) void jacobi_lds_kernel(const double *__restrict__ Ag, int n, double *evals, double *evecs,
                                                        int *status, long sA, long sw, long sZ, int status_stride) {
    Ag += blockIdx.x * sA;
    evals += blockIdx.x * sw;
    evecs += blockIdx.x * sZ;
    status += (long)blockIdx.x * status_stride;
    extern __shared__ double smem[];
    const int ld = n | 1;
    double *A = smem;
    double *V = A + n * ld;
    double *cs = V + n * ld;
    double *red = cs + 2 * (JACOBI_LDS_MAX / 2 + 1);
    __shared__ int pq[2 * JACOBI_LDS_MAX + 2];
    for (int e = threadIdx.x; e < n * n; e += NT) A[(e / n) * ld + (e % n)] = Ag[e];
    __syncthreads();
    jacobi_body<NT>(A, ld, V, ld, n, evals, evecs, n, status, cs, pq, red);
}

template <int NT>
__global__ __launch_bounds__(NT) void jacobi_global_kernel(double *A, double *V, int n, double *evals, double *evecs,
                                                           int *status) {
    __shared__ double cs[JACOBI_MAX_N + 2];
    __shared__ int pq[JACOBI_MAX_N + 2];
    __shared__ double red[JACOBI_MAX_N > NT ? JACOBI_MAX_N : NT];
    jacobi_body<NT>(A, n, V, n, n, evals, evecs, n, status, cs, pq, red);
}

static bool env_flag(const char *name, const char *value) {
    const char *e = getenv(name);
    return e && std::string(e) == value;
}
constexpr int EIG_BATCH_MIN_N = 16;      // smallest problem sent through the tridiagonalisation + D&C pipeline
static bool force_jacobi() {
    static const bool v = env_flag("GPCSD_EIGH", "jacobi");
    return v;
}
static bool graphs_enabled() {
    static const bool v = !env_flag("GPCSD_NO_GRAPH", "1");
    return v;
}
static bool symfold_enabled() {
    static const bool v = !env_flag("GPCSD_NO_SYMFOLD", "1");
    return v;
}

static void eigh_jacobi(gpcsd_ctx *c, const EigReq &q, int *d_status, int status_stride, hipStream_t s) {
    const int n = q.n, count = q.count > 1 ? q.count : 1;
    ProfScope ps(c, "eigh_jacobi", 9.0 * (double)n * n * n * count, s);
    if (n <= JACOBI_LDS_MAX) {
        const int ld = n | 1;
        const size_t sh = sizeof(double) * (2 * (size_t)n * ld + 2 * (JACOBI_LDS_MAX / 2 + 1) + 256);
        hipLaunchKernelGGL((jacobi_lds_kernel<256>), dim3(count), dim3(256), sh, s, (const double *)q.A, n, q.w, q.Z, d_status,
                           q.sA, q.sw, q.sZ, status_stride);
    } else {                                   // GPCSD_EIGH=jacobi diagnostics path: one replica at a time
        double *V = c->buf<double>(std::string("eigh_V_") + (q.tag ? q.tag : ""), (size_t)n * n);
        for (int r = 0; r < count; ++r)
            hipLaunchKernelGGL((jacobi_global_kernel<1024>), dim3(1), dim3(1024), 0, s, q.A + r * q.sA, V, n, q.w + r * q.sw,
                               q.Z + r * q.sZ, d_status + (long)r * status_stride);
    }
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// symmetry folding
// ------------------------------------------------------------------------------------------------------------------
// orbit a < ns has representatives (rep_i[a], rep_j[a]); pairs come first (a < na), fixed points have rep_i == rep_j.
//   Ksym[a][b] = wa wb sum_{r in {i,j}} sum_{c in {k,l}} K[r][c]   (w = 1/sqrt2 for a pair, 1 for a fixed point)
//   Kanti[a][b] = 1/2 (K[i][k] - K[i][l] - K[j][k] + K[j][l])      (pairs only)
// blockIdx.y = replica (K n*n apart, Ksym ns*ns apart, Kanti na*na apart)
__global__ __launch_bounds__(256) void sym_fold_kernel(const double *__restrict__ K, int n, SymDev sy, double *__restrict__ Ksym,
                                                       double *__restrict__ Kanti) {
    const long e = blockIdx.x * 256L + threadIdx.x;
    const int ns = sy.ns, na = sy.na;
    K += (long)blockIdx.y * n * n;
    Ksym += (long)blockIdx.y * ns * ns;
    Kanti += (long)blockIdx.y * na * na;
    if (e >= (long)ns * ns) return;
    const int a = (int)(e / ns), b = (int)(e % ns);
    const int i = sy.rep_i[a], j = sy.rep_j[a], k = sy.rep_i[b], l = sy.rep_j[b];
    const double kik = K[(long)i * n + k];
    double ssum = kik, asum = kik;
    if (l != k) {
        const double v = K[(long)i * n + l];
        ssum += v;
        asum -= v;
    }
    if (j != i) {
        const double v = K[(long)j * n + k];
        ssum += v;
        asum -= v;
        if (l != k) {
            const double v2 = K[(long)j * n + l];
            ssum += v2;
            asum += v2;
        }
    }
    const double isq2 = 0.70710678118654752440;
    const double wa = (i == j) ? 1.0 : isq2, wb = (k == l) ? 1.0 : isq2;
    Ksym[e] = wa * wb * ssum;
    if (a < na && b < na) Kanti[(long)a * na + b] = 0.5 * asum;
}

// The same fold for a positive semi-definite K (a covariance matrix), with everything the tridiagonalisation's preparation
// launches do fused in: replica r folds K + r * sK (sK = 0: one source), adds shift[r] to both blocks' diagonals (F (K + s I)
// F^T = F K F^T + s I: the jitter of the log-likelihood's Ks needs no copy of Ks), divides each block by a power of two >=
// its largest entry and writes it where the tridiagonalisation reads it (EigArenaView), reflector storage zeroed, *amax set.
// The largest entry of a PSD block is on its diagonal, so the scale comes from ns (na) diagonal entries that every workgroup
// evaluates for itself -- no absmax pass, no second launch.  (A block that is rounding noise next to the other one -- its
// diagonal may then be exceeded by its off-diagonal noise -- is scaled by at least 2^-30 of the larger block's scale.)
// Replaces copy -> add_diag -> fold -> absmax -> scale/copy/zero.  Non-finite entries are zeroed and reported (status 4).
struct PsdFoldFillArgs {
    const double *K;
    long sK;
    int n;
    double shift[2];
    const HpDev *tab;            // non-null: replica r adds tab[r].jitter (any number of replicas: the batched evaluations of fit)
    SymDev sy;
    double *A0s, *A0a, *Vs, *Va, *taus, *taua, *amaxs, *amaxa;
    long blks, blka;
    int *status;
    int status_stride;
    int elem_blocks;
};
__device__ __forceinline__ double pow2_at_least(double b) {
    if (!(b > 0.0 && b <= 1.7e308)) return 1.0;
    int ex = 0;
    (void)frexp(b, &ex);                       // b = f 2^ex, f in [0.5, 1)
    const double m = ldexp(1.0, ex);
    return (m <= 1.7e308) ? m : b;
}
__global__ __launch_bounds__(256) void psd_fold_fill_kernel(PsdFoldFillArgs g) {
    const int rep = blockIdx.y;
    const SymDev sy = g.sy;
    const int ns = sy.ns, na = sy.na, n = g.n;
    const double *__restrict__ K = g.K + rep * g.sK;
    const double shift = g.tab ? g.tab[rep].jitter : g.shift[rep];
    __shared__ double red[4];
    // scales: largest diagonal entry of each block (every workgroup for itself: <= 4 orbits per thread, L2 hits)
    const double isq2 = 0.70710678118654752440;
    double ds = 0.0, da = 0.0;
    for (int a = threadIdx.x; a < ns; a += 256) {
        const int i = sy.rep_i[a], j = sy.rep_j[a];
        const double kii = K[(long)i * n + i];
        if (i == j) ds = fmax(ds, fabs(kii + shift));
        else {
            const double kij = K[(long)i * n + j], kji = K[(long)j * n + i], kjj = K[(long)j * n + j];
            ds = fmax(ds, fabs(0.5 * (((kii + kij) + kji) + kjj) + shift));
            da = fmax(da, fabs(0.5 * (((kii - kij) - kji) + kjj) + shift));
        }
    }
    ds = block_max256(ds, red);
    __syncthreads();
    da = block_max256(da, red);
    const double big = fmax(ds, da);
    const double m_s = pow2_at_least(fmax(ds, big * 9.3132257461547852e-10));
    const double m_a = pow2_at_least(fmax(da, big * 9.3132257461547852e-10));
    if ((int)blockIdx.x >= g.elem_blocks) {
        const long i0 = ((long)blockIdx.x - g.elem_blocks) * 256 + threadIdx.x, stride = ((long)gridDim.x - g.elem_blocks) * 256;
        double *Vs = g.Vs + rep * g.blks, *Va = g.Va + rep * g.blka, *ts = g.taus + rep * g.blks, *ta = g.taua + rep * g.blka;
        const long nvs = (long)(ns + 64) * ns, nva = (long)(na + 64) * na;
        for (long i = i0; i < nvs; i += stride) Vs[i] = 0.0;
        for (long i = i0; i < nva; i += stride) Va[i] = 0.0;
        for (long i = i0; i < ns + 64; i += stride) ts[i] = 0.0;
        for (long i = i0; i < na + 64; i += stride) ta[i] = 0.0;
        if (blockIdx.x == g.elem_blocks && threadIdx.x == 0) {
            g.amaxs[rep * g.blks] = m_s;
            g.amaxa[rep * g.blka] = m_a;
        }
        return;
    }
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= (long)ns * ns) return;
    const int a = (int)(e / ns), b = (int)(e % ns);
    const int i = sy.rep_i[a], j = sy.rep_j[a], k = sy.rep_i[b], l = sy.rep_j[b];
    const double kik = K[(long)i * n + k];
    double ssum = kik, asum = kik;
    if (l != k) {
        const double v = K[(long)i * n + l];
        ssum += v;
        asum -= v;
    }
    if (j != i) {
        const double v = K[(long)j * n + k];
        ssum += v;
        asum -= v;
        if (l != k) {
            const double v2 = K[(long)j * n + l];
            ssum += v2;
            asum += v2;
        }
    }
    const double wa = (i == j) ? 1.0 : isq2, wb = (k == l) ? 1.0 : isq2;
    bool bad = false;
    auto scaled = [&](double x, double m) {
        x = (m > 1e300) ? x / m : x * (1.0 / m);
        if (!(fabs(x) <= 2.0)) {
            x = 0.0;
            bad = true;
        }
        return x;
    };
    const double dshift = (a == b) ? shift : 0.0;
    g.A0s[rep * g.blks + e] = scaled(wa * wb * ssum + dshift, m_s);
    if (a < na && b < na) g.A0a[rep * g.blka + (long)a * na + b] = scaled(0.5 * asum + dshift, m_a);
    if (bad && g.status) atomicMax(g.status + (long)rep * g.status_stride, 4);
}

void k_psd_fold_fill(gpcsd_ctx *c, const double *K, int n, long sK, int nrep, const double *shift, const SymDev &sy,
                     const EigArenaView &as, const EigArenaView &aa, int *status, int status_stride, hipStream_t s,
                     const HpDev *tab, bool tab_shifts_nonneg) {
    GP_REQUIRE(nrep >= 1 && (tab || nrep <= 2), -3, "psd fold fill: %d replicas (1 or 2 without a hyper-parameter table)", nrep);
    GP_REQUIRE(sy.ns > 0 && sy.ns + sy.na == n, -3, "psd fold fill: the symmetry does not cover the %d points", n);
    PsdFoldFillArgs g{};
    g.K = K; g.sK = sK; g.n = n;
    for (int r = 0; r < nrep && r < 2; ++r) g.shift[r] = shift ? shift[r] : 0.0;
    g.tab = tab;
    g.sy = sy;
    g.A0s = as.A0; g.A0a = aa.A0; g.Vs = as.V; g.Va = aa.V; g.taus = as.tau; g.taua = aa.tau; g.amaxs = as.amax; g.amaxa = aa.amax;
    g.blks = as.blk; g.blka = aa.blk;
    g.status = status;
    g.status_stride = status_stride;
    g.elem_blocks = ceil_div((long)sy.ns * sy.ns, 256);
    // K is positive semi-definite by this routine's contract and every shift is checked (host-side ones here, a table's by the
    // callers' hyper-parameter validation): the two arenas are announced as PSD -- only then may the tail stop early
    bool nonneg = tab ? tab_shifts_nonneg : true;
    for (int r = 0; r < nrep && r < 2 && !tab; ++r) nonneg = nonneg && g.shift[r] >= 0.0;
    *as.psd = *aa.psd = nonneg;
    ProfScope ps(c, "psd_fold_fill", 0.0, s);
    hipLaunchKernelGGL(psd_fold_fill_kernel, dim3(g.elem_blocks + 48, nrep), dim3(256), 0, s, g);
    GP_HIP(hipGetLastError());
}

// merge the two ascending spectra and expand the half-size eigenvectors: Z[:, rank] from (Us | Ua)
__global__ __launch_bounds__(256) void sym_unfold_kernel(int n, SymDev sy, const double *__restrict__ ws,
                                                         const double *__restrict__ Us, const double *__restrict__ wa,
                                                         const double *__restrict__ Ua, double *__restrict__ w,
                                                         double *__restrict__ Z) {
    const int ns = sy.ns, na = sy.na;
    {                                           // blockIdx.y = replica: half spectra n apart, half eigenvectors ns^2+na^2 apart
        const long r = blockIdx.y, su = (long)ns * ns + (long)na * na;
        ws += r * n; wa += r * n; Us += r * su; Ua += r * su;
        w += r * n; Z += r * (long)n * n;
    }
    const int col = blockIdx.x;                 // 0..ns-1: symmetric eigenvectors, ns..n-1: antisymmetric
    const bool is_sym = col < ns;
    const int cc = is_sym ? col : col - ns;
    const double val = is_sym ? ws[cc] : wa[cc];
    // stable merge rank: symmetric entries go first on ties
    int lo = 0, hi = is_sym ? na : ns;
    const double *other = is_sym ? wa : ws;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const bool before = is_sym ? (other[mid] < val) : (other[mid] <= val);
        if (before) lo = mid + 1; else hi = mid;
    }
    const int rank = cc + lo;
    if (threadIdx.x == 0) w[rank] = val;
    const double isq2 = 0.70710678118654752440;
    for (int row = threadIdx.x; row < n; row += 256) {
        const int a = sy.orb[row];
        const int sg = sy.sgn[row];             // 0 fixed point, +1 first of a pair, -1 second of a pair
        double v;
        if (is_sym) v = Us[(long)a * ns + cc] * (sg == 0 ? 1.0 : isq2);
        else v = (sg == 0) ? 0.0 : Ua[(long)a * na + cc] * isq2 * (double)sg;
        Z[(long)row * n + rank] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------------------------
static bool fold_applies(const SymDev *sy, int n) {
    return sy && sy->ns > 0 && sy->ns + sy->na == n && n > JACOBI_LDS_MAX && symfold_enabled() && !force_jacobi();
}

// a problem that can be solved in stages (see eigh_pair_device): folded, both halves on the tridiagonalisation path and
// within the fused back-transformation's capacity
bool eigh_stageable(const SymDev *sy, int n) {
    if (!fold_applies(sy, n)) return false;
    const int lo = std::min(sy->ns, sy->na), hi = std::max(sy->ns, sy->na);
    return lo > JACOBI_LDS_MAX && wy_fused_supported(hi);
}

// The half-size spectra and eigenvectors of problem `slot` (0: first / spatial, 1: second / temporal) of eigh_pair_device,
// in FOLD order: w = (ws | wa), U = (Us (ns x ns) | Ua (na x na)), eigenvectors in columns.  The same predicate and buffers
// as the solver itself uses, so callers that stay in the folded basis (capi.hip) read what the last solve left there.
FoldView eigh_fold_view(gpcsd_ctx *c, int slot, const SymDev *sy, int n, int count) {
    FoldView v;
    if (!fold_applies(sy, n)) return v;
    if (count < 1) count = 1;
    const std::string T = std::string("fold_p") + (slot ? "1" : "0") + "_";
    v.on = true;
    v.ns = sy->ns;
    v.na = sy->na;
    v.sw = n;
    v.sU = (long)v.ns * v.ns + (long)v.na * v.na;
    // outputs: one set per generation of the slot (gpcsd_ctx::par) -- the solver writes, and callers read, the current one
    const char *gen = c->par[slot ? 1 : 0] ? "#1" : "#0";
    v.w = c->buf<double>(T + "w" + gen, (size_t)n * count);
    v.U = c->buf<double>(T + "U" + gen, (size_t)v.sU * count);
    return v;
}

// [slot][generation]: the temporal classes exist twice (gpcsd_ctx::tgen)
static const char *const FOLD_TAGS[2][2][3] = {{{"p0", "p0s", "p0a"}, {"p0", "p0s", "p0a"}}, {{"p1", "p1s", "p1a"}, {"p1g", "p1gs", "p1ga"}}};
const char *const *eigh_fold_tags(const gpcsd_ctx *c, int slot) { return &FOLD_TAGS[slot ? 1 : 0][slot ? (c->tgen & 1) : 0][1]; }

static void eigh_pair_enqueue(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, const SymDev *sym0, double *A1, int n1,
                              double *w1, double *Z1, const SymDev *sym1, int *d_status, hipStream_t s, bool need_merged,
                              int count, int status_stride, int count1, int prefolded_mask, int stage) {
    const int cnt[2] = {count, count1 > 0 ? count1 : count};      // replicas of problem 0 / problem 1
    double *A[2] = {A0, A1}, *w[2] = {w0, w1}, *Z[2] = {Z0, Z1};
    const int n[2] = {n0, n1};
    const SymDev *sym[2] = {sym0, sym1};
    EigReq large[MAX_EIG_BATCH];
    int nlarge = 0;
    struct Fold {
        bool on = false;
        double *ws, *Us, *wa, *Ua;
    } fold[2];
    const char *const *tags[2] = {FOLD_TAGS[0][0], FOLD_TAGS[1][c->tgen & 1]};     // [problem] -> {whole, symmetric, antisymmetric}
    // A small problem next to a large one rides along in the large problem's launches for free (GPCSD1D: 24 electrodes
    // next to 500 time points), instead of a serial 250 us single-workgroup Jacobi in front of them.
    const bool pair_has_large = !force_jacobi() && (n0 > JACOBI_LDS_MAX || n1 > JACOBI_LDS_MAX);
    auto submit = [&](double *Am, int nm, double *wm, double *Zm, const char *tag, long sA, long sw, long sZ, int nrep,
                      bool prefilled = false) {
        if (nm <= 0) return;
        EigReq q;
        q.A = Am; q.n = nm; q.w = wm; q.Z = Zm; q.tag = tag;
        q.count = nrep; q.sA = sA; q.sw = sw; q.sZ = sZ;
        q.prefilled = prefilled;
        const bool small = nm <= JACOBI_LDS_MAX && !(pair_has_large && nm >= EIG_BATCH_MIN_N);
        GP_REQUIRE(!prefilled || !(small || force_jacobi()), -3, "eigh: a prefilled class must take the tridiagonalisation path (n=%d)", nm);
        if (small || force_jacobi()) eigh_jacobi(c, q, d_status, status_stride, s);
        else large[nlarge++] = q;
    };
    for (int p = 0; p < 2; ++p) {
        if (n[p] <= 0) continue;
        const SymDev *sy = sym[p];
        const long nn = (long)n[p] * n[p];
        if (fold_applies(sy, n[p])) {
            const int ns = sy->ns, na = sy->na;
            const std::string T = std::string("fold_") + tags[p][0] + "_";
            double *Ks = c->buf<double>(T + "Ks", (size_t)ns * ns * cnt[p]),
                   *Ka = c->buf<double>(T + "Ka", (size_t)std::max(na, 1) * na * cnt[p]);
            const FoldView fv = eigh_fold_view(c, p, sy, n[p], cnt[p]);
            fold[p].on = true;
            fold[p].ws = fv.w;
            fold[p].Us = fv.U;
            fold[p].wa = fv.w + ns;
            fold[p].Ua = fv.U + (size_t)ns * ns;
            const bool pre = ((prefolded_mask >> p) & 1) || stage >= 2;     // the scaled halves are in the class arenas already
            if (!pre)
                hipLaunchKernelGGL(sym_fold_kernel, dim3(ceil_div((long)ns * ns, 256), cnt[p]), dim3(256), 0, s, (const double *)A[p],
                                   n[p], *sy, Ks, Ka);
            submit(Ks, ns, fold[p].ws, fold[p].Us, tags[p][1], (long)ns * ns, fv.sw, fv.sU, cnt[p], pre);
            submit(Ka, na, fold[p].wa, fold[p].Ua, tags[p][2], (long)na * na, fv.sw, fv.sU, cnt[p], pre);
        } else {
            GP_REQUIRE(stage == 0, -3, "eigh: staged solves need symmetry-folded problems (problem %d)", p);
            GP_REQUIRE(!((prefolded_mask >> p) & 1), -3, "eigh: problem %d was announced as prefolded but symmetry folding does not apply", p);
            submit(A[p], n[p], w[p], Z[p], tags[p][0], nn, n[p], nn, cnt[p]);
        }
    }
    if (nlarge) eigh_large_multi(c, large, nlarge, d_status, status_stride, s, stage);
    // need_merged == false: the caller stays in the folded basis (eigh_fold_view) and never reads w / Z of a folded problem
    for (int p = 0; p < 2; ++p)
        if (fold[p].on && need_merged && (stage == 0 || stage == 4))
            hipLaunchKernelGGL(sym_unfold_kernel, dim3(n[p], cnt[p]), dim3(256), 0, s, n[p], *sym[p], (const double *)fold[p].ws,
                               (const double *)fold[p].Us, (const double *)fold[p].wa, (const double *)fold[p].Ua, w[p], Z[p]);
    GP_HIP(hipGetLastError());
}

// The large-n path is hundreds of dependent launches with nothing decided on the host (deflation counts stay on the
// device), so it replays as a hipGraph: first call eager (allocates workspaces), second call captured, later calls
// replayed.  A graph is retired whenever any context buffer is (re)allocated, since it holds raw device pointers.
void eigh_pair_device(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, const SymDev *sym0, double *A1, int n1,
                      double *w1, double *Z1, const SymDev *sym1, int *d_status, hipStream_t s, bool need_merged, int count,
                      int status_stride, int count1, int prefolded_mask, int stage) {
    if (count < 1) count = 1;
    if (count1 < 1) count1 = count;
    if (stage < 2) {                           // (stages 2, 3 and 4 continue the solve stage 1 started)
        if (n0 > 0) ++c->eig_gen[0];           // whatever a previous call left in this slot's outputs is about to be replaced
        if (n1 > 0) ++c->eig_gen[1];
    }
    // the limit applies to what the solver actually factorises: a symmetry-folded problem is two half-size ones
    const int m0 = fold_applies(sym0, n0) ? std::max(sym0->ns, sym0->na) : n0;
    const int m1 = fold_applies(sym1, n1) ? std::max(sym1->ns, sym1->na) : n1;
    // (GPCSD_EIGH=jacobi, the single-workgroup cross-check, keeps its own smaller limit)
    const int cap = force_jacobi() ? JACOBI_MAX_N : EIG_MAXN;
    GP_REQUIRE(m0 <= cap && m1 <= cap, GPCSD_ERR_CAPACITY,
               "eigh: matrix order %d / %d (after symmetry folding: %d / %d) exceeds the eigensolver's capacity of %d rows "
               "(GPCSD_MAX_EIG_N)", n0, n1, m0, m1, cap);
    const bool any_large = !force_jacobi() && (n0 > JACOBI_LDS_MAX || n1 > JACOBI_LDS_MAX);
    // profiling mode 3 keeps replaying graphs, so the outer scopes time the chains as they run in production
    const bool prof_graph = c->prof_mode == 3;
    // (stage 5 records events and queues products on the main stream between its launches: never captured)
    if (!any_large || (c->prof_on && !prof_graph) || !graphs_enabled() || stage == 5) {
        eigh_pair_enqueue(c, A0, n0, w0, Z0, sym0, A1, n1, w1, Z1, sym1, d_status, s, need_merged, count, status_stride, count1, prefolded_mask, stage);
        return;
    }
    // (the generation of each slot picks the fold-order output buffers, which are not among the arguments)
    // (... and SytrdProb::psd of the prefilled classes, a kernel argument: what their fills announced, and the context's switch)
    int psd_sig = (c->tail_early_exit ? 16 : 0) | (c->claim_psd ? 32 : 0) |
                  (c->pipe_req ? 128 : 0);                                                           // (progress words: a kernel argument)
    for (int p = 0; p < 2; ++p) {
        const char *const *tg = eigh_fold_tags(c, p);
        for (int h = 0; h < 2; ++h) {
            const auto it = c->arena_psd.find(tg[h]);
            if (((prefolded_mask >> p) & 1) && it != c->arena_psd.end() && it->second) psd_sig |= 1 << (2 * p + h);
        }
    }
    char key[448];
    const gpcsd_ctx::QPipeX qx = stage == 5 ? c->q_pipe_x : gpcsd_ctx::QPipeX();       // (stage 5 launches the caller's products)
    int nk = snprintf(key, sizeof(key), "x%p|%p|%d|%d|%d|%d|%d|", (const void *)qx.in, (void *)qx.out, qx.M, qx.ld, qx.c0[0], qx.c0[1], qx.rep);
    snprintf(key + nk, sizeof(key) - nk, "eigh|%p|%d|%p|%p|%p|%d|%p|%d|%p|%p|%p|%d|%p|%p|%d|%d|%d|%d%d|%d|%d|%d|%d|%d", (void *)A0, n0, (void *)w0, (void *)Z0,
             (void *)(sym0 ? sym0->rep_i : nullptr), sym0 ? sym0->ns : 0, (void *)A1, n1, (void *)w1, (void *)Z1,
             (void *)(sym1 ? sym1->rep_i : nullptr), sym1 ? sym1->ns : 0, (void *)d_status, (void *)s, (int)need_merged, count,
             status_stride, c->par[0], c->par[1], count1, prefolded_mask, (int)(c->prof_mode == 3),   // (mode 3 graphs carry clock stamps)
             stage + 8 * (c->tgen & 1), psd_sig);                                     // (the generation picks the temporal arenas)
    gpcsd_ctx::GraphSlot &g = c->graphs[key];
    if (g.exec && g.epoch == c->alloc_epoch) {
        GP_HIP(hipGraphLaunch(g.exec, s));
        return;
    }
    if (g.seen_epoch == c->alloc_epoch) {          // allocations are stable since the last eager run: capture now
        if (g.exec) {
            (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        GP_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        c->capturing = true;
        try {
            eigh_pair_enqueue(c, A0, n0, w0, Z0, sym0, A1, n1, w1, Z1, sym1, d_status, s, need_merged, count, status_stride, count1, prefolded_mask, stage);
        } catch (...) {
            c->capturing = false;
            (void)hipStreamEndCapture(s, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        c->capturing = false;
        GP_HIP(hipStreamEndCapture(s, &graph));
        if (g.seen_epoch == c->alloc_epoch) {
            GP_HIP(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
            g.epoch = c->alloc_epoch;
        }
        (void)hipGraphDestroy(graph);
        if (g.exec && g.epoch == c->alloc_epoch) {
            GP_HIP(hipGraphLaunch(g.exec, s));
            return;
        }
    }
    eigh_pair_enqueue(c, A0, n0, w0, Z0, sym0, A1, n1, w1, Z1, sym1, d_status, s, need_merged, count, status_stride, count1, prefolded_mask, stage);
    g.seen_epoch = c->alloc_epoch;
}

void eigh_device(gpcsd_ctx *c, double *A, int n, double *evals, double *evecs, int *d_status, hipStream_t s,
                 const char *tag) {
    GP_REQUIRE(n >= 1, -3, "eigh: n=%d must be positive", n);
    GP_REQUIRE(n <= (force_jacobi() ? JACOBI_MAX_N : EIG_MAXN), GPCSD_ERR_CAPACITY,
               "eigh: matrix order %d exceeds the eigensolver's capacity of %d rows (GPCSD_MAX_EIG_N)", n,
               force_jacobi() ? JACOBI_MAX_N : EIG_MAXN);
    (void)tag;
    eigh_pair_device(c, A, n, evals, evecs, nullptr, nullptr, 0, nullptr, nullptr, nullptr, d_status, s, true, 1, 0, -1, 0, 0);
}

}  // namespace gpcsd
