// Divide & conquer eigensolver for symmetric tridiagonal matrices on gfx950 (stage 2 of the large-n eigensolver; the
// work LAPACK does in dstedc behind numpy.linalg.eigh, src/gpcsd/utility_functions.py:58-59).
//
// Cuppen tearing with every tear applied up front; leaves (<= DC_LEAF rows) by Jacobi in LDS, one wave each; then
// bottom-up merges.  Per merge: z from the boundary rows of the two eigenvector blocks, stable merge order, dlaed2-style
// deflation scan, secular equation with origin shift (one wave per root, rational two-pole iteration safeguarded by
// bisection), Gu/Eisenstat z-hat through the Loewner formula so the eigenvector block is orthogonal to working
// precision, U = zhat_i / (d_i - lam_j) normalised, Z_new = Q2 U, rank sort of {roots} and {deflated poles}.
//
// Launch structure (latency is what matters at n <= 1024):
//   * several independent problems share every launch (blockIdx.z);
//   * levels whose merges have <= DC_SMALL rows run as ONE launch each (a workgroup walks all phases of its merge,
//     matmul included); larger levels run one launch per phase and the Q2 U product on the fp64 MFMA GEMM with the
//     non-deflated count read on the device (no host round trip anywhere).
// tools/dc_prototype.py is the NumPy model this was validated against before porting.
#include <algorithm>
#include <string>
#include <vector>

#include "devutil.hpp"
#include "jacobi.hpp"
#include "kernels.hpp"
#include "wy_prep.hpp"

namespace gpcsd {

constexpr int DC_LEAF = 8;
constexpr int DC_SMALL = 64;          // merges up to this size use the fused single-launch level kernel
constexpr int NT_SMALL = 512;

struct Seg {
    int lo, mid, hi;
};

struct DcWork {            // per problem, all device pointers
    int n;
    double *dcur, *dnext;  // eigenvalues of the current / next level (n)
    double *Qcur, *Qnext;  // block-diagonal eigenvector matrices (n x n)
    double *dwork;         // torn diagonal, later d after deflation (n)
    double *dk, *zk;       // compacted non-deflated poles / weights, stored at [lo, lo+K)
    double *mu, *lam, *zhat, *invn;
    int *org, *ndidx, *deflidx, *rota, *rotb, *meta;   // meta[2*m] = K, meta[2*m+1] = nrot
    double *rotc, *rots;
    double *Q2w, *Uw, *Ww; // n x n workspaces (diagonal blocks used)
    const double *d0, *e;  // the input tridiagonal
    int *Kdyn;             // K per merge of the current level (device ints for the dynamic-size GEMM)
    const int *tbl;        // leaf_lo | bounds | per-level Seg triples
    double *wout, *Zout;   // caller's outputs
    // replicas of the class (see EigReq): workspace pointers are those of replica 0 in an arena of `blk` doubles per
    // replica; the input tridiagonal lives in the tridiagonalisation's arena (s_in), the outputs are the caller's (sw, sZ)
    long blk, s_in, sw, sZ;
    int top;               // this level is the top merge: it places its result in wout / Zout instead of dnext / Qnext
};

struct DcLevel {           // kernel argument of one level: one DcWork per CLASS
    DcWork w[MAX_BATCH];
    int seg_off[MAX_BATCH];   // offset of this level's Seg triples inside tbl
    int nseg[MAX_BATCH];      // 0: problem absent at this level
    int aux[MAX_BATCH];       // leaves: number of leaves; tears: number of boundaries (offset in seg_off)
    int start[MAX_BATCH + 1]; // prefix sums of the replica counts (class_of)
    int status_stride;        // replica r reports failure in status[r * status_stride]
    int prio;                 // the level's waves raise their issue priority (GPCSD_DC_PRIO=0: A/B)
};

// workgroup's problem index g -> its class and replica, and the class's DcWork with every pointer moved to the replica
__device__ __forceinline__ DcWork dc_resolve(const DcLevel &L, int g, int &cls, int &rep) {
    class_of(L.start, g, cls, rep);
    DcWork w = L.w[cls];
    const long o = rep * w.blk, oi = 2 * o;           // int slices live in the same arena: 2 ints per double
    w.dcur += o; w.dnext += o; w.Qcur += o; w.Qnext += o; w.dwork += o; w.dk += o; w.zk += o;
    w.mu += o; w.lam += o; w.zhat += o; w.invn += o; w.rotc += o; w.rots += o; w.Q2w += o; w.Uw += o; w.Ww += o;
    w.org += oi; w.ndidx += oi; w.deflidx += oi; w.rota += oi; w.rotb += oi; w.meta += oi; w.Kdyn += oi;
    w.d0 += rep * w.s_in; w.e += rep * w.s_in;
    w.wout += rep * w.sw; w.Zout += rep * w.sZ;
    if (w.top) {
        w.dnext = w.wout;
        w.Qnext = w.Zout;
    }
    return w;
}

__device__ __forceinline__ Seg load_seg(const DcWork &w, int off, int m) {
    const int *p = w.tbl + off + 3 * m;
    return Seg{p[0], p[1], p[2]};
}

// ------------------------------------------------------------------------------------------------------------------
// tears + leaves
// ------------------------------------------------------------------------------------------------------------------
// eigenvalues / eigenvectors of the top level into the caller's arrays, all problems in one launch
__global__ __launch_bounds__(256) void dc_output_kernel(DcLevel L) {
    int cls, rep;
    const DcWork w = dc_resolve(L, blockIdx.y, cls, rep);
    const long n = w.n, nn = n * n;
    const long stride = (long)gridDim.x * 256, i0 = blockIdx.x * 256L + threadIdx.x;
    for (long i = i0; i < nn; i += stride) w.Zout[i] = w.Qcur[i];
    for (long i = i0; i < n; i += stride) w.wout[i] = w.dcur[i];
}

// Register-resident Jacobi for a leaf (m <= 8): lane (i, j) = (lane >> 3, lane & 7) holds A[i][j] and V[i][j]; rotation
// parameters and partner rows / columns move with lane shuffles, nothing goes through LDS arrays and there is no barrier.
// Same cyclic round-robin schedule and the same (p < q) rotation formulas as jacobi_body.
__device__ __forceinline__ int rr_partner8(int x, int step7) {        // partner of index x at round-robin step (8 players)
    if (x == 7) return step7;
    if (x == step7) return 7;
    int y = 2 * step7 - x;
    y += (y < 0) ? 7 : 0;
    y -= (y >= 7) ? 7 : 0;
    return y;
}

// Workgroups [0, nleaf_max) solve leaves; workgroup nleaf_max + r zeroes row r of both ping-pong eigenvector matrices
// outside the row's own leaf block (a merge reads the off-diagonal blocks between its halves, which no level writes).
// The Cuppen tears are applied here as well: every leaf boundary is the boundary of some merge, so the first / last
// diagonal entry of a leaf loses |e| of the coupling it was cut from.  One launch instead of tear + leaf.
// One wave = one unit of work: bx < nleaf_max is a leaf, bx - nleaf_max a row to zero-fill; no barriers, no LDS, so the
// body runs as a 64-thread workgroup (dc_leaf_reg_kernel) or as one wave of a larger one (dc_leaf_wyprep_kernel).
__device__ __forceinline__ void dc_leaf_reg_body(const DcLevel &L, int *status, const int nleaf_max, const int bx, const int by,
                                                 const int lane) {
    int cls, rep;
    const DcWork w = dc_resolve(L, by, cls, rep);
    status += (long)rep * L.status_stride;
    const int n = w.n, nleaf = L.aux[cls];
    if (bx >= nleaf_max) {
        const int r = bx - nleaf_max;
        if (r >= n) return;
        int lf = 0;
        while (lf + 1 < nleaf && w.tbl[lf + 1] <= r) ++lf;          // wave-uniform scan of <= 128 leaf bounds
        const int blo = w.tbl[lf], bhi = w.tbl[lf + 1];
        for (int cidx = lane; cidx < n; cidx += 64) {
            if (cidx < blo || cidx >= bhi) w.Qcur[(long)r * n + cidx] = 0.0;
            w.Qnext[(long)r * n + cidx] = 0.0;
        }
        return;
    }
    if (bx >= nleaf) return;
    const int lo = w.tbl[bx], hi = w.tbl[bx + 1];
    const int m = hi - lo;
    const int i = lane >> 3, j = lane & 7;
    // A non-finite d / e (they should not occur: the drivers sanitise their input) must not reach the merges, whose rank
    // sorts would index out of range: read as zero, repaired in place for the later launches, reported as a failure.
    bool bad = false;
    auto san = [&](double x) {
        if (fabs(x) <= 1.7e308) return x;
        bad = true;
        return 0.0;
    };
    double a = 0.0;
    if (i < m && j < m) {
        if (i == j) {
            a = san(w.d0[lo + i]);
            if (i == 0 && lo > 0) a -= fabs(san(w.e[lo - 1]));
            if (i == m - 1 && hi < n) {
                const double eb = w.e[hi - 1];
                if (!(fabs(eb) <= 1.7e308)) const_cast<double *>(w.e)[hi - 1] = 0.0;   // the coupling this leaf owns
                a -= fabs(san(eb));
            }
        } else if (j == i + 1) {
            const double ee = w.e[lo + i];
            if (!(fabs(ee) <= 1.7e308)) const_cast<double *>(w.e)[lo + i] = 0.0;
            a = san(ee);
        } else if (i == j + 1) a = san(w.e[lo + j]);
    }
    if (__ballot(bad) != 0ull && lane == 0) atomicMax(status, 4);
    double v = (i == j) ? 1.0 : 0.0;
    const double thresh = sqrt(wave_sum(a * a)) * EPS_U / (double)m;
    bool converged = (m <= 1);
    int sweep = 0;
    for (; sweep < JACOBI_MAX_SWEEPS && !converged; ++sweep) {
        bool rotated = false;
        for (int step = 0; step < 7; ++step) {
            // rotation of the column pair {j, pc}: parameters from A[p][p], A[q][q], A[p][q] (p < q)
            const int pc = rr_partner8(j, step), p = min(j, pc), q = max(j, pc);
            const double app = __shfl(a, 9 * p, 64), aqq = __shfl(a, 9 * q, 64), apq = __shfl(a, 8 * p + q, 64);
            const bool act = (q < m) && (fabs(apq) > thresh);
            double c = 1.0, sn = 0.0;
            if (act) {
                const double theta = (aqq - app) * jac_rcp(2.0 * apq);
                double t;
                if (fabs(theta) > 1e150) t = 0.5 * jac_rcp(theta);
                else {
                    const double th2 = theta * theta + 1.0;
                    t = copysign(1.0, theta) * jac_rcp(fabs(theta) + th2 * jac_rsqrt(th2));
                }
                c = jac_rsqrt(t * t + 1.0);
                sn = t * c;
            }
            rotated |= act;
            // column phase: A <- A J, V <- V J
            {
                const double ao = __shfl(a, 8 * i + pc, 64), vo = __shfl(v, 8 * i + pc, 64);
                a = (j == p) ? c * a - sn * ao : sn * ao + c * a;
                v = (j == p) ? c * v - sn * vo : sn * vo + c * v;
            }
            // row phase: A <- J^T A with the parameters of the row pair {i, pr} (held by lane (pr.., i) as a column pair)
            {
                const int pr = rr_partner8(i, step), rp = min(i, pr), rq = max(i, pr);
                const double cr = __shfl(c, i, 64), sr = __shfl(sn, i, 64);       // lane (0, i): column pair of index i
                const double ao = __shfl(a, 8 * pr + j, 64);
                double an = (i == rp) ? cr * a - sr * ao : sr * ao + cr * a;
                const bool ract = (sr != 0.0) || (cr != 1.0);
                if (ract && ((i == rp && j == rq) || (i == rq && j == rp))) an = 0.0;   // the annihilated element, exactly
                a = an;
            }
        }
        converged = (__ballot(rotated) == 0ull);
    }
    // ascending order by rank counting (ties by index), scatter eigenpairs
    const double dj = __shfl(a, 9 * j, 64);                   // eigenvalue of column j
    int rank = 0;
    for (int k = 0; k < m; ++k) {
        const double dk = __shfl(a, 9 * k, 64);
        rank += (dk < dj) || (!(dj < dk) && k < j);         // a valid permutation even for unordered (NaN) values
    }
    if (i < m && j < m) {
        if (i == 0) w.dcur[lo + rank] = dj;
        w.Qcur[(long)(lo + i) * n + lo + rank] = v;
    }
    if (lane == 0 && !converged) atomicMax(status, 1);
}

__global__ __launch_bounds__(64) void dc_leaf_reg_kernel(DcLevel L, int *status, int nleaf_max) {
    dc_leaf_reg_body(L, status, nleaf_max, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x);
}

// The leaf launch with a second role: workgroups x < nprep form the T factors of the compact-WY panels of the
// back-transformation (wy_prep.hpp) -- they depend on the reflectors only, i.e. on the same predecessor as the leaves, and
// would otherwise be a 35 us launch of their own later in the chain; the others run sixteen leaf / zero-fill units each.
__global__ __launch_bounds__(1024) void dc_leaf_wyprep_kernel(DcLevel L, int *status, int nleaf_max, int nunits, WyBatch wb,
                                                              int nprep) {
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    if ((int)blockIdx.x < nprep) {
        wy_prep_role(wy_resolve(wb, blockIdx.y), (int)blockIdx.x, (int)threadIdx.x);
        return;
    }
    const int bx = ((int)blockIdx.x - nprep) * 16 + ((int)threadIdx.x >> 6);
    if (bx >= nunits) return;
    dc_leaf_reg_body(L, status, nleaf_max, bx, (int)blockIdx.y, (int)threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------------------------------
// merge phases as device functions (called by the per-phase kernels and by the fused small-level kernel)
// ------------------------------------------------------------------------------------------------------------------
// 37 bytes of LDS per row of the merge (57 until round 3: a merge of 4096 rows now fits the 160 KB of a CU): the survivors of
// the negligible-weight test are compacted IN PLACE (cd = ds, cz = zs: the compaction only moves entries towards the front,
// one wave, chunk after chunk, every lane's reads in front of the chunk's writes), and their local indices take the storage of
// sz, which is dead once the merged order has been formed.
template <int CAP>
struct SetupSharedT {
    double sd[CAP], sz[CAP];             // by local index (sz: until the merge ranks; then cidx())
    double ds[CAP], zs[CAP];             // in merged ascending order; from the compaction on: the survivors, merged order
    int sperm[CAP];
    unsigned char sdefl[CAP];
    double red[16];
    int K, nrot, nsurv, close_pairs;
    __device__ __forceinline__ int *cidx() { return reinterpret_cast<int *>(sz); }
};
using SetupShared = SetupSharedT<EIG_MAXN>;
// the fused small-level kernel merges at most SM rows (SM = 32 or DC_SMALL = 64) and reuses sd | sz | ds | zs as its SM x SM U tile
template <int SM>
using SetupSharedSmallT = SetupSharedT<SM * SM / 4>;

// z, merged order, deflation scan, compacted poles.  Whole workgroup (NW waves).
template <int NW, class SS>
__device__ void dc_setup_body(const DcWork &w, const Seg sg, const int m, SS &S) {
    constexpr int NT = 64 * NW;
    const int lo = sg.lo, mid = sg.mid, hi = sg.hi, n = w.n;
    const int N = hi - lo, n1 = mid - lo;
    const int tid = threadIdx.x;
    const double beta = w.e[mid - 1];
    const double rho = 2.0 * fabs(beta);
    const double sgn = beta >= 0.0 ? 1.0 : -1.0;
    const double isq2 = 0.70710678118654752440;
    double dmax = 0.0, zmax = 0.0;
    for (int i = tid; i < N; i += NT) {
        double dv = w.dcur[lo + i];
        dv = (dv == dv) ? dv : 1.7e308;          // an unordered value would break the merge ranks (indices out of range)
        const double zv = (i < n1 ? w.Qcur[(long)(mid - 1) * n + lo + i] : sgn * w.Qcur[(long)mid * n + lo + i]) * isq2;
        S.sd[i] = dv;
        S.sz[i] = zv;
        S.sdefl[i] = 0;
        S.sperm[i] = i;                          // every slot holds a valid index even if the halves were not ascending
        dmax = fmax(dmax, fabs(dv));
        zmax = fmax(zmax, fabs(zv));
    }
    dmax = block_max<NW>(dmax, S.red);
    zmax = block_max<NW>(zmax, S.red);
    const double tol = 8.0 * EPS_U * fmax(dmax, zmax);
    // stable merge ranks of the two ascending halves
    for (int i = tid; i < N; i += NT) {
        const double v = S.sd[i];
        int cnt;
        if (i < n1) {                      // # of second-half entries strictly below v
            int a = n1, bnd = N;
            while (a < bnd) {
                const int mdl = (a + bnd) >> 1;
                if (S.sd[mdl] < v) a = mdl + 1; else bnd = mdl;
            }
            cnt = i + (a - n1);
        } else {                           // # of first-half entries <= v
            int a = 0, bnd = n1;
            while (a < bnd) {
                const int mdl = (a + bnd) >> 1;
                if (S.sd[mdl] <= v) a = mdl + 1; else bnd = mdl;
            }
            cnt = (i - n1) + a;
        }
        S.sperm[cnt] = i;
        S.ds[cnt] = v;
        S.zs[cnt] = S.sz[i];
    }
    __syncthreads();
    const bool all_defl = (rho * zmax <= tol);
    // type-(a) deflation (negligible weight) is elementwise; survivors are compacted in merged order so the
    // sequential part below only walks poles that can still pair up
    for (int jj = tid; jj < N; jj += NT)
        if (all_defl || rho * fabs(S.zs[jj]) <= tol) S.sdefl[S.sperm[jj]] = 1;
    __syncthreads();
    if (tid < 64) {                      // wave 0: stable compaction of the survivors by ballot prefix, in place
        int *const cidx = S.cidx();
        int base = 0;
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int jj = j0 + tid;
            const int sp = (jj < N) ? S.sperm[jj] : 0;
            const double dv = (jj < N) ? S.ds[jj] : 0.0, zv = (jj < N) ? S.zs[jj] : 0.0;   // the chunk is read ...
            const bool keep = (jj < N) && !S.sdefl[sp];
            const unsigned long long mask = __ballot(keep);
            if (keep) {                                                                  // ... before any of it is overwritten
                const int pos = base + __popcll(mask & ((1ull << tid) - 1ull));          // pos <= jj
                cidx[pos] = sp;
                S.ds[pos] = dv;
                S.zs[pos] = zv;
            }
            base += __popcll(mask);
        }
        if (tid == 0) S.nsurv = base;
    }
    __syncthreads();
    // Close-pair (type b) deflation chains are rare for well separated spectra (none in the Matern/SE temporal Gram
    // matrices): test every neighbouring pair of survivors with its ORIGINAL values in parallel.  Until the first pair
    // passes, that is exactly what the sequential scan evaluates, so "no pair passes" means K = nsurv and no rotations,
    // and the survivors are copied out in parallel.  Otherwise the sequential scan below decides.
    {
        const int ns = S.nsurv;
        if (tid == 0) S.close_pairs = 0;
        __syncthreads();
        bool hit = false;
        for (int p = 1 + tid; p < ns; p += NT) {
            const double t = S.ds[p] - S.ds[p - 1], zp = S.zs[p - 1], zn = S.zs[p];
            hit |= fabs(t) * fabs(zp * zn) <= tol * (zp * zp + zn * zn) * (1.0 + 1e-10);
        }
        if (hit) S.close_pairs = 1;                 // benign race: every writer stores 1
        __syncthreads();
        if (!S.close_pairs) {
            for (int p = tid; p < ns; p += NT) {
                w.ndidx[lo + p] = S.cidx()[p];
                w.dk[lo + p] = S.ds[p];
                w.zk[lo + p] = S.zs[p];
            }
            if (tid == 0) {
                S.K = ns;
                S.nrot = 0;
                w.meta[2 * m] = ns;
                w.meta[2 * m + 1] = 0;
                w.Kdyn[m] = ns;
            }
        }
    }
    if (tid == 0 && S.close_pairs) {
        // sequential dlaed2 close-pair scan over the survivors; the pending pole lives in registers and the next
        // survivor is loaded one iteration ahead so LDS latency overlaps the arithmetic
        const int ns = S.nsurv;
        int K = 0, nrot = 0;
        if (ns > 0) {
            const int *const cidx = S.cidx();
            int pj = cidx[0];
            double dpj = S.ds[0], zpj = S.zs[0];
            int nidx = ns > 1 ? cidx[1] : 0;
            double nd = ns > 1 ? S.ds[1] : 0.0, nz = ns > 1 ? S.zs[1] : 0.0;
            for (int p = 1; p < ns; ++p) {
                const int idx = nidx;
                double dn = nd, zn = nz;
                if (p + 1 < ns) {
                    nidx = cidx[p + 1];
                    nd = S.ds[p + 1];
                    nz = S.zs[p + 1];
                }
                const double t = dn - dpj;
                bool merged = false;
                // |t c s| <= tol  <=>  |t| |zp zn| <= tol (zp^2 + zn^2): sqrt- and division-free reject
                if (fabs(t) * fabs(zpj * zn) <= tol * (zpj * zpj + zn * zn) * (1.0 + 1e-10)) {
                    const double tau = hypot(zn, zpj);
                    const double c_ = zn / tau, s_ = -zpj / tau;
                    if (fabs(t * c_ * s_) <= tol) {
                        w.rota[lo + nrot] = pj;
                        w.rotb[lo + nrot] = idx;
                        w.rotc[lo + nrot] = c_;
                        w.rots[lo + nrot] = s_;
                        ++nrot;
                        S.sd[pj] = dpj * c_ * c_ + dn * s_ * s_;
                        S.sdefl[pj] = 1;
                        dn = dpj * s_ * s_ + dn * c_ * c_;
                        zn = tau;
                        merged = true;
                    }
                }
                if (!merged) {
                    w.ndidx[lo + K] = pj;
                    w.dk[lo + K] = dpj;
                    w.zk[lo + K] = zpj;
                    ++K;
                }
                pj = idx; dpj = dn; zpj = zn;
            }
            w.ndidx[lo + K] = pj;
            w.dk[lo + K] = dpj;
            w.zk[lo + K] = zpj;
            ++K;
        }
        S.K = K;
        S.nrot = nrot;
        w.meta[2 * m] = K;
        w.meta[2 * m + 1] = nrot;
        w.Kdyn[m] = K;
    }
    __syncthreads();
    if (tid < 64) {                      // deflated local indices in index order (final positions come from the rank sort)
        int base = 0;
        for (int i0 = 0; i0 < N; i0 += 64) {
            const int i = i0 + tid;
            const bool df = (i < N) && S.sdefl[i];
            const unsigned long long mask = __ballot(df);
            if (df) w.deflidx[lo + base + __popcll(mask & ((1ull << tid) - 1ull))] = i;
            base += __popcll(mask);
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += NT) w.dwork[lo + i] = S.sd[i];
}

// Apply the recorded Givens chain to row r of the merge block (in place).  One thread per row; the chained column
// stays in a register, the partner column of the next rotation is independent of the chain.
__device__ __forceinline__ void dc_rotate_row(const DcWork &w, const int lo, const int nrot, double *q) {
    if (nrot <= 0) return;
    int a = w.rota[lo];
    double carry = q[a];
    for (int t = 0; t < nrot; ++t) {
        const int bb = w.rotb[lo + t];
        const double c_ = w.rotc[lo + t], s_ = w.rots[lo + t];
        const double qb = q[bb];
        q[a] = c_ * carry + s_ * qb;
        carry = -s_ * carry + c_ * qb;
        if (t + 1 < nrot) {
            const int an = w.rota[lo + t + 1];
            if (an != bb) {
                q[bb] = carry;
                carry = q[an];
            }
            a = an;
        } else {
            q[bb] = carry;
        }
    }
}

// rows [r0, r1) of the merge block: rotations (one thread per row) then compaction of the non-deflated columns into
// Q2w[:, lo : lo+K), coalesced along the column index.  Whole workgroup.
template <int NT>
__device__ void dc_rotate_compact_body(const DcWork &w, const Seg sg, const int K, const int nrot, const int r0, const int r1) {
    const int lo = sg.lo, n = w.n;
    if (nrot > 0) {
        const int r = r0 + (int)threadIdx.x;
        if (r < r1) dc_rotate_row(w, lo, nrot, w.Qcur + (long)r * n + lo);
        __syncthreads();
    }
    const int rows = r1 - r0;
    for (int idx = threadIdx.x; idx < rows * K; idx += NT) {
        const int r = r0 + idx / K, t = idx % K;
        w.Q2w[(long)r * n + lo + t] = w.Qcur[(long)r * n + lo + w.ndidx[lo + t]];
    }
}

// One wave: root i of 1 + rho sum_j zk_j^2 / (dk_j - lam) = 0, i in [0, K).  Writes org / mu / lam.
__device__ void dc_secular_root(const DcWork &w, const int lo, const int K, const double rho, const int i) {
    const int lane = threadIdx.x & 63;
    const double *__restrict__ dk = w.dk + lo;
    const double *__restrict__ zk = w.zk + lo;
    if (K == 1) {
        if (lane == 0) {
            const double m = rho * zk[0] * zk[0];
            w.org[lo] = 0;
            w.mu[lo] = m;
            w.lam[lo] = dk[0] + m;
        }
        return;
    }
    const bool last = (i == K - 1);
    int org;
    double lo_b, hi_b;
    if (last) {
        org = K - 1;
        double s = 0.0;
        for (int j = lane; j < K; j += 64) s += zk[j] * zk[j];
        s = wave_sum(s);
        lo_b = 0.0;
        hi_b = rho * s;
    } else {
        const double di = dk[i];
        const double half = 0.5 * (dk[i + 1] - di);
        double s = 0.0;
        for (int j = lane; j < K; j += 64) s += zk[j] * zk[j] / ((dk[j] - di) - half);
        const double fmid = 1.0 + rho * wave_sum(s);
        if (fmid >= 0.0) {
            org = i;
            lo_b = 0.0;
            hi_b = half;
        } else {
            org = i + 1;
            lo_b = -half;
            hi_b = 0.0;
        }
    }
    const double dorg = dk[org];
    const double pl = dk[i] - dorg;
    const double pr = last ? 0.0 : dk[i + 1] - dorg;
    double mu = last ? 0.5 * hi_b : 0.5 * (lo_b + hi_b);
    for (int it = 0; it < 100; ++it) {
        double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0;
        for (int j = lane; j < K; j += 64) {
            const double rinv = 1.0 / ((dk[j] - dorg) - mu);
            const double term = zk[j] * zk[j] * rinv;
            if (j <= i) {
                psi += term;
                dpsi += term * rinv;
            } else {
                phi += term;
                dphi += term * rinv;
            }
        }
        psi = rho * wave_sum(psi);
        dpsi = rho * wave_sum(dpsi);
        phi = rho * wave_sum(phi);
        dphi = rho * wave_sum(dphi);
        const double f = 1.0 + psi + phi;
        const double err = 8.0 * EPS_U * (1.0 + fabs(psi) + fabs(phi)) + fabs(mu) * EPS_U * (dpsi + dphi);
        if (fabs(f) <= err) break;
        if (f < 0.0) lo_b = fmax(lo_b, mu);
        else hi_b = fmin(hi_b, mu);
        if (hi_b - lo_b <= 2.0 * EPS_U * fmax(fabs(lo_b), fabs(hi_b))) break;
        const double D1 = pl - mu;
        double eta = INFINITY;
        if (last) {
            const double g = 1.0 + psi - dpsi * D1;
            if (g > 0.0) eta = D1 + dpsi * D1 * D1 / g;
        } else {
            const double D2 = pr - mu;
            const double A = f - dpsi * D1 - dphi * D2;
            const double B = A * (D1 + D2) + dpsi * D1 * D1 + dphi * D2 * D2;
            const double C = D1 * D2 * f;
            double disc = B * B - 4.0 * A * C;
            if (disc < 0.0) disc = 0.0;
            const double sq = sqrt(disc);
            if (A == 0.0) {
                if (B != 0.0) eta = C / B;
            } else {
                double r1, r2;
                if (B >= 0.0) {
                    r2 = (B + sq) / (2.0 * A);
                    r1 = (B + sq) != 0.0 ? (2.0 * C) / (B + sq) : (B - sq) / (2.0 * A);
                } else {
                    r1 = (B - sq) / (2.0 * A);
                    r2 = (B - sq) != 0.0 ? (2.0 * C) / (B - sq) : (B + sq) / (2.0 * A);
                }
                const bool ok1 = isfinite(r1) && r1 > D1 && r1 < D2;
                const bool ok2 = isfinite(r2) && r2 > D1 && r2 < D2;
                if (ok1 && ok2) eta = fabs(r1) <= fabs(r2) ? r1 : r2;
                else if (ok1) eta = r1;
                else if (ok2) eta = r2;
            }
        }
        double nw = mu + eta;
        if (!isfinite(nw) || nw <= lo_b || nw >= hi_b) {
            if (lo_b > 0.0 && hi_b / lo_b > 16.0) nw = sqrt(lo_b * hi_b);
            else if (hi_b < 0.0 && lo_b / hi_b > 16.0) nw = -sqrt(lo_b * hi_b);
            else {
                nw = 0.5 * (lo_b + hi_b);
                if (nw == lo_b || nw == hi_b) break;
            }
        }
        mu = nw;
    }
    if (lane == 0) {
        w.org[lo + i] = org;
        w.mu[lo + i] = mu;
        w.lam[lo + i] = dorg + mu;
    }
}

// One wave: zhat_i = sign(z_i) sqrt( prod_j (lam_j - d_i) / (rho prod_{j != i} (d_j - d_i)) )
__device__ __forceinline__ void dc_zhat_one(const DcWork &w, const int lo, const int K, const double rho, const int i) {
    const int lane = threadIdx.x & 63;
    const double *__restrict__ dk = w.dk + lo;
    const double di = dk[i];
    double p = 1.0;
    for (int j = lane; j < K; j += 64) {
        const double num = (dk[w.org[lo + j]] - di) + w.mu[lo + j];       // lam_j - d_i
        const double den = (j == i) ? 1.0 : dk[j] - di;
        p *= num / den;
    }
    p = wave_prod(p);
    if (lane == 0) {
        const double zh = sqrt(fabs(p / rho));
        w.zhat[lo + i] = (w.zk[lo + i] >= 0.0) ? zh : -zh;
    }
}

// One wave: 1 / || zhat_i / (d_i - lam_j) ||_2 for root j
__device__ __forceinline__ void dc_colnorm_one(const DcWork &w, const int lo, const int K, const int j) {
    const int lane = threadIdx.x & 63;
    const double *__restrict__ dk = w.dk + lo;
    const double dorg = dk[w.org[lo + j]], muj = w.mu[lo + j];
    double s = 0.0;
    for (int i = lane; i < K; i += 64) {
        const double u = w.zhat[lo + i] / ((dk[i] - dorg) - muj);
        s += u * u;
    }
    s = wave_sum(s);
    if (lane == 0) w.invn[lo + j] = 1.0 / sqrt(s);
}

__device__ __forceinline__ double dc_u_elem(const DcWork &w, const int lo, const int i, const int j) {
    const double *__restrict__ dk = w.dk + lo;
    return w.zhat[lo + i] / ((dk[i] - dk[w.org[lo + j]]) - w.mu[lo + j]) * w.invn[lo + j];
}

// rank-sort {roots} U {deflated poles}: dnext ascending; rank / source tables into rota / rotb.  Whole workgroup.
template <int NT>
__device__ void dc_rank_body(const DcWork &w, const Seg sg, const int K, double *val, int *src) {
    const int lo = sg.lo, N = sg.hi - sg.lo;
    const int tid = threadIdx.x;
    for (int t = tid; t < N; t += NT) {
        if (t < K) {
            val[t] = w.lam[lo + t];
            src[t] = -1 - t;                         // column t of Ww
        } else {
            const int idx = w.deflidx[lo + (t - K)];
            val[t] = w.dwork[lo + idx];
            src[t] = idx;                            // column idx of Qcur
        }
    }
    __syncthreads();
    for (int t = tid; t < N; t += NT) {
        const double v = val[t];
        int rk = 0;
        for (int u = 0; u < N; ++u) {
            const double x = val[u];
            rk += (x < v) || (!(v < x) && u < t);           // a valid permutation even for unordered (NaN) values
        }
        w.dnext[lo + rk] = v;
        w.rota[lo + t] = rk;                         // the rotation list has been consumed: reuse as tables
        w.rotb[lo + t] = src[t];
    }
}

// rows [r0, r1): Qnext[:, rank[t]] = root ? Ww[:, t] : Qcur[:, deflated idx]
template <int NT>
__device__ void dc_place_body(const DcWork &w, const Seg sg, const int r0, const int r1) {
    const int lo = sg.lo, n = w.n, N = sg.hi - sg.lo;
    const int rows = r1 - r0;
    for (int idx = threadIdx.x; idx < rows * N; idx += NT) {
        const int r = r0 + idx / N, t = idx % N;
        const int sc = w.rotb[lo + t];
        const double v = (sc < 0) ? w.Ww[(long)r * n + lo + (-1 - sc)] : w.Qcur[(long)r * n + lo + sc];
        w.Qnext[(long)r * n + lo + w.rota[lo + t]] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// per-phase kernels for the large levels (grid: x = work tiles, y = merge, z = problem)
// ------------------------------------------------------------------------------------------------------------------
#define DC_PROLOGUE                                                \
    int cls, rep;                                                  \
    const DcWork w = dc_resolve(L, blockIdx.z, cls, rep);          \
    const int m = blockIdx.y;                                      \
    if (m >= L.nseg[cls]) return;                                  \
    const Seg sg = load_seg(w, L.seg_off[cls], m);

__global__ __launch_bounds__(256) void dc_setup_kernel(DcLevel L) {
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    DC_PROLOGUE
    __shared__ SetupShared S;
    dc_setup_body<4>(w, sg, m, S);
}

constexpr int ROT_ROWS = 32;

__global__ __launch_bounds__(256) void dc_place_kernel(DcLevel L) {
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    DC_PROLOGUE
    const int r0 = sg.lo + blockIdx.x * 4;
    if (r0 >= sg.hi) return;
    dc_place_body<256>(w, sg, r0, min(r0 + 4, sg.hi));
}

// ---- large levels, fused launches: phases that only share their inputs run as one launch with a role per workgroup
// (every launch is a dependent ~5 us step of the chain).
//   A: rotations + compaction of Q2 | secular equation           (both need the deflation scan only)
//   B: z-hat                         | rank sort of roots + poles (both need the roots only)
//   C: column norms                  | UNNORMALISED U             (both need z-hat; the norms scale the GEMM's columns)
__global__ __launch_bounds__(256) void dc_rotsec_kernel(DcLevel L, int rot_blocks) {
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    DC_PROLOGUE
    const int K = w.meta[2 * m];
    if ((int)blockIdx.x < rot_blocks) {
        const int r0 = sg.lo + blockIdx.x * ROT_ROWS;
        if (r0 >= sg.hi) return;
        dc_rotate_compact_body<256>(w, sg, K, w.meta[2 * m + 1], r0, min(r0 + ROT_ROWS, sg.hi));
    } else {
        const int i = ((int)blockIdx.x - rot_blocks) * 4 + (threadIdx.x >> 6);
        if (i >= K) return;
        dc_secular_root(w, sg.lo, K, 2.0 * fabs(w.e[sg.mid - 1]), i);
    }
}

__global__ __launch_bounds__(256) void dc_zhat_rank_kernel(DcLevel L, int zhat_blocks) {
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    DC_PROLOGUE
    const int K = w.meta[2 * m];
    if ((int)blockIdx.x < zhat_blocks) {
        const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i >= K) return;
        dc_zhat_one(w, sg.lo, K, 2.0 * fabs(w.e[sg.mid - 1]), i);
    } else {
        __shared__ double val[EIG_MAXN];
        __shared__ int src[EIG_MAXN];
        dc_rank_body<256>(w, sg, K, val, src);       // consumes rota / rotb: the rotations ran in the previous launch
    }
}

__global__ __launch_bounds__(256) void dc_colnorm_U_kernel(DcLevel L, int norm_blocks, int tiles_j) {
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    DC_PROLOGUE
    const int lo = sg.lo, n = w.n;
    const int K = w.meta[2 * m];
    if ((int)blockIdx.x < norm_blocks) {
        const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (j >= K) return;
        dc_colnorm_one(w, lo, K, j);
    } else {
        const int bx = (int)blockIdx.x - norm_blocks;
        const int j = (bx % tiles_j) * 64 + (threadIdx.x & 63);
        const int i0 = (bx / tiles_j) * 16 + (threadIdx.x >> 6) * 4;
        if (j >= K) return;
        const double *__restrict__ dk = w.dk + lo;
        const double dorg = dk[w.org[lo + j]], muj = w.mu[lo + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if (i < K) w.Uw[(long)(lo + i) * n + lo + j] = w.zhat[lo + i] / ((dk[i] - dorg) - muj);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// fused level kernel for small merges (N <= DC_SMALL): one workgroup walks every phase of its merge
// ------------------------------------------------------------------------------------------------------------------
// ---- helpers of the fused kernel: everything between the deflation scan and the placement stays in LDS / registers.
// fast_rcp (devutil.hpp: v_rcp_f64 + two Newton steps, 5 instructions instead of the ~35 of an IEEE division) is used where
// the result feeds a sum whose rounding error is of the same order anyway.
// sum / product over the 8 lanes of an aligned lane octet (two quads), result in all 8 lanes
__device__ __forceinline__ double oct_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}
__device__ __forceinline__ double oct_prod(double v) {
    v *= dpp_mov<0xB1>(v);
    v *= dpp_mov<0x4E>(v);
    v *= dpp_mov<0x141>(v);
    return v;
}

template <int SM>
struct SmallSharedT {
    double kd[SM], kz[SM];                      // non-deflated poles / weights
    double mu[SM], lam[SM], zh[SM];
    int org[SM];
};

// Secular equation, EIGHT lanes per root (octet o = root, lane s of the octet sums poles s, s+8, ...): all K <= 64 roots
// of the merge advance together on the 512 threads.  Same iteration as dc_secular_root (origin shift to the nearer pole,
// two-pole rational step, bisection safeguard); the step itself uses fast reciprocals, the convergence test does not
// depend on them.
template <class SQ>
__device__ void dc_secular_oct(const SQ &Q, SQ &Qw, const int K, const double rho) {
    const int i = threadIdx.x >> 3, s = threadIdx.x & 7;
    const double *__restrict__ dk = Q.kd;
    const double *__restrict__ zk = Q.kz;
    if (i >= K) return;                               // whole octets leave together
    if (K == 1) {
        if (s == 0) {
            const double m = rho * zk[0] * zk[0];
            Qw.org[0] = 0;
            Qw.mu[0] = m;
            Qw.lam[0] = dk[0] + m;
        }
        return;
    }
    const bool last = (i == K - 1);
    int org;
    double lo_b, hi_b;
    if (last) {
        org = K - 1;
        double t = 0.0;
        for (int j = s; j < K; j += 8) t += zk[j] * zk[j];
        lo_b = 0.0;
        hi_b = rho * oct_sum(t);
    } else {
        const double di = dk[i];
        const double half = 0.5 * (dk[i + 1] - di);
        double t = 0.0;
        for (int j = s; j < K; j += 8) t += zk[j] * zk[j] / ((dk[j] - di) - half);
        const double fmid = 1.0 + rho * oct_sum(t);
        if (fmid >= 0.0) {
            org = i;
            lo_b = 0.0;
            hi_b = half;
        } else {
            org = i + 1;
            lo_b = -half;
            hi_b = 0.0;
        }
    }
    const double dorg = dk[org];
    const double pl = dk[i] - dorg;
    const double pr = last ? 0.0 : dk[i + 1] - dorg;
    // this lane's poles, shifted to the origin, stay in registers
    double del[8], zz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = s + 8 * q;
        del[q] = (j < K) ? dk[j] - dorg : 1e300;            // padding: rinv ~ 1e-300, weight 0
        zz[q] = (j < K) ? zk[j] * zk[j] : 0.0;
    }
    double mu = last ? 0.5 * hi_b : 0.5 * (lo_b + hi_b);
    for (int it = 0; it < 100; ++it) {
        double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double rinv = fast_rcp(del[q] - mu);
            const double term = zz[q] * rinv;
            if (s + 8 * q <= i) {
                psi += term;
                dpsi += term * rinv;
            } else {
                phi += term;
                dphi += term * rinv;
            }
        }
        psi = rho * oct_sum(psi);
        dpsi = rho * oct_sum(dpsi);
        phi = rho * oct_sum(phi);
        dphi = rho * oct_sum(dphi);
        const double f = 1.0 + psi + phi;
        const double err = 16.0 * EPS_U * (1.0 + fabs(psi) + fabs(phi)) + fabs(mu) * EPS_U * (dpsi + dphi);
        if (fabs(f) <= err) break;
        if (f < 0.0) lo_b = fmax(lo_b, mu);
        else hi_b = fmin(hi_b, mu);
        if (hi_b - lo_b <= 2.0 * EPS_U * fmax(fabs(lo_b), fabs(hi_b))) break;
        const double D1 = pl - mu;
        double eta = INFINITY;
        if (last) {
            const double g = 1.0 + psi - dpsi * D1;
            if (g > 0.0) eta = D1 + dpsi * D1 * D1 * fast_rcp(g);
        } else {
            const double D2 = pr - mu;
            const double A = f - dpsi * D1 - dphi * D2;
            const double B = A * (D1 + D2) + dpsi * D1 * D1 + dphi * D2 * D2;
            const double C = D1 * D2 * f;
            double disc = B * B - 4.0 * A * C;
            if (disc < 0.0) disc = 0.0;
            const double sq = sqrt(disc);
            if (A == 0.0) {
                if (B != 0.0) eta = C * fast_rcp(B);
            } else {
                const double bs = (B >= 0.0) ? B + sq : B - sq;          // the larger-magnitude combination
                const double ia = fast_rcp(2.0 * A);
                const double r_big = bs * ia;                             // (B +- sq) / 2A
                const double r_small = (bs != 0.0) ? 2.0 * C * fast_rcp(bs) : ((B >= 0.0) ? B - sq : B + sq) * ia;
                const bool ok1 = isfinite(r_small) && r_small > D1 && r_small < D2;
                const bool ok2 = isfinite(r_big) && r_big > D1 && r_big < D2;
                if (ok1 && ok2) eta = fabs(r_small) <= fabs(r_big) ? r_small : r_big;
                else if (ok1) eta = r_small;
                else if (ok2) eta = r_big;
            }
        }
        double nw = mu + eta;
        if (!isfinite(nw) || nw <= lo_b || nw >= hi_b) {
            if (lo_b > 0.0 && hi_b / lo_b > 16.0) nw = sqrt(lo_b * hi_b);
            else if (hi_b < 0.0 && lo_b / hi_b > 16.0) nw = -sqrt(lo_b * hi_b);
            else {
                nw = 0.5 * (lo_b + hi_b);
                if (nw == lo_b || nw == hi_b) break;
            }
        }
        mu = nw;
    }
    if (s == 0) {
        Qw.org[i] = org;
        Qw.mu[i] = mu;
        Qw.lam[i] = dorg + mu;
    }
}

typedef double dc_d4 __attribute__((ext_vector_type(4)));

// Everything after the deflation scan of a small merge, on LDS-resident data: roots (8 lanes each), z-hat (8 lanes per
// pole), U with normalised columns, W = Q2 U on fp64 MFMA 16x16x4 (operands from LDS), rank sort, placement.
template <int SM>
__device__ void dc_small_tail(const DcWork &w, const Seg sg, SetupSharedSmallT<SM> &S, SmallSharedT<SM> &Q, double *Q2s, const int K,
                              const double rho) {
    constexpr int NT = 8 * SM, Q2_LD = SM + 1;
    const int lo = sg.lo, hi = sg.hi, n = w.n, N = hi - lo;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int o = tid >> 3, s = tid & 7;                  // octet index (root / pole / column), lane inside the octet
    dc_secular_oct(Q, Q, K, rho);
    __syncthreads();
    if (tid < K) w.lam[lo + tid] = Q.lam[tid];            // the rank sort reads the roots from global memory
    if (o < K) {                                          // zhat_o = sign(z_o) sqrt( prod_j (lam_j - d_o) / (rho prod_{j != o} (d_j - d_o)) )
        const double di = Q.kd[o];
        double p = 1.0;
        for (int j = s; j < K; j += 8) {
            const double num = (Q.kd[Q.org[j]] - di) + Q.mu[j];
            const double den = (j == o) ? 1.0 : Q.kd[j] - di;
            p *= num / den;
        }
        p = oct_prod(p);
        if (s == 0) {
            const double zh = sqrt(fabs(p / rho));
            Q.zh[o] = (Q.kz[o] >= 0.0) ? zh : -zh;
        }
    }
    __syncthreads();
    double *U = S.sd;                                     // sd | sz | ds | zs: 4 * (SM * SM / 4) doubles = SM x SM
    {                                                     // column o of U, rows s, s+8, ...; zero padded to SM x SM
        double u[SM / 8], ss = 0.0;
        const bool colok = o < K;
        const double dorg = colok ? Q.kd[Q.org[o]] : 0.0, muo = colok ? Q.mu[o] : 0.0;
#pragma unroll
        for (int q = 0; q < SM / 8; ++q) {
            const int i = s + 8 * q;
            u[q] = (colok && i < K) ? Q.zh[i] / ((Q.kd[i] - dorg) - muo) : 0.0;
            ss += u[q] * u[q];
        }
        ss = oct_sum(ss);
        const double inv = colok ? 1.0 / sqrt(ss) : 0.0;
#pragma unroll
        for (int q = 0; q < SM / 8; ++q) U[(s + 8 * q) * SM + o] = u[q] * inv;
    }
    for (int idx = tid; idx < SM * SM; idx += NT) {
        const int r = idx / SM, j = idx % SM;
        Q2s[r * Q2_LD + j] = (r < N && j < K) ? w.Q2w[(long)(lo + r) * n + lo + j] : 0.0;
    }
    __syncthreads();
    // W = Q2 U: (SM / 16)^2 output fragments of 16 x 16 over the SM / 8 waves -- two per wave at SM = 64, one at 32 (A lane l holds
    // A[l&15][l>>4], B holds B[l>>4][l&15])
    constexpr int TPD = SM / 16, NWV = SM / 8, TPW = (TPD * TPD) / NWV;
    static_assert(TPW >= 1 && TPW * NWV == TPD * TPD, "fragments must divide evenly over the waves");
    dc_d4 acc[TPW];
    const int fr = lane & 15, fq = lane >> 4;
    const int ksteps = (K + 3) >> 2;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wid + NWV * t, tm = tile / TPD, tn = tile % TPD;
        acc[t] = dc_d4{0.0, 0.0, 0.0, 0.0};
        if (16 * tm < N && 16 * tn < K) {                 // wave-uniform
            const double *qa = Q2s + (16 * tm + fr) * Q2_LD + fq;
            const double *ub = U + fq * SM + 16 * tn + fr;
            for (int ks = 0; ks < ksteps; ++ks)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[4 * ks], ub[4 * ks * SM], acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                      // every fragment read of Q2s is done: reuse it for W
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wid + NWV * t, tm = tile / TPD, tn = tile % TPD;
#pragma unroll
        for (int r = 0; r < 4; ++r) Q2s[(16 * tm + fq + 4 * r) * Q2_LD + 16 * tn + fr] = acc[t][r];
    }
    __syncthreads();
    dc_rank_body<NT>(w, sg, K, S.sd, S.sperm);            // U is dead: its storage holds the sort keys
    __syncthreads();
    for (int idx = tid; idx < N * N; idx += NT) {         // Qnext[:, rank[t]] = root ? W[:, t] : Qcur[:, deflated idx]
        const int r = idx / N, t = idx % N;
        const int sc = w.rotb[lo + t];
        const double v = (sc < 0) ? Q2s[r * Q2_LD + (-1 - sc)] : w.Qcur[(long)(lo + r) * n + lo + sc];
        w.Qnext[(long)(lo + r) * n + lo + w.rota[lo + t]] = v;
    }
}

// SM = 64: 512 threads, 74 KB of LDS.  SM = 32 (round 5: merges of up to 32 rows -- the two lowest levels of a 192- or 250-row
// problem): 256 threads, 20 KB -- a workgroup the size of a GEMM tile's.  The levels run under the previous call's fused last
// product, whose workgroups leave 43 KB of LDS and 120 registers per lane free on a CU: the 64-row form has to wait for two of
// them to retire together (measured 87 + 71 us for the two lowest levels of the cfg3 spatial problem, 20 + 21 alone).
template <int SM>
__global__ __launch_bounds__(8 * SM) void dc_small_level_kernel(DcLevel L) {
    int cls, rep;
    const DcWork w = dc_resolve(L, blockIdx.z, cls, rep);
    const int m = blockIdx.x;
    if (m >= L.nseg[cls]) return;
    const Seg sg = load_seg(w, L.seg_off[cls], m);
    constexpr int NT = 8 * SM, NW = NT / 64;
    // (a chain of dependent scalar work beside a flood of GEMM waves: its instructions go first on the SIMDs they share)
    if (L.prio) __builtin_amdgcn_s_setprio(3);
    __shared__ SetupSharedSmallT<SM> S;
    __shared__ SmallSharedT<SM> Q;
    __shared__ double Q2s[SM * (SM + 1)];
    const int lo = sg.lo, hi = sg.hi;
    const int tid = threadIdx.x;
    dc_setup_body<NW>(w, sg, m, S);
    __syncthreads();
    const int K = S.K, nrot = S.nrot;
    dc_rotate_compact_body<NT>(w, sg, K, nrot, lo, hi);
    const double rho = 2.0 * fabs(w.e[sg.mid - 1]);
    if (tid < SM) {
        Q.kd[tid] = (tid < K) ? w.dk[lo + tid] : 0.0;
        Q.kz[tid] = (tid < K) ? w.zk[lo + tid] : 0.0;
    }
    __syncthreads();
    dc_small_tail<SM>(w, sg, S, Q, Q2s, K, rho);
}
static_assert(offsetof(SetupSharedSmallT<64>, zs) == 3 * (64 * 64 / 4) * sizeof(double) &&
                  offsetof(SetupSharedSmallT<32>, zs) == 3 * (32 * 32 / 4) * sizeof(double),
              "U tile must fit the reused, contiguous setup arrays");

__global__ void copy_vec_kernel(const double *__restrict__ a, double *__restrict__ b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

// ------------------------------------------------------------------------------------------------------------------
// host side: plan (depends only on n), workspace, level loop
// ------------------------------------------------------------------------------------------------------------------
struct DcPlan {
    int n = 0;
    std::vector<int> leaf_lo;                 // leaves + 1 entries
    std::vector<int> bounds;                  // interior leaf boundaries (tears)
    std::vector<std::vector<Seg>> levels;     // bottom-up
    std::vector<int> tbl;                     // leaf_lo | bounds | level triples
    int off_bounds = 0;
    std::vector<int> lvl_off;
};

static DcPlan make_plan(int n) {
    DcPlan p;
    p.n = n;
    std::vector<std::pair<int, int>> segs = {{0, n}};
    std::vector<std::vector<Seg>> rev;
    for (;;) {
        int mx = 0;
        for (auto &sg : segs) mx = std::max(mx, sg.second - sg.first);
        if (mx <= DC_LEAF) break;
        std::vector<std::pair<int, int>> nxt;
        std::vector<Seg> lvl;
        for (auto &sg : segs) {
            const int m = (sg.first + sg.second) / 2;
            nxt.push_back({sg.first, m});
            nxt.push_back({m, sg.second});
            lvl.push_back(Seg{sg.first, m, sg.second});
        }
        rev.push_back(lvl);
        segs = nxt;
    }
    for (auto &sg : segs) p.leaf_lo.push_back(sg.first);
    p.leaf_lo.push_back(n);
    for (size_t i = 1; i + 1 < p.leaf_lo.size(); ++i) p.bounds.push_back(p.leaf_lo[i]);
    p.levels.assign(rev.rbegin(), rev.rend());
    p.tbl.insert(p.tbl.end(), p.leaf_lo.begin(), p.leaf_lo.end());
    p.off_bounds = (int)p.tbl.size();
    p.tbl.insert(p.tbl.end(), p.bounds.begin(), p.bounds.end());
    for (auto &lv : p.levels) {
        p.lvl_off.push_back((int)p.tbl.size());
        for (auto &sg : lv) {
            p.tbl.push_back(sg.lo);
            p.tbl.push_back(sg.mid);
            p.tbl.push_back(sg.hi);
        }
    }
    return p;
}

struct StedcProb {               // one class of tridiagonal problems (replicas: d / e s_in apart, outputs sw / sZ apart)
    const double *d, *e;
    int n;
    double *w, *Z;
    std::string tag;
    int count = 1;
    long s_in = 0, sw = 0, sZ = 0;
};

// Workspace of a class as ONE arena of `count` identical blocks (dc_resolve adds replica * blk to every pointer); the int
// slices are carved out of the same arena (two ints per double, 16-byte aligned slices).
static DcWork make_work(gpcsd_ctx *c, const StedcProb &p, const DcPlan &plan, hipStream_t s) {
    const int n = p.n;
    const size_t nn = (size_t)n * n;
    const std::string T = "dc_" + p.tag + "_";
    size_t off = 0;
    auto take = [&](size_t ndoubles) {
        const size_t o = off;
        off += (ndoubles + 1) & ~(size_t)1;
        return o;
    };
    auto take_int = [&](size_t nints) { return take((nints + 1) / 2); };
    const size_t o_dcur = take(n), o_dnext = take(n), o_Qcur = take(nn), o_Qnext = take(nn), o_dwork = take(n), o_dk = take(n),
                 o_zk = take(n), o_mu = take(n), o_lam = take(n), o_zhat = take(n), o_invn = take(n), o_rotc = take(n),
                 o_rots = take(n), o_Q2w = take(nn), o_Uw = take(nn), o_Ww = take(nn);
    const size_t o_org = take_int(n), o_ndidx = take_int(n), o_deflidx = take_int(n), o_rota = take_int(n), o_rotb = take_int(n),
                 o_meta = take_int(2 * n + 2), o_Kdyn = take_int(n + 2);
    double *base = c->buf<double>(T + "arena", off * (size_t)std::max(p.count, 1));
    DcWork w;
    w.n = n;
    w.d0 = p.d;
    w.e = p.e;
    w.dcur = base + o_dcur; w.dnext = base + o_dnext; w.Qcur = base + o_Qcur; w.Qnext = base + o_Qnext;
    w.dwork = base + o_dwork; w.dk = base + o_dk; w.zk = base + o_zk; w.mu = base + o_mu; w.lam = base + o_lam;
    w.zhat = base + o_zhat; w.invn = base + o_invn; w.rotc = base + o_rotc; w.rots = base + o_rots;
    w.Q2w = base + o_Q2w; w.Uw = base + o_Uw; w.Ww = base + o_Ww;
    auto ip = [&](size_t o) { return reinterpret_cast<int *>(base + o); };
    w.org = ip(o_org); w.ndidx = ip(o_ndidx); w.deflidx = ip(o_deflidx); w.rota = ip(o_rota); w.rotb = ip(o_rotb);
    w.meta = ip(o_meta); w.Kdyn = ip(o_Kdyn);
    w.blk = (long)off;
    w.s_in = p.s_in; w.sw = p.sw; w.sZ = p.sZ;
    w.top = 0;
    // the table depends only on n: its own buffer per (tag, n), uploaded once.  (One buffer per tag that was re-uploaded when
    // n changed put a synchronising copy inside a later stream capture of a chain whose eager run had seen the other n.)
    const std::string PN = T + "plan_" + std::to_string(n);
    int *dtbl = c->buf<int>(PN, plan.tbl.size() + 4);
    int &cached_n = c->int_cache[PN];
    if (cached_n != n) {
        c->copy_in(dtbl, plan.tbl.data(), plan.tbl.size() * sizeof(int), s);
        GP_HIP(hipStreamSynchronize(s));  // plan is a temporary of the caller
        cached_n = n;
    }
    w.tbl = dtbl;
    w.wout = p.w;
    w.Zout = p.Z;
    return w;
}

void stedc_batch_device(gpcsd_ctx *c, StedcProb *probs, int nclass, int *d_status, int status_stride, hipStream_t s,
                        const WyBatch *wy) {
    GP_REQUIRE(nclass >= 1 && nclass <= MAX_BATCH, -3, "stedc: %d problem classes outside [1,%d]", nclass, MAX_BATCH);
    std::vector<DcPlan> plans;
    DcLevel L{};
    {
        static const bool prio_off = getenv("GPCSD_DC_PRIO") && getenv("GPCSD_DC_PRIO")[0] == '0';
        L.prio = prio_off ? 0 : 1;
    }
    L.status_stride = status_stride;
    size_t nlevels = 0;
    int nmax = 0, max_leaves = 0, count = 0;
    for (int p = 0; p < MAX_BATCH; ++p) {
        L.start[p] = count;
        if (p >= nclass) continue;
        GP_REQUIRE(probs[p].n >= 1 && probs[p].n <= EIG_MAXN, -3, "stedc: n=%d outside [1,%d]", probs[p].n, EIG_MAXN);
        plans.push_back(make_plan(probs[p].n));
        L.w[p] = make_work(c, probs[p], plans[p], s);
        nlevels = std::max(nlevels, plans[p].levels.size());
        nmax = std::max(nmax, probs[p].n);
        max_leaves = std::max(max_leaves, (int)plans[p].leaf_lo.size() - 1);
        count += std::max(probs[p].count, 1);
    }
    L.start[MAX_BATCH] = count;                 // workgroup rows / planes of every launch: all replicas of all classes
    // tears + leaves (one launch: dc_leaf_reg_body applies the tears and zero-fills the off-diagonal blocks)
    static_assert(DC_LEAF == 8, "the register leaf solver maps an 8 x 8 block onto one wave");
    for (int p = 0; p < nclass; ++p) L.aux[p] = (int)plans[p].leaf_lo.size() - 1;
    if (wy) {                                  // the T factors of the back-transformation ride in the leaf launch
        int nprep = 0;
        for (int p = 0; p < nclass; ++p) nprep = std::max(nprep, wy->p[p].npanels);
        const int nunits = max_leaves + nmax;
        hipLaunchKernelGGL(dc_leaf_wyprep_kernel, dim3(nprep + ceil_div(nunits, 16), count), dim3(1024), 0, s, L, d_status,
                           max_leaves, nunits, *wy, nprep);
    } else {
        hipLaunchKernelGGL(dc_leaf_reg_kernel, dim3(max_leaves + nmax, count), dim3(64), 0, s, L, d_status, max_leaves);
    }
    GP_HIP(hipGetLastError());

    // Problems with fewer levels idle at the bottom: level index counts from the TOP so the final merges align.
    for (size_t li = 0; li < nlevels; ++li) {
        int maxN = 0, max_seg = 0;
        for (int p = 0; p < nclass; ++p) {
            const long own = (long)plans[p].levels.size() - (long)(nlevels - li);   // this problem's level index, or < 0
            if (own < 0) {
                L.nseg[p] = 0;
                L.seg_off[p] = 0;
                continue;
            }
            const auto &lv = plans[p].levels[own];
            L.nseg[p] = (int)lv.size();
            L.seg_off[p] = plans[p].lvl_off[own];
            max_seg = std::max(max_seg, (int)lv.size());
            for (auto &sg : lv) maxN = std::max(maxN, sg.hi - sg.lo);
        }
        if (li + 1 == nlevels)                 // the top merge covers the whole matrix: placed in the caller's arrays (dc_resolve)
            for (int p = 0; p < nclass; ++p)
                if (L.nseg[p] > 0) L.w[p].top = 1;
        static const bool small32 = !(getenv("GPCSD_DC_SMALL32") && getenv("GPCSD_DC_SMALL32")[0] == '0');     // (A/B)
        if (maxN <= 32 && small32) {
            hipLaunchKernelGGL(dc_small_level_kernel<32>, dim3(max_seg, 1, count), dim3(8 * 32), 0, s, L);
        } else if (maxN <= DC_SMALL) {
            hipLaunchKernelGGL(dc_small_level_kernel<DC_SMALL>, dim3(max_seg, 1, count), dim3(NT_SMALL), 0, s, L);
        } else {
            hipLaunchKernelGGL(dc_setup_kernel, dim3(1, max_seg, count), dim3(256), 0, s, L);
            const int rot_blocks = ceil_div(maxN, ROT_ROWS), root_blocks = ceil_div(maxN, 4), tj = ceil_div(maxN, 64);
            hipLaunchKernelGGL(dc_rotsec_kernel, dim3(rot_blocks + root_blocks, max_seg, count), dim3(256), 0, s, L, rot_blocks);
            hipLaunchKernelGGL(dc_zhat_rank_kernel, dim3(root_blocks + 1, max_seg, count), dim3(256), 0, s, L, root_blocks);
            hipLaunchKernelGGL(dc_colnorm_U_kernel, dim3(root_blocks + tj * ceil_div(maxN, 16), max_seg, count), dim3(256), 0, s, L,
                               root_blocks, tj);
            GP_HIP(hipGetLastError());
            // W = Q2 (N x K) U (K x K) diag(invn), K read on the device.  The merges of one class are one batched launch
            // when they have the same size (their blocks sit a constant stride apart on the diagonal); the replicas of the
            // class are the outer batch level (arena stride).
            for (int p = 0; p < nclass; ++p) {
                const long own = (long)plans[p].levels.size() - (long)(nlevels - li);
                if (own < 0) continue;
                const auto &lv = plans[p].levels[own];
                const DcWork &w = L.w[p];
                const int n = w.n;
                bool uniform = true;
                for (size_t m = 1; m < lv.size(); ++m)
                    uniform = uniform && (lv[m].hi - lv[m].lo == lv[0].hi - lv[0].lo) && (lv[m].lo - lv[m - 1].lo == lv[1].lo - lv[0].lo);
                const int nlaunch = uniform ? 1 : (int)lv.size();
                for (int m = 0; m < nlaunch; ++m) {
                    const Seg &sg = lv[m];
                    const int N = sg.hi - sg.lo;
                    GemmDesc g;
                    g.M = N; g.N = N; g.K = N;
                    g.A = w.Q2w + (long)sg.lo * n + sg.lo; g.lda = n;
                    g.B = w.Uw + (long)sg.lo * n + sg.lo; g.ldb = n;
                    g.C = w.Ww + (long)sg.lo * n + sg.lo; g.ldc = n;
                    g.colscale = w.invn + sg.lo;
                    g.dyn = w.Kdyn + m;
                    if (uniform && lv.size() > 1) {
                        const long step = lv[1].lo - lv[0].lo;
                        g.batch = (int)lv.size();
                        g.sA = g.sB = g.sC = step * n + step;
                        g.sColscale = step;
                    }
                    g.batch2 = L.start[p + 1] - L.start[p];
                    g.sA2 = g.sB2 = g.sC2 = g.sColscale2 = w.blk;
                    g.sDyn2 = 2 * w.blk;
                    g.prof_name = "gemm_dc_merge";
                    gemm_f64(c, g, s);
                }
            }
            hipLaunchKernelGGL(dc_place_kernel, dim3(ceil_div(maxN, 4), max_seg, count), dim3(256), 0, s, L);
        }
        GP_HIP(hipGetLastError());
        for (int p = 0; p < nclass; ++p) {
            if (L.nseg[p] == 0) continue;
            std::swap(L.w[p].dcur, L.w[p].dnext);
            std::swap(L.w[p].Qcur, L.w[p].Qnext);
        }
    }
    bool need_copy = false;                    // only a problem without any merge level (n <= DC_LEAF) still needs the copy
    for (int p = 0; p < nclass; ++p) need_copy = need_copy || plans[p].levels.empty();
    if (need_copy) hipLaunchKernelGGL(dc_output_kernel, dim3(64, count), dim3(256), 0, s, L);
    GP_HIP(hipGetLastError());
}

void stedc_device(gpcsd_ctx *c, const double *d, const double *e, int n, double *wout, double *Zout, int *d_status,
                  hipStream_t s, const char *tag) {
    StedcProb p;
    p.d = d; p.e = e; p.n = n; p.w = wout; p.Z = Zout; p.tag = tag ? tag : "";
    stedc_batch_device(c, &p, 1, d_status, 0, s, nullptr);
}

}  // namespace gpcsd
