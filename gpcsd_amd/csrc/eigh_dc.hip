// Large-n symmetric eigensolver for gfx950 (replaces LAPACK dsyevd behind numpy.linalg.eigh at
// src/gpcsd/utility_functions.py:58-59).  Three stages, all on the device:
//
//  1. sytrd  -- Householder tridiagonalisation, ONE launch per column, batched over independent problems.
//               Every workgroup owns a slab of trailing rows (full symmetric storage, ping-pong buffers); it redundantly
//               rebuilds w from the previous step's y = A v, applies the rank-2 update to its slab, derives the next
//               reflector from the updated pivot row and accumulates its part of the next y in the same pass.  No
//               atomics, no grid barriers, deterministic; the only cross-workgroup hand-off is the kernel boundary.
//  2. stedc  -- Cuppen divide & conquer on the tridiagonal: Jacobi leaves in LDS, then per level: deflation scan,
//               secular equation (one wave per root, origin-shifted), Gu/Eisenstat z-hat (Loewner) for orthogonality,
//               eigenvector block via the fp64 MFMA GEMM with the deflated size read on the device, rank-sort merge.
//  3. ormtr  -- back-transformation Z <- (H_0 ... H_{n-3}) Z in compact-WY panels of 64 reflectors: T factors for all
//               panels in one batched GEMM + one launch, then two MFMA GEMMs per panel.
//
// tools/dc_prototype.py is the NumPy model of stage 2 used to validate the algorithm before porting.
#include <algorithm>
#include <vector>

#include "jacobi.hpp"
#include "kernels.hpp"

namespace gpcsd {

constexpr int EIG_MAXN = 1024;       // LDS vectors are sized for this
constexpr int DC_LEAF = 32;
constexpr int SY_RPW = 8;            // trailing rows per workgroup in the sytrd step kernel
constexpr int WY_NB = 64;            // reflectors per compact-WY panel
constexpr int MAX_BATCH = 4;
constexpr double EPS_U = 1.1102230246251565e-16;   // unit roundoff (dlamch('E'))

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_prod(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// sum over a 256-thread workgroup; result valid in every thread.  red: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum256(double v, double *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max256(double v, double *red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------------------------
// stage 0: scaling  A0 = A / max|A|
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const double *__restrict__ A, long n2, double *out) {
    __shared__ double red[4];
    double m = 0.0;
    for (long i = threadIdx.x; i < n2; i += 256) m = fmax(m, fabs(A[i]));
    m = block_max256(m, red);
    if (threadIdx.x == 0) out[0] = (m > 0.0 && m < 1e300) ? m : 1.0;
}
__global__ void scale_copy_kernel(const double *__restrict__ A, long n2, const double *__restrict__ amax, double *__restrict__ out) {
    const double inv = 1.0 / amax[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) out[i] = A[i] * inv;
}
__global__ void scale_vec_kernel(double *w, int n, const double *__restrict__ amax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] *= amax[0];
}

// ------------------------------------------------------------------------------------------------------------------
// stage 1: tridiagonalisation
// ------------------------------------------------------------------------------------------------------------------
struct SytrdProb {
    double *A0, *A1;      // ping-pong full symmetric storage (n x n)
    double *V;            // (n + WY_NB) x n, row k = reflector k (zero for j <= k, V[k][k+1] = 1)
    double *tau, *d, *e;  // n each
    double *y0, *y1;      // ping-pong A v products
    int n;
};
struct SytrdBatch {
    SytrdProb p[MAX_BATCH];
};

__global__ __launch_bounds__(256) void sytrd_step_kernel(SytrdBatch b, int k) {
    const SytrdProb &P = b.p[blockIdx.y];
    const int n = P.n;
    if (k > n - 2) return;
    const int row0 = k + 1 + blockIdx.x * SY_RPW;
    if (row0 >= n) return;
    const double *__restrict__ Ain = (k & 1) ? P.A1 : P.A0;
    double *__restrict__ Aout = (k & 1) ? P.A0 : P.A1;
    const double *__restrict__ yin = (k & 1) ? P.y1 : P.y0;
    double *__restrict__ yout = (k & 1) ? P.y0 : P.y1;
    __shared__ double sv[EIG_MAXN], sw[EIG_MAXN], svn[EIG_MAXN];
    __shared__ double red[4];
    __shared__ double s_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    // 1. w of the previous reflector: w = tau*y - (tau^2 (y.v)/2) v, on indices [k, n)
    if (k > 0) {
        const double *__restrict__ vp = P.V + (long)(k - 1) * n;
        const double taup = P.tau[k - 1];
        double part = 0.0;
        for (int j = k + tid; j < n; j += 256) {
            const double vj = vp[j], yj = yin[j];
            sv[j] = vj;
            sw[j] = yj;
            part += vj * yj;
        }
        const double dot = block_sum256(part, red);
        const double cc = 0.5 * taup * taup * dot;
        for (int j = k + tid; j < n; j += 256) sw[j] = taup * sw[j] - cc * sv[j];
    } else {
        for (int j = tid; j < n; j += 256) {
            sv[j] = 0.0;
            sw[j] = 0.0;
        }
    }
    __syncthreads();
    // 2. updated pivot row k -> d_k and the new reflector from x = a'[k, k+1:]
    const double vk = sv[k], wk = sw[k];
    const double *__restrict__ arow = Ain + (long)k * n;
    double part = 0.0;
    for (int j = k + tid; j < n; j += 256) {
        const double a = arow[j] - vk * sw[j] - wk * sv[j];
        svn[j] = a;
        if (j >= k + 2) part += a * a;
    }
    const double xnorm2 = block_sum256(part, red);     // (barriers inside also publish svn)
    const double dk = svn[k];
    const double alpha = svn[k + 1];
    double tau = 0.0, beta = alpha, scal = 0.0;
    if (k <= n - 3 && xnorm2 > 0.0) {
        beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
    }
    __syncthreads();
    for (int j = k + 1 + tid; j < n; j += 256) svn[j] = (j == k + 1) ? 1.0 : svn[j] * scal;
    __syncthreads();
    if (blockIdx.x == 0) {
        double *vrow = P.V + (long)k * n;
        for (int j = tid; j < n; j += 256) vrow[j] = (j >= k + 1) ? svn[j] : 0.0;
        if (tid == 0) {
            P.d[k] = dk;
            P.e[k] = beta;
            P.tau[k] = tau;
        }
    }
    // 3. slab rows: rank-2 update, store, and this slab's entries of y = A' v_new
    const int rend = min(row0 + SY_RPW, n);
    for (int i = row0 + wid; i < rend; i += 4) {
        const double vi = sv[i], wi = sw[i];
        const double *__restrict__ ai = Ain + (long)i * n;
        double *__restrict__ ao = Aout + (long)i * n;
        double acc = 0.0;
        for (int j = k + 1 + lane; j < n; j += 64) {
            const double a = ai[j] - vi * sw[j] - wi * sv[j];
            ao[j] = a;
            acc += a * svn[j];
        }
        acc = wave_sum(acc);
        if (lane == 0) yout[i] = acc;
    }
}

// d[n-1] after the final step: the single trailing element lives in the buffer written by step n-2
__global__ void sytrd_last_diag_kernel(SytrdBatch b) {
    const SytrdProb &P = b.p[blockIdx.x];
    if (threadIdx.x != 0 || P.n < 1) return;
    const int n = P.n;
    if (n == 1) {
        P.d[0] = P.A0[0];
        return;
    }
    const int k = n - 2;                                  // last step wrote Aout of parity k
    const double *Aout = (k & 1) ? P.A0 : P.A1;
    P.d[n - 1] = Aout[(long)(n - 1) * n + (n - 1)];
    P.e[n - 1] = 0.0;
    P.tau[n - 1] = 0.0;
    if (n >= 2) P.tau[n - 2] = 0.0;
}

static void sytrd_batch_launch(gpcsd_ctx *c, const SytrdBatch &b, int count, int nmax, hipStream_t s) {
    for (int k = 0; k <= nmax - 2; ++k) {
        const int m = nmax - k - 1;
        dim3 grid(ceil_div(m, SY_RPW), count);
        hipLaunchKernelGGL(sytrd_step_kernel, grid, dim3(256), 0, s, b, k);
    }
    hipLaunchKernelGGL(sytrd_last_diag_kernel, dim3(count), dim3(64), 0, s, b);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// stage 2: divide & conquer on tridiag(d, e)
// ------------------------------------------------------------------------------------------------------------------
struct Seg {
    int lo, mid, hi;
};

struct DcWork {            // per problem, all device pointers
    int n;
    double *dcur, *dnext;  // eigenvalues of the current / next level (n)
    double *Qcur, *Qnext;  // block-diagonal eigenvector matrices (n x n)
    double *dwork, *zwork; // d and z after deflation (n)
    double *dk, *zk;       // compacted non-deflated poles / weights, stored at [lo, lo+K)
    double *mu, *lam, *zhat, *invn;
    int *org, *ndidx, *deflidx, *rota, *rotb, *meta;   // meta[2*m] = K, meta[2*m+1] = nrot
    double *rotc, *rots;
    double *Q2w, *Uw, *Ww; // n x n workspaces (diagonal blocks used)
    const double *e;       // off-diagonals of the tridiagonal
    int *Kdyn;             // K per merge of the current level (device ints for the dynamic-size GEMM)
};

// tears: d[a-1] -= |e[a-1]|, d[a] -= |e[a-1]| at every leaf boundary a; copy into dcur
__global__ void dc_tear_kernel(const double *__restrict__ d, const double *__restrict__ e, int n, const int *__restrict__ bounds,
                               int nb, double *__restrict__ dout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = d[i];
    for (int t = 0; t < nb; ++t) {
        const int a = bounds[t];
        if (a - 1 == i || a == i) v -= fabs(e[a - 1]);
    }
    dout[i] = v;
}

// one workgroup per leaf: dense leaf matrix in LDS -> Jacobi -> dcur[lo:hi], Qcur diagonal block
__global__ __launch_bounds__(256) void dc_leaf_kernel(const double *__restrict__ dt, const double *__restrict__ e, int n,
                                                      const int *__restrict__ leaf_lo, double *dcur, double *Qcur, int *status) {
    const int lo = leaf_lo[blockIdx.x], hi = leaf_lo[blockIdx.x + 1];
    const int m = hi - lo;
    constexpr int LD = DC_LEAF + 1;
    __shared__ double A[DC_LEAF * LD], V[DC_LEAF * LD], cs[DC_LEAF + 2], red[256];
    __shared__ int pq[2 * DC_LEAF + 2];
    for (int idx = threadIdx.x; idx < m * m; idx += 256) {
        const int i = idx / m, j = idx % m;
        double v = 0.0;
        if (i == j) v = dt[lo + i];
        else if (j == i + 1) v = e[lo + i];
        else if (i == j + 1) v = e[lo + j];
        A[i * LD + j] = v;
    }
    __syncthreads();
    jacobi_body<256>(A, LD, V, LD, m, dcur + lo, Qcur + (long)lo * n + lo, (long)n, status, cs, pq, red);
}

// one workgroup per merge: z, merged order, deflation scan (thread 0), compacted poles
__global__ __launch_bounds__(256) void dc_merge_setup_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.x];
    const int lo = sg.lo, mid = sg.mid, hi = sg.hi, n = w.n;
    const int N = hi - lo, n1 = mid - lo;
    __shared__ double sd[EIG_MAXN], sz[EIG_MAXN];
    __shared__ int sperm[EIG_MAXN];
    __shared__ unsigned char sdefl[EIG_MAXN];
    __shared__ double red[4];
    __shared__ int s_K, s_nrot;
    const int tid = threadIdx.x;
    const double beta = w.e[mid - 1];
    const double rho = 2.0 * fabs(beta);
    const double sgn = beta >= 0.0 ? 1.0 : -1.0;
    const double isq2 = 0.70710678118654752440;
    double dmax = 0.0, zmax = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double dv = w.dcur[lo + i];
        const double zv = (i < n1 ? w.Qcur[(long)(mid - 1) * n + lo + i] : sgn * w.Qcur[(long)mid * n + lo + i]) * isq2;
        sd[i] = dv;
        sz[i] = zv;
        sdefl[i] = 0;
        dmax = fmax(dmax, fabs(dv));
        zmax = fmax(zmax, fabs(zv));
    }
    dmax = block_max256(dmax, red);
    zmax = block_max256(zmax, red);
    const double tol = 8.0 * EPS_U * fmax(dmax, zmax);
    // stable merge ranks of the two ascending halves
    for (int i = tid; i < N; i += 256) {
        const double v = sd[i];
        int cnt;
        if (i < n1) {                      // # of second-half entries strictly below v
            int a = n1, bnd = N;
            while (a < bnd) {
                const int mdl = (a + bnd) >> 1;
                if (sd[mdl] < v) a = mdl + 1; else bnd = mdl;
            }
            cnt = i + (a - n1);
        } else {                           // # of first-half entries <= v
            int a = 0, bnd = n1;
            while (a < bnd) {
                const int mdl = (a + bnd) >> 1;
                if (sd[mdl] <= v) a = mdl + 1; else bnd = mdl;
            }
            cnt = (i - n1) + a;
        }
        sperm[cnt] = i;
    }
    __syncthreads();
    if (tid == 0) {
        int K = 0, nrot = 0;
        if (rho * zmax <= tol) {
            for (int i = 0; i < N; ++i) sdefl[i] = 1;
        } else {
            int pj = -1;
            for (int jj = 0; jj < N; ++jj) {
                const int idx = sperm[jj];
                if (rho * fabs(sz[idx]) <= tol) {
                    sdefl[idx] = 1;
                    continue;
                }
                if (pj < 0) {
                    pj = idx;
                    continue;
                }
                const double t = sd[idx] - sd[pj];
                bool merged = false;
                const double zp = sz[pj], zn = sz[idx];
                // |t c s| <= tol  <=>  |t| |zp zn| <= tol (zp^2 + zn^2): division- and sqrt-free reject for the common case
                if (fabs(t) * fabs(zp * zn) <= tol * (zp * zp + zn * zn) * (1.0 + 1e-10)) {
                    double s_ = zp, c_ = zn;
                    const double tau = hypot(c_, s_);
                    c_ /= tau;
                    s_ = -s_ / tau;
                    if (fabs(t * c_ * s_) <= tol) {
                        sz[idx] = tau;
                        sz[pj] = 0.0;
                        w.rota[lo + nrot] = pj;
                        w.rotb[lo + nrot] = idx;
                        w.rotc[lo + nrot] = c_;
                        w.rots[lo + nrot] = s_;
                        ++nrot;
                        const double tt = sd[pj] * c_ * c_ + sd[idx] * s_ * s_;
                        sd[idx] = sd[pj] * s_ * s_ + sd[idx] * c_ * c_;
                        sd[pj] = tt;
                        sdefl[pj] = 1;
                        merged = true;
                    }
                }
                if (!merged) w.ndidx[lo + K++] = pj;
                pj = idx;
            }
            if (pj >= 0) w.ndidx[lo + K++] = pj;
        }
        s_K = K;
        s_nrot = nrot;
        w.meta[2 * blockIdx.x] = K;
        w.meta[2 * blockIdx.x + 1] = nrot;
        w.Kdyn[blockIdx.x] = K;
    }
    __syncthreads();
    const int K = s_K;
    for (int i = tid; i < N; i += 256) {
        w.dwork[lo + i] = sd[i];
        w.zwork[lo + i] = sz[i];
    }
    for (int t = tid; t < K; t += 256) {
        const int idx = w.ndidx[lo + t];
        w.dk[lo + t] = sd[idx];
        w.zk[lo + t] = sz[idx];
    }
    // deflated indices, in local index order (final positions come from the rank sort)
    if (tid == 0) {
        int q = 0;
        for (int i = 0; i < N; ++i)
            if (sdefl[i]) w.deflidx[lo + q++] = i;
    }
}

// one thread per row of the merge block: apply the recorded Givens chain to Qcur in place, then compact the
// non-deflated columns into Q2w[:, lo : lo+K)
__global__ __launch_bounds__(256) void dc_rotate_compact_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.y];
    const int lo = sg.lo, hi = sg.hi, n = w.n;
    const int r = lo + blockIdx.x * 256 + threadIdx.x;
    if (r >= hi) return;
    const int K = w.meta[2 * blockIdx.y], nrot = w.meta[2 * blockIdx.y + 1];
    double *q = w.Qcur + (long)r * n + lo;
    for (int t = 0; t < nrot; ++t) {
        const int a = w.rota[lo + t], bb = w.rotb[lo + t];
        const double c_ = w.rotc[lo + t], s_ = w.rots[lo + t];
        const double qa = q[a], qb = q[bb];
        q[a] = c_ * qa + s_ * qb;
        q[bb] = -s_ * qa + c_ * qb;
    }
    double *q2 = w.Q2w + (long)r * n + lo;
    for (int t = 0; t < K; ++t) q2[t] = q[w.ndidx[lo + t]];
}

// one wave per root of the secular equation 1 + rho sum_j zk_j^2 / (dk_j - lam) = 0
__global__ __launch_bounds__(256) void dc_secular_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.y];
    const int lo = sg.lo;
    const int K = w.meta[2 * blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= K) return;
    const double rho = 2.0 * fabs(w.e[sg.mid - 1]);
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

// one wave per pole i: zhat_i = sign(z_i) sqrt( prod_j (lam_j - d_i) / (rho prod_{j != i} (d_j - d_i)) )
__global__ __launch_bounds__(256) void dc_zhat_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.y];
    const int lo = sg.lo;
    const int K = w.meta[2 * blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= K) return;
    const double rho = 2.0 * fabs(w.e[sg.mid - 1]);
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

// one wave per root j: 1 / || zhat_i / (d_i - lam_j) ||_2
__global__ __launch_bounds__(256) void dc_colnorm_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.y];
    const int lo = sg.lo;
    const int K = w.meta[2 * blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= K) return;
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

// U[i][j] = zhat_i / (d_i - lam_j) * invn_j  into the diagonal block of Uw (coalesced along j)
__global__ __launch_bounds__(256) void dc_build_U_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.z];
    const int lo = sg.lo, n = w.n;
    const int K = w.meta[2 * blockIdx.z];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (j >= K) return;
    const double *__restrict__ dk = w.dk + lo;
    const double dorg = dk[w.org[lo + j]], muj = w.mu[lo + j], inj = w.invn[lo + j];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + q;
        if (i < K) w.Uw[(long)(lo + i) * n + lo + j] = w.zhat[lo + i] / ((dk[i] - dorg) - muj) * inj;
    }
}

// one workgroup per merge: rank-sort {roots} U {deflated d}; dnext and Qnext columns in ascending order
__global__ __launch_bounds__(256) void dc_finalize_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.x];
    const int lo = sg.lo, hi = sg.hi, n = w.n;
    const int N = hi - lo;
    const int K = w.meta[2 * blockIdx.x];
    __shared__ double val[EIG_MAXN];
    __shared__ int rank[EIG_MAXN], src[EIG_MAXN];
    const int tid = threadIdx.x;
    for (int t = tid; t < N; t += 256) {
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
    for (int t = tid; t < N; t += 256) {
        const double v = val[t];
        int rk = 0;
        for (int u = 0; u < N; ++u) {
            const double x = val[u];
            rk += (x < v) || (x == v && u < t);
        }
        w.dnext[lo + rk] = v;
        w.rota[lo + t] = rk;          // rotation list is consumed by now: reuse as rank / source tables
        w.rotb[lo + t] = src[t];
    }
    (void)rank;
    (void)n;
    (void)hi;
}

// grid (row tiles, merges): Qnext[:, rank[t]] = (root ? Ww[:, t] : Qcur[:, deflated idx]); reads coalesced along t
__global__ __launch_bounds__(256) void dc_place_kernel(DcWork w, const Seg *__restrict__ segs) {
    const Seg sg = segs[blockIdx.y];
    const int lo = sg.lo, hi = sg.hi, n = w.n;
    const int N = hi - lo;
    const int r = lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= hi) return;
    for (int t = threadIdx.x & 63; t < N; t += 64) {
        const int sc = w.rotb[lo + t];
        const double v = (sc < 0) ? w.Ww[(long)r * n + lo + (-1 - sc)] : w.Qcur[(long)r * n + lo + sc];
        w.Qnext[(long)r * n + lo + w.rota[lo + t]] = v;
    }
}

__global__ void copy_vec_kernel(const double *__restrict__ a, double *__restrict__ b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

struct DcPlan {             // host-side partition of [0, n): leaves and merge levels (cached per n)
    int n = 0;
    std::vector<int> leaf_lo;                 // leaves + 1 entries
    std::vector<int> bounds;                  // interior leaf boundaries (tears)
    std::vector<std::vector<Seg>> levels;     // bottom-up
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
    return p;
}

void stedc_device(gpcsd_ctx *c, const double *d, const double *e, int n, double *wout, double *Zout, int *d_status,
                  hipStream_t s, const char *tag) {
    GP_REQUIRE(n >= 1 && n <= EIG_MAXN, -3, "stedc: n=%d outside [1,%d]", n, EIG_MAXN);
    const std::string T = std::string("dc_") + (tag ? tag : "") + "_";
    const DcPlan plan = make_plan(n);
    const size_t nn = (size_t)n * n;
    DcWork w;
    w.n = n;
    w.e = e;
    w.dcur = c->buf<double>(T + "dcur", n);
    w.dnext = c->buf<double>(T + "dnext", n);
    w.Qcur = c->buf<double>(T + "Qcur", nn);
    w.Qnext = c->buf<double>(T + "Qnext", nn);
    w.dwork = c->buf<double>(T + "dwork", n);
    w.zwork = c->buf<double>(T + "zwork", n);
    w.dk = c->buf<double>(T + "dk", n);
    w.zk = c->buf<double>(T + "zk", n);
    w.mu = c->buf<double>(T + "mu", n);
    w.lam = c->buf<double>(T + "lam", n);
    w.zhat = c->buf<double>(T + "zhat", n);
    w.invn = c->buf<double>(T + "invn", n);
    w.org = c->buf<int>(T + "org", n);
    w.ndidx = c->buf<int>(T + "ndidx", n);
    w.deflidx = c->buf<int>(T + "deflidx", n);
    w.rota = c->buf<int>(T + "rota", n);
    w.rotb = c->buf<int>(T + "rotb", n);
    w.rotc = c->buf<double>(T + "rotc", n);
    w.rots = c->buf<double>(T + "rots", n);
    w.meta = c->buf<int>(T + "meta", 2 * n + 2);
    w.Kdyn = c->buf<int>(T + "Kdyn", n + 2);
    w.Q2w = c->buf<double>(T + "Q2w", nn);
    w.Uw = c->buf<double>(T + "Uw", nn);
    w.Ww = c->buf<double>(T + "Ww", nn);

    // plan tables -> device (tiny; cached buffers, re-uploaded per call: the content depends only on n)
    std::vector<int> tbl;
    tbl.insert(tbl.end(), plan.leaf_lo.begin(), plan.leaf_lo.end());
    const int off_bounds = (int)tbl.size();
    tbl.insert(tbl.end(), plan.bounds.begin(), plan.bounds.end());
    std::vector<int> lvl_off;
    for (auto &lv : plan.levels) {
        lvl_off.push_back((int)tbl.size());
        for (auto &sg : lv) {
            tbl.push_back(sg.lo);
            tbl.push_back(sg.mid);
            tbl.push_back(sg.hi);
        }
    }
    int *dtbl = c->buf<int>(T + "plan", tbl.size() + 4);
    int &cached_n = c->int_cache[T + "plan_n"];
    if (cached_n != n) {                  // the table depends only on n: upload once per (tag, n)
        GP_HIP(hipMemcpyAsync(dtbl, tbl.data(), tbl.size() * sizeof(int), hipMemcpyHostToDevice, s));
        GP_HIP(hipStreamSynchronize(s));  // tbl is a stack temporary
        cached_n = n;
    }

    const int nleaves = (int)plan.leaf_lo.size() - 1;
    // both ping-pong matrices start at zero: a merge reads the off-diagonal blocks between its two halves, which no
    // earlier level writes (blocks are nested, so zeros outside the blocks of a level survive until they are merged)
    GP_HIP(hipMemsetAsync(w.Qcur, 0, nn * sizeof(double), s));
    GP_HIP(hipMemsetAsync(w.Qnext, 0, nn * sizeof(double), s));
    hipLaunchKernelGGL(dc_tear_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, d, e, n, (const int *)(dtbl + off_bounds),
                       (int)plan.bounds.size(), w.dwork);
    hipLaunchKernelGGL(dc_leaf_kernel, dim3(nleaves), dim3(256), 0, s, (const double *)w.dwork, e, n, (const int *)dtbl, w.dcur,
                       w.Qcur, d_status);
    GP_HIP(hipGetLastError());

    for (size_t L = 0; L < plan.levels.size(); ++L) {
        const auto &lv = plan.levels[L];
        const int nm = (int)lv.size();
        const Seg *dsegs = reinterpret_cast<const Seg *>(dtbl + lvl_off[L]);
        int maxN = 0;
        for (auto &sg : lv) maxN = std::max(maxN, sg.hi - sg.lo);
        hipLaunchKernelGGL(dc_merge_setup_kernel, dim3(nm), dim3(256), 0, s, w, dsegs);
        hipLaunchKernelGGL(dc_rotate_compact_kernel, dim3(ceil_div(maxN, 256), nm), dim3(256), 0, s, w, dsegs);
        hipLaunchKernelGGL(dc_secular_kernel, dim3(ceil_div(maxN, 4), nm), dim3(256), 0, s, w, dsegs);
        hipLaunchKernelGGL(dc_zhat_kernel, dim3(ceil_div(maxN, 4), nm), dim3(256), 0, s, w, dsegs);
        hipLaunchKernelGGL(dc_colnorm_kernel, dim3(ceil_div(maxN, 4), nm), dim3(256), 0, s, w, dsegs);
        hipLaunchKernelGGL(dc_build_U_kernel, dim3(ceil_div(maxN, 64), ceil_div(maxN, 16), nm), dim3(256), 0, s, w, dsegs);
        GP_HIP(hipGetLastError());
        for (int m = 0; m < nm; ++m) {                    // W = Q2 (N x K) U (K x K), K read on the device
            const Seg &sg = lv[m];
            const int N = sg.hi - sg.lo;
            GemmDesc g;
            g.M = N; g.N = N; g.K = N;
            g.A = w.Q2w + (long)sg.lo * n + sg.lo; g.lda = n;
            g.B = w.Uw + (long)sg.lo * n + sg.lo; g.ldb = n;
            g.C = w.Ww + (long)sg.lo * n + sg.lo; g.ldc = n;
            g.dyn = w.Kdyn + m;
            g.prof_name = "gemm_dc_merge";
            gemm_f64(c, g, s);
        }
        hipLaunchKernelGGL(dc_finalize_kernel, dim3(nm), dim3(256), 0, s, w, dsegs);
        hipLaunchKernelGGL(dc_place_kernel, dim3(ceil_div(maxN, 4), nm), dim3(256), 0, s, w, dsegs);
        GP_HIP(hipGetLastError());
        std::swap(w.dcur, w.dnext);
        std::swap(w.Qcur, w.Qnext);
    }
    hipLaunchKernelGGL(copy_vec_kernel, dim3(1), dim3(256), 0, s, (const double *)w.dcur, wout, (long)n);
    hipLaunchKernelGGL(copy_vec_kernel, dim3(256), dim3(256), 0, s, (const double *)w.Qcur, Zout, (long)nn);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// stage 3: back-transformation with compact-WY panels
// ------------------------------------------------------------------------------------------------------------------
// one workgroup per panel: T upper triangular, T[j][j] = tau_j, T[0:j, j] = -tau_j T[0:j,0:j] G[0:j, j]
__global__ __launch_bounds__(256) void wy_T_kernel(const double *__restrict__ G, const double *__restrict__ tau, int nrefl,
                                                   double *__restrict__ Tout) {
    const int p = blockIdx.x;
    __shared__ double T[WY_NB][WY_NB + 1];
    __shared__ double g[WY_NB][WY_NB + 1];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < WY_NB * WY_NB; idx += 256) {
        const int i = idx / WY_NB, j = idx % WY_NB;
        g[i][j] = G[(long)p * WY_NB * WY_NB + idx];
        T[i][j] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < WY_NB; ++j) {
        const int kk = p * WY_NB + j;
        const double tj = (kk < nrefl) ? tau[kk] : 0.0;
        if (tid < j) {
            double s = 0.0;
            for (int l = tid; l < j; ++l) s += T[tid][l] * g[l][j];      // T upper triangular: l >= row
            T[tid][j] = -tj * s;
        }
        if (tid == j) T[j][j] = tj;
        __syncthreads();
    }
    for (int idx = tid; idx < WY_NB * WY_NB; idx += 256) Tout[(long)p * WY_NB * WY_NB + idx] = T[idx / WY_NB][idx % WY_NB];
}

static void ormtr_device(gpcsd_ctx *c, const double *V, const double *tau, int n, double *Z, hipStream_t s,
                         const std::string &T) {
    const int nrefl = n - 2;
    if (nrefl <= 0) return;
    const int P = ceil_div(nrefl, WY_NB);
    double *G = c->buf<double>(T + "wyG", (size_t)P * WY_NB * WY_NB);
    double *Tm = c->buf<double>(T + "wyT", (size_t)P * WY_NB * WY_NB);
    double *VT = c->buf<double>(T + "wyVT", (size_t)P * n * WY_NB);
    double *W1 = c->buf<double>(T + "wyW1", (size_t)WY_NB * n);
    GemmDesc gg;                                  // G_p = V_p V_p^T
    gg.M = WY_NB; gg.N = WY_NB; gg.K = n;
    gg.A = V; gg.lda = n; gg.B = V; gg.ldb = n; gg.transB = true;
    gg.C = G; gg.ldc = WY_NB;
    gg.batch = P; gg.sA = (long)WY_NB * n; gg.sB = (long)WY_NB * n; gg.sC = (long)WY_NB * WY_NB;
    gg.prof_name = "gemm_wy_gram";
    gemm_f64(c, gg, s);
    hipLaunchKernelGGL(wy_T_kernel, dim3(P), dim3(256), 0, s, (const double *)G, tau, nrefl, Tm);
    GP_HIP(hipGetLastError());
    GemmDesc gv;                                  // VT_p = V_p^T T_p   (n x nb)
    gv.M = n; gv.N = WY_NB; gv.K = WY_NB;
    gv.A = V; gv.lda = n; gv.transA = true; gv.B = Tm; gv.ldb = WY_NB;
    gv.C = VT; gv.ldc = WY_NB;
    gv.batch = P; gv.sA = (long)WY_NB * n; gv.sB = (long)WY_NB * WY_NB; gv.sC = (long)n * WY_NB;
    gv.prof_name = "gemm_wy_vt";
    gemm_f64(c, gv, s);
    for (int p = P - 1; p >= 0; --p) {
        GemmDesc g1;                              // W1 = V_p Z  (nb x n)
        g1.M = WY_NB; g1.N = n; g1.K = n;
        g1.A = V + (long)p * WY_NB * n; g1.lda = n; g1.B = Z; g1.ldb = n; g1.C = W1; g1.ldc = n;
        g1.prof_name = "gemm_wy_vz";
        gemm_f64(c, g1, s);
        GemmDesc g2;                              // Z -= VT_p W1
        g2.M = n; g2.N = n; g2.K = WY_NB;
        g2.A = VT + (long)p * n * WY_NB; g2.lda = WY_NB; g2.B = W1; g2.ldb = n; g2.C = Z; g2.ldc = n;
        g2.alpha = -1.0; g2.epi = EPI_ACCUM;
        g2.prof_name = "gemm_wy_update";
        gemm_f64(c, g2, s);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------------------------
struct EigProb {
    double *A, *w, *Z;
    int n;
    std::string tag;
    double *amax;
    SytrdProb sp;
};

static void prep_problem(gpcsd_ctx *c, EigProb &p, hipStream_t s) {
    const int n = p.n;
    const size_t nn = (size_t)n * n;
    const std::string T = "eig_" + p.tag + "_";
    p.amax = c->buf<double>(T + "amax", 2);
    p.sp.n = n;
    p.sp.A0 = c->buf<double>(T + "A0", nn);
    p.sp.A1 = c->buf<double>(T + "A1", nn);
    p.sp.V = c->buf<double>(T + "V", (size_t)(n + WY_NB) * n);
    p.sp.tau = c->buf<double>(T + "tau", n + WY_NB);
    p.sp.d = c->buf<double>(T + "d", n);
    p.sp.e = c->buf<double>(T + "e", n);
    p.sp.y0 = c->buf<double>(T + "y0", n);
    p.sp.y1 = c->buf<double>(T + "y1", n);
    hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, s, (const double *)p.A, (long)nn, p.amax);
    hipLaunchKernelGGL(scale_copy_kernel, dim3(256), dim3(256), 0, s, (const double *)p.A, (long)nn, (const double *)p.amax,
                       p.sp.A0);
    GP_HIP(hipMemsetAsync(p.sp.V, 0, (size_t)(n + WY_NB) * n * sizeof(double), s));
    GP_HIP(hipMemsetAsync(p.sp.tau, 0, (size_t)(n + WY_NB) * sizeof(double), s));
    GP_HIP(hipGetLastError());
}

void sytrd_device(gpcsd_ctx *c, double *A, int n, double *d, double *e, double *V, double *tau, hipStream_t s) {
    GP_REQUIRE(n >= 1 && n <= EIG_MAXN, -3, "sytrd: n=%d outside [1,%d]", n, EIG_MAXN);
    EigProb p;
    p.A = A; p.n = n; p.tag = "dbg"; p.w = nullptr; p.Z = nullptr;
    prep_problem(c, p, s);
    SytrdBatch b;
    b.p[0] = p.sp;
    sytrd_batch_launch(c, b, 1, n, s);
    GP_HIP(hipMemcpyAsync(d, p.sp.d, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(e, p.sp.e, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(tau, p.sp.tau, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(V, p.sp.V, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, s));
    // d, e are those of A / max|A|; rescale so the caller sees the tridiagonal of A itself
    hipLaunchKernelGGL(scale_vec_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, d, n, (const double *)p.amax);
    hipLaunchKernelGGL(scale_vec_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, e, n, (const double *)p.amax);
    GP_HIP(hipGetLastError());
}

void eigh_large_batch(gpcsd_ctx *c, EigProb *probs, int count, int *d_status, hipStream_t s) {
    GP_REQUIRE(count >= 1 && count <= MAX_BATCH, -3, "eigh batch size %d outside [1,%d]", count, MAX_BATCH);
    SytrdBatch b;
    int nmax = 0;
    for (int i = 0; i < count; ++i) {
        GP_REQUIRE(probs[i].n > 2 && probs[i].n <= EIG_MAXN, -3, "eigh(large): n=%d outside (2,%d]", probs[i].n, EIG_MAXN);
        prep_problem(c, probs[i], s);
        b.p[i] = probs[i].sp;
        nmax = std::max(nmax, probs[i].n);
    }
    {
        ProfScope ps(c, "eigh_sytrd", 0.0, s);
        sytrd_batch_launch(c, b, count, nmax, s);
    }
    for (int i = 0; i < count; ++i) {
        EigProb &p = probs[i];
        {
            ProfScope ps(c, "eigh_stedc", 0.0, s);
            stedc_device(c, p.sp.d, p.sp.e, p.n, p.w, p.Z, d_status, s, p.tag.c_str());
        }
        {
            ProfScope ps(c, "eigh_backtransform", 0.0, s);
            ormtr_device(c, p.sp.V, p.sp.tau, p.n, p.Z, s, "eig_" + p.tag + "_");
        }
        hipLaunchKernelGGL(scale_vec_kernel, dim3(ceil_div(p.n, 256)), dim3(256), 0, s, p.w, p.n, (const double *)p.amax);
    }
    GP_HIP(hipGetLastError());
}

void eigh_large_pair(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, const char *tag0, double *A1, int n1, double *w1,
                     double *Z1, const char *tag1, int *d_status, hipStream_t s) {
    EigProb probs[2];
    int count = 0;
    if (A0 && n0 > 0) {
        probs[count].A = A0; probs[count].n = n0; probs[count].w = w0; probs[count].Z = Z0; probs[count].tag = tag0;
        ++count;
    }
    if (A1 && n1 > 0) {
        probs[count].A = A1; probs[count].n = n1; probs[count].w = w1; probs[count].Z = Z1; probs[count].tag = tag1;
        ++count;
    }
    if (count) eigh_large_batch(c, probs, count, d_status, s);
}

}  // namespace gpcsd
