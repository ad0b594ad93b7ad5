// Two-sided cyclic Jacobi eigensolver body (device template), shared by eigh.hip (small dense problems) and
// eigh_dc.hip (leaves of the divide-and-conquer tridiagonal solver).
#pragma once
#include <hip/hip_runtime.h>

namespace gpcsd {

constexpr int JACOBI_LDS_MAX = 64;
constexpr int JACOBI_MAX_N = 1024;
constexpr int JACOBI_MAX_SWEEPS = 60;

// reciprocal and reciprocal square root to ~2 ulp: hardware seed + two Newton steps
__device__ __forceinline__ double jac_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double jac_rsqrt(double x) {        // x in [1, 1e300]
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}

// pair k of round-robin step `step` over m (even) players; returns p < q
__device__ __forceinline__ void rr_pair(int k, int step, int m, int &p, int &q) {
    int a, b;
    if (k == 0) {
        a = m - 1;
        b = step % (m - 1);
    } else {
        a = (step + k) % (m - 1);
        b = (step - k + (m - 1)) % (m - 1);
    }
    p = a < b ? a : b;
    q = a < b ? b : a;
}

template <int NT>
__device__ void jacobi_body(double *A, int lda, double *V, int ldv, int n, double *evals, double *evecs, long ldz,
                            int *status, double *cs, int *pq, double *red) {
    const int tid = threadIdx.x;
    const int m = (n + 1) & ~1;
    const int npairs = m / 2;
    __shared__ int s_rot;
    __shared__ double s_thresh;

    // Scale to max|a| = 1 as LAPACK does, so the squares in the norm and the rotation formulas neither overflow nor vanish
    // (1e290 * G and 1e-300 * G used to come back wrong / NaN); non-finite entries are zeroed and reported as a failure.
    __shared__ double s_scale;
    {
        double mx = 0.0;
        int bad = 0;
        for (int e = tid; e < n * n; e += NT) {
            const double a = fabs(A[(e / n) * lda + (e % n)]);
            if (a <= 1.7e308) mx = fmax(mx, a);
            else bad = 1;
        }
        red[tid] = mx;
        bad = __syncthreads_or(bad);
        for (int w = NT / 2; w > 0; w >>= 1) {
            if (tid < w) red[tid] = fmax(red[tid], red[tid + w]);
            __syncthreads();
        }
        if (tid == 0) {
            s_scale = red[0] > 0.0 ? red[0] : 1.0;
            if (bad) atomicMax(status, 4);
        }
        __syncthreads();
        const double sc = s_scale;
        for (int e = tid; e < n * n; e += NT) {
            const int idx = (e / n) * lda + (e % n);
            const double a = A[idx];
            A[idx] = (fabs(a) <= 1.7e308) ? a / sc : 0.0;
        }
        __syncthreads();
    }
    // V = I, Frobenius norm
    double acc = 0.0;
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e % n;
        V[i * ldv + j] = (i == j) ? 1.0 : 0.0;
        const double a = A[i * lda + j];
        acc += a * a;
    }
    red[tid] = acc;
    __syncthreads();
    for (int w = NT / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) s_thresh = sqrt(red[0]) * 1.1102230246251565e-16 / (double)n;
    __syncthreads();
    const double thresh = s_thresh;

    int sweep = 0;
    bool converged = (n <= 1);
    for (; sweep < JACOBI_MAX_SWEEPS && !converged; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int step = 0; step < m - 1; ++step) {
            // (a) rotation parameters for the disjoint pairs of this step
            for (int k = tid; k < npairs; k += NT) {
                int p, q;
                rr_pair(k, step, m, p, q);
                double c = 1.0, s = 0.0;
                int active = 0;
                if (q < n) {
                    const double apq = A[p * lda + q];
                    if (fabs(apq) > thresh) {
                        // The rotation angle only steers convergence; what must hold to rounding is c^2 + s^2 = 1, and
                        // s = t c with c = rsqrt(1 + t^2) gives that for any t.  So the reciprocal / root chain uses the
                        // ~2 ulp Newton forms (15 instructions) instead of three IEEE sequences (~110): this section is
                        // a serial dependency of every Jacobi step.
                        const double app = A[p * lda + p], aqq = A[q * lda + q];
                        const double theta = (aqq - app) * jac_rcp(2.0 * apq);
                        double t;
                        if (fabs(theta) > 1e150) t = 0.5 * jac_rcp(theta);
                        else {
                            const double th2 = theta * theta + 1.0;
                            t = copysign(1.0, theta) * jac_rcp(fabs(theta) + th2 * jac_rsqrt(th2));
                        }
                        c = jac_rsqrt(t * t + 1.0);
                        s = t * c;
                        active = 1;
                    }
                }
                cs[2 * k] = c;
                cs[2 * k + 1] = s;
                pq[2 * k] = active ? p : -1;
                pq[2 * k + 1] = q;
                if (active) s_rot = 1;
            }
            __syncthreads();
            // (b) column phase: A <- A J, V <- V J.  Consecutive threads take consecutive pairs of ONE row.
            for (int e = tid; e < npairs * n; e += NT) {
                const int k = e % npairs, i = e / npairs;
                const int p = pq[2 * k];
                if (p < 0) continue;
                const int q = pq[2 * k + 1];
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double aip = A[i * lda + p], aiq = A[i * lda + q];
                A[i * lda + p] = c * aip - s * aiq;
                A[i * lda + q] = s * aip + c * aiq;
                const double vip = V[i * ldv + p], viq = V[i * ldv + q];
                V[i * ldv + p] = c * vip - s * viq;
                V[i * ldv + q] = s * vip + c * viq;
            }
            __syncthreads();
            // (c) row phase: A <- J^T A.  Consecutive threads take consecutive columns of one row pair.
            for (int e = tid; e < npairs * n; e += NT) {
                const int j = e % n, k = e / n;
                const int p = pq[2 * k];
                if (p < 0) continue;
                const int q = pq[2 * k + 1];
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double apj = A[p * lda + j], aqj = A[q * lda + j];
                double np_ = c * apj - s * aqj, nq_ = s * apj + c * aqj;
                if (j == q) np_ = 0.0;      // the annihilated element, exactly
                if (j == p) nq_ = 0.0;
                A[p * lda + j] = np_;
                A[q * lda + j] = nq_;
            }
            __syncthreads();
        }
        converged = (s_rot == 0);
        __syncthreads();
    }

    // sort ascending (rank by counting; ties by index) and scatter eigenpairs
    for (int i = tid; i < n; i += NT) red[i] = A[i * lda + i];
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const double di = red[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double dj = red[j];
            rank += (dj < di) || (!(di < dj) && j < i);      // a valid permutation even for unordered (NaN) values
        }
        pq[i] = rank;
        evals[rank] = di * s_scale;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e % n;
        evecs[(long)i * ldz + pq[j]] = V[i * ldv + j];
    }
    if (tid == 0 && !converged) atomicMax(status, 1);
}


}  // namespace gpcsd
