// Blocked right-looking Cholesky: numpy.linalg.cholesky at gpcsd1d.py:303-304 and
// gpcsd2d.py:343-350 (sample_prior), plus the dense-K cross-check path (potrf + log-det + trsm).
//
// Per 64-column block step:
//   1. one workgroup factors the diagonal block in LDS and inverts the 64x64 triangle (trtri) in LDS,
//   2. the panel solve L21 = A21 L11^{-T} is an fp64 MFMA GEMM against the inverted triangle,
//   3. the trailing update A22 -= L21 L21^T is the fp64 MFMA GEMM (rank-64 update, EPI_ACCUM, alpha = -1).
// Triangular solves with many right-hand sides use the same inverted-diagonal-block + GEMM scheme.
#include "kernels.hpp"

namespace gpcsd {

constexpr int NB = 64;

// Inverse of the lower-triangular nb x nb block L (LDS) into X (LDS): thread j owns column j.
__device__ void trtri_block_lds(double (*L)[NB + 1], double (*X)[NB + 1], int nb) {
    const int j = threadIdx.x;
    if (j < nb) {
        for (int i = 0; i < j; ++i) X[i][j] = 0.0;
        X[j][j] = 1.0 / L[j][j];
        for (int i = j + 1; i < nb; ++i) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v += L[i][k] * X[k][j];
            X[i][j] = -v / L[i][i];
        }
    }
    __syncthreads();
}

// One workgroup: Cholesky of the diagonal block in LDS, write L11 (upper zeroed) back, write inv(L11) to Linv.
__global__ __launch_bounds__(256) void potrf_diag_kernel(double *A, int n, int k0, int nb, double *Linv, int *status) {
    __shared__ double L[NB][NB + 1];
    __shared__ double X[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        L[i][j] = A[(long)(k0 + i) * n + (k0 + j)];
    }
    for (int j = 0; j < nb; ++j) {
        __syncthreads();
        const double d = L[j][j];
        if (tid == 0 && !(d > 0.0)) atomicCAS(status, 0, k0 + j + 1);
        const double ljj = sqrt(d);
        __syncthreads();
        if (tid == 0) L[j][j] = ljj;
        for (int i = j + 1 + tid; i < nb; i += 256) L[i][j] = L[i][j] / ljj;
        __syncthreads();
        const int rem = nb - (j + 1);
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = j + 1 + e / rem, c = j + 1 + e % rem;
            if (c <= i) L[i][c] -= L[i][j] * L[c][j];
        }
    }
    __syncthreads();
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        A[(long)(k0 + i) * n + (k0 + j)] = (j <= i) ? L[i][j] : 0.0;
    }
    trtri_block_lds(L, X, nb);
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        Linv[e] = (i < nb && j < nb) ? X[i][j] : 0.0;
    }
}

// inv of an already-factored diagonal block (for triangular solves)
__global__ __launch_bounds__(256) void trtri_diag_kernel(const double *__restrict__ Lm, int n, int k0, int nb, double *Linv) {
    __shared__ double L[NB][NB + 1];
    __shared__ double X[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        L[i][j] = Lm[(long)(k0 + i) * n + (k0 + j)];
    }
    __syncthreads();
    trtri_block_lds(L, X, nb);
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        Linv[e] = (i < nb && j < nb) ? X[i][j] : 0.0;
    }
}

__global__ void copy2d_kernel(const double *__restrict__ src, long lds_, double *__restrict__ dst, long ldd, int rows, int cols) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < (long)rows * cols) {
        const int i = (int)(e / cols), j = (int)(e % cols);
        dst[(long)i * ldd + j] = src[(long)i * lds_ + j];
    }
}

__global__ void zero_upper_kernel(double *A, int n) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < (long)n * n) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (j > i) A[e] = 0.0;
    }
}

void potrf_device(gpcsd_ctx *c, double *A, int n, int *d_status, hipStream_t s) {
    ProfScope ps(c, "potrf", (double)n * n * n / 3.0, s);
    double *Linv = c->buf<double>("chol_Linv", NB * NB);
    double *W = c->buf<double>("chol_panel", (size_t)n * NB);
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = (n - k0 < NB) ? (n - k0) : NB;
        const int rows = n - (k0 + nb);
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, s, A, n, k0, nb, Linv, d_status);
        if (rows > 0) {
            GemmDesc p;                                   // W = A21 * inv(L11)^T
            p.M = rows; p.N = nb; p.K = nb;
            p.A = A + (long)(k0 + nb) * n + k0; p.lda = n; p.transA = false;
            p.B = Linv; p.ldb = NB; p.transB = true;
            p.C = W; p.ldc = NB;
            p.prof_name = "potrf_panel";
            gemm_f64(c, p, s);
            hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)rows * nb, 256)), dim3(256), 0, s, (const double *)W,
                               (long)NB, A + (long)(k0 + nb) * n + k0, (long)n, rows, nb);
            GemmDesc g;                                   // A22 -= L21 L21^T
            g.M = rows; g.N = rows; g.K = nb;
            g.A = W; g.lda = NB; g.transA = false;
            g.B = W; g.ldb = NB; g.transB = true;
            g.C = A + (long)(k0 + nb) * n + (k0 + nb); g.ldc = n;
            g.alpha = -1.0; g.epi = EPI_ACCUM;
            g.prof_name = "potrf_syrk";
            gemm_f64(c, g, s);
        }
    }
    hipLaunchKernelGGL(zero_upper_kernel, dim3(ceil_div((long)n * n, 256)), dim3(256), 0, s, A, n);
    GP_HIP(hipGetLastError());
}

void trsm_lower_device(gpcsd_ctx *c, const double *L, int n, double *B, int nrhs, hipStream_t s) {
    ProfScope ps(c, "trsm", (double)n * n * nrhs, s);
    double *Linv = c->buf<double>("chol_Linv", NB * NB);
    double *Xb = c->buf<double>("trsm_xblk", (size_t)NB * nrhs);
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = (n - k0 < NB) ? (n - k0) : NB;
        hipLaunchKernelGGL(trtri_diag_kernel, dim3(1), dim3(256), 0, s, L, n, k0, nb, Linv);
        GemmDesc d;                                       // X_blk = inv(L11) B_blk
        d.M = nb; d.N = nrhs; d.K = nb;
        d.A = Linv; d.lda = NB; d.transA = false;
        d.B = B + (long)k0 * nrhs; d.ldb = nrhs; d.transB = false;
        d.C = Xb; d.ldc = nrhs;
        d.prof_name = "trsm_gemm";
        gemm_f64(c, d, s);
        hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)nb * nrhs, 256)), dim3(256), 0, s, (const double *)Xb,
                           (long)nrhs, B + (long)k0 * nrhs, (long)nrhs, nb, nrhs);
        const int rows = n - (k0 + nb);
        if (rows > 0) {
            GemmDesc g;                                   // B[below] -= L21 X_blk
            g.M = rows; g.N = nrhs; g.K = nb;
            g.A = L + (long)(k0 + nb) * n + k0; g.lda = n; g.transA = false;
            g.B = Xb; g.ldb = nrhs; g.transB = false;
            g.C = B + (long)(k0 + nb) * nrhs; g.ldc = nrhs;
            g.alpha = -1.0; g.epi = EPI_ACCUM;
            g.prof_name = "trsm_gemm";
            gemm_f64(c, g, s);
        }
    }
    GP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void logdet_chol_kernel(const double *__restrict__ L, int n, double *out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += log(L[(long)i * n + i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = 2.0 * sh[0];
}

void logdet_chol_device(gpcsd_ctx *c, const double *L, int n, double *out, hipStream_t s) {
    hipLaunchKernelGGL(logdet_chol_kernel, dim3(1), dim3(256), 0, s, L, n, out);
    GP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const double *__restrict__ x, long n, double *partials) {
    __shared__ double sh[256];
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ p, int n, double *out) {
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

void sumsq_device(gpcsd_ctx *c, const double *x, long n, double *out, hipStream_t s) {
    int blocks = (int)((n + 2047) / 2048);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    double *part = c->buf<double>("sumsq_partials", 256);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, x, n, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double *)part, blocks, out);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
