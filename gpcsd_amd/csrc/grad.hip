// Analytic gradient of the log marginal likelihood: the contractions that replace the
// reference's autograd tape (gpcsd1d.py:211, gpcsd2d.py:250).  With B_r = alpha_r / D (eigenbasis) and scalar sig2n:
//   Ghat_s = 1/2 sum_r B_r diag(et) B_r^T - R/2 diag(a),  a_x = sum_i et_i / D_xi,   Gs = Qs Ghat_s Qs^T
//   Ghat_t = 1/2 sum_r B_r^T diag(es) B_r - R/2 diag(b),  b_i = sum_x es_x / D_xi,   Gt = Qt Ghat_t Qt^T
//   dL/dtheta_t = <Gt, dKt/dtheta_t>,  dL/dell_s = <A^T Gs A, dKgl/dell_s>,  dL/dR = 2 <Gs (A Kgl), dA/dR>,
//   dL/dsig2n  = -R/2 sum 1/D + 1/2 sum B^2.
// The GEMMs run on the fp64 MFMA core (capi.hip wires them); this file holds the elementwise derivative kernels and
// the deterministic reductions (per-block partials in fixed order, then one block per output value).
#include "devutil.hpp"
#include "kernels.hpp"

namespace gpcsd {

constexpr int GR_MAXV = 2 * GPCSD_MAX_TEMPORAL;

template <int NV>
__device__ __forceinline__ void block_partials(double (&v)[NV], double *partials, int nblocks, int b) {
    __shared__ double red[NV][4];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const double s = wave_sum(v[q]);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) partials[(long)threadIdx.x * nblocks + b] = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
                                                                      (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// out[v] = scale[v-th] * sum_b partials[v*nblocks + b]; one workgroup per value
// blockIdx.y = hyper-parameter set of a batched evaluation (partials pstride apart, outputs ostride apart)
__global__ __launch_bounds__(256) void final_sums_kernel(const double *__restrict__ partials, int nblocks, double *out,
                                                         long pstride = 0, long ostride = 0) {
    __shared__ double red[4];
    partials += blockIdx.y * pstride;
    out += blockIdx.y * ostride;
    const int v = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partials[(long)v * nblocks + b];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) out[v] = s;
}

// a_x = sum_i et_i / D_xi and per-row partial of sum 1/D; one workgroup per eigen-row x
__global__ __launch_bounds__(256) void d_rowsums_kernel(const double *__restrict__ D, const double *__restrict__ et, int nt,
                                                        double *__restrict__ a, double *__restrict__ s1row) {
    __shared__ double red[4];
    {                                           // blockIdx.y = hyper-parameter set (gridDim.x = nx rows each)
        const long b = blockIdx.y, nx = gridDim.x;
        D += b * nx * nt; et += b * nt; a += b * nx; s1row += b * nx;
    }
    const int x = blockIdx.x;
    double sa = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nt; i += 256) {
        const double inv = 1.0 / D[(long)x * nt + i];
        sa += et[i] * inv;
        s1 += inv;
    }
    sa = block_sum256(sa, red);
    s1 = block_sum256(s1, red);
    if (threadIdx.x == 0) {
        a[x] = sa;
        s1row[x] = s1;
    }
}

// b_i = sum_x es_x / D_xi.  A workgroup owns 32 eigen-columns (coalesced across i); eight row groups walk the rows x = g, g + 8, ..
// in order and their partial sums are added in a fixed order.  (One thread per column over all nx rows -- nx dependent divisions
// on two workgroups -- was 104 us of every objective + gradient evaluation at 384 x 500.)
constexpr int DCS_COLS = 32, DCS_GROUPS = 8;
__global__ __launch_bounds__(256) void d_colsums_kernel(const double *__restrict__ D, const double *__restrict__ es, int nx,
                                                        int nt, double *__restrict__ b) {
    __shared__ double part[DCS_GROUPS][DCS_COLS];
    {                                           // blockIdx.y = hyper-parameter set
        const long q = blockIdx.y;
        D += q * nx * nt; es += q * nx; b += q * nt;
    }
    const int col = threadIdx.x % DCS_COLS, g = threadIdx.x / DCS_COLS;
    const int i = blockIdx.x * DCS_COLS + col;
    double s = 0.0;
    if (i < nt)
        for (int x = g; x < nx; x += DCS_GROUPS) s += es[x] / D[(long)x * nt + i];
    part[g][col] = s;
    __syncthreads();
    if (g == 0 && i < nt) {
        double t = part[0][col];
#pragma unroll
        for (int k = 1; k < DCS_GROUPS; ++k) t += part[k][col];
        b[i] = t;
    }
}

void k_D_sums(gpcsd_ctx *c, const double *D, const double *es, const double *et, int nx, int nt, double *a, double *b,
              double *s1_out, hipStream_t s, int B, long s_s1) {
    double *s1row = c->buf<double>("grad_s1row", (size_t)nx * B);
    hipLaunchKernelGGL(d_rowsums_kernel, dim3(nx, B), dim3(256), 0, s, D, et, nt, a, s1row);
    hipLaunchKernelGGL(d_colsums_kernel, dim3(ceil_div(nt, DCS_COLS), B), dim3(256), 0, s, D, es, nx, nt, b);
    hipLaunchKernelGGL(final_sums_kernel, dim3(1, B), dim3(256), 0, s, (const double *)s1row, nx, s1_out, (long)nx, s_s1);
    GP_HIP(hipGetLastError());
}

// out[e] = scale * sum_b in[b*stride + e]  (+ dscale * dvec[i] on the diagonal of the n x n output)
// LPE lanes share one output element: lane g adds the terms b = g, g + LPE, ... in order, the partial sums are combined in a
// fixed xor tree.  With one lane per element a long sum (one term per trial: 200 at cfg5) is 200 dependent loads per
// thread on a handful of workgroups -- 80 us for a 24 x 24 output; eight lanes per element bring it to ~10 us.
template <int LPE>
__global__ __launch_bounds__(256) void batch_reduce_kernel(const double *__restrict__ in, int nb, long stride, int n, double scale,
                                                           const double *__restrict__ dvec, double dscale,
                                                           double *__restrict__ out, long s_in, long s_dvec, long s_out) {
    in += blockIdx.y * s_in;                    // blockIdx.y = hyper-parameter set
    dvec += blockIdx.y * s_dvec;
    out += blockIdx.y * s_out;
    const long e = (blockIdx.x * 256L + threadIdx.x) / LPE;
    const int g = threadIdx.x % LPE;
    const bool ok = e < (long)n * n;
    const int i = ok ? (int)(e / n) : 0, j = ok ? (int)(e % n) : 0;
    // every summand is a symmetric product (B diag(w) B^T), formed on and below the diagonal only (GemmDesc::lower: the tiles above
    // it exit at once): an entry above the diagonal is its mirror image's sum -- half the flops, and the result exactly symmetric
    const long src = j > i ? (long)j * n + i : e;
    double s = 0.0;
    if (ok)
        for (int b = g; b < nb; b += LPE) s += in[b * stride + src];
#pragma unroll
    for (int off = LPE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);   // (LPE lanes of an element sit in one wave)
    if (!ok || g != 0) return;
    s *= scale;
    if (i == j) s += dscale * dvec[i];
    out[e] = s;
}

void k_batch_reduce(gpcsd_ctx *c, const double *in, int nb, long stride, int n, double scale, const double *dvec, double dscale,
                    double *out, hipStream_t s, int B, long s_in, long s_dvec, long s_out) {
    if (s_dvec < 0) s_dvec = n;
    if (s_out < 0) s_out = (long)n * n;
    // the split depends on the sum's length only (never on B): a set is reduced the same way alone or in a batch
    if (nb >= 32)
        hipLaunchKernelGGL(batch_reduce_kernel<8>, dim3(ceil_div((long)n * n * 8, 256), B), dim3(256), 0, s, in, nb, stride, n, scale,
                           dvec, dscale, out, s_in, s_dvec, s_out);
    else
        hipLaunchKernelGGL(batch_reduce_kernel<1>, dim3(ceil_div((long)n * n, 256), B), dim3(256), 0, s, in, nb, stride, n, scale,
                           dvec, dscale, out, s_in, s_dvec, s_out);
    GP_HIP(hipGetLastError());
}

// ---- temporal: sum_ij Gt_ij dKt_ij/d(ell_c), sum_ij Gt_ij dKt_ij/d(sigma2_c)    (covariances.py:269-270, :303-304)
struct TGradParams {
    int ncomp;
    int kind[GPCSD_MAX_TEMPORAL];
    double ell[GPCSD_MAX_TEMPORAL];
    double sigma2[GPCSD_MAX_TEMPORAL];
};

__global__ __launch_bounds__(256) void temporal_grad_kernel(TGradParams p, const double *__restrict__ Gt,
                                                            const double *__restrict__ t, int nt, double *partials,
                                                            const HpDev *__restrict__ tab) {
    if (tab) {                                  // blockIdx.y = hyper-parameter set: Gt nt*nt apart, partials GR_MAXV*1024 apart
        const HpDev &h = tab[blockIdx.y];
        p.ncomp = h.ncomp;
#pragma unroll
        for (int cc = 0; cc < GPCSD_MAX_TEMPORAL; ++cc) {
            p.kind[cc] = h.kind[cc];
            p.ell[cc] = h.ell_t[cc];
            p.sigma2[cc] = h.sigma2_t[cc];
        }
        Gt += (long)blockIdx.y * nt * nt;
        partials += (long)blockIdx.y * GR_MAXV * 1024;
    }
    double v[GR_MAXV];
#pragma unroll
    for (int q = 0; q < GR_MAXV; ++q) v[q] = 0.0;
    const long n2 = (long)nt * nt;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n2; e += (long)gridDim.x * 256) {
        const int i = (int)(e / nt), j = (int)(e % nt);
        const double g = Gt[e];
        const double d = t[i] - t[j];
#pragma unroll
        for (int cc = 0; cc < GPCSD_MAX_TEMPORAL; ++cc) {
            if (cc < p.ncomp) {
                const double ell = p.ell[cc];
                double k, dk;                     // unit-variance kernel and its ell-derivative
                if (p.kind[cc] == GPCSD_KIND_SE) {
                    k = exp(-0.5 * (d * d) / (ell * ell));
                    dk = k * (d * d) / (ell * ell * ell);
                } else {
                    const double ad = fabs(d);
                    k = exp(-ad / ell);
                    dk = k * ad / (ell * ell);
                }
                v[2 * cc] += g * p.sigma2[cc] * dk;
                v[2 * cc + 1] += g * k;
            }
        }
    }
    block_partials<GR_MAXV>(v, partials, gridDim.x, blockIdx.x);
}

void k_temporal_grad(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *Gt, const double *t, int nt, double *out2C,
                     hipStream_t s, const HpDev *tab, int B, long s_out) {
    TGradParams p{};
    p.ncomp = hp->n_temporal;
    for (int i = 0; i < hp->n_temporal; ++i) {
        p.kind[i] = hp->kind[i];
        p.ell[i] = hp->ell_t[i];
        p.sigma2[i] = hp->sigma2_t[i];
    }
    const int nblocks = 256, nb = tab ? B : 1;
    double *part = c->buf<double>("grad_partials_t", (size_t)GR_MAXV * 1024 * nb);      // (own scratch: runs beside the spatial half, capi_grad.inl)
    // the sums land directly in the caller's gradient vector(s): 2 * n_temporal values per set, s_out apart
    hipLaunchKernelGGL(temporal_grad_kernel, dim3(nblocks, nb), dim3(256), 0, s, p, Gt, t, nt, part, tab);
    hipLaunchKernelGGL(final_sums_kernel, dim3(2 * hp->n_temporal, nb), dim3(256), 0, s, (const double *)part, nblocks, out2C,
                       (long)GR_MAXV * 1024, s_out);
    GP_HIP(hipGetLastError());
}

// ---- temporal, user-defined covariances: out[k] = sum_ij Gt_ij dK_k,ij for nm caller-supplied derivative matrices
// (dK_k = d Kt / d theta_k evaluated by the covariance object's own compute_dKt; nm <= 2 * GPCSD_MAX_TEMPORAL)
// out[b][i * R + r] = in[b * s_in + i]: a per-row factor spread over the (row, trial) index of the flat layout, so that a GEMM
// contracting over that index can scale its operand on the way (GemmDesc::kscale) instead of reading a pre-scaled copy
__global__ __launch_bounds__(256) void repeat_rows_kernel(const double *__restrict__ in, long s_in, long nR, int R,
                                                           double *__restrict__ out) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < nR) out[blockIdx.y * nR + i] = in[blockIdx.y * s_in + i / R];
}

void k_repeat_rows(gpcsd_ctx *c, const double *in, long s_in, int n, int R, int B, double *out, hipStream_t s) {
    const long nR = (long)n * R;
    hipLaunchKernelGGL(repeat_rows_kernel, dim3((unsigned)((nR + 255) / 256), B), dim3(256), 0, s, in, s_in, nR, R, out);
    GP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void frob_inner_kernel(const double *__restrict__ Gt, const double *__restrict__ dK, long n2, int nm,
                                                         double *partials) {
    double v[GR_MAXV];
#pragma unroll
    for (int q = 0; q < GR_MAXV; ++q) v[q] = 0.0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n2; e += (long)gridDim.x * 256) {
        const double g = Gt[e];
#pragma unroll
        for (int q = 0; q < GR_MAXV; ++q)
            if (q < nm) v[q] += g * dK[(long)q * n2 + e];
    }
    block_partials<GR_MAXV>(v, partials, gridDim.x, blockIdx.x);
}

void k_frob_inner(gpcsd_ctx *c, const double *Gt, const double *dK, long n2, int nm, double *out, hipStream_t s) {
    GP_REQUIRE(nm >= 1 && nm <= GR_MAXV, -3, "frob_inner: %d matrices (max %d)", nm, GR_MAXV);
    const int nblocks = 256;
    double *part = c->buf<double>("grad_partials_t", (size_t)GR_MAXV * 1024);      // (own scratch: runs beside the spatial half, capi_grad.inl)
    hipLaunchKernelGGL(frob_inner_kernel, dim3(nblocks), dim3(256), 0, s, Gt, dK, n2, nm, part);
    hipLaunchKernelGGL(final_sums_kernel, dim3(nm, 1), dim3(256), 0, s, (const double *)part, nblocks, out, (long)GR_MAXV * 1024, 0L);
    GP_HIP(hipGetLastError());
}

// ---- spatial length scales: sum_gh M_gh Kgl_gh d_gh^2 / ell^3       (covariances.py:89, :216)
__global__ __launch_bounds__(256) void kgl_grad_kernel(const double *__restrict__ M, const double *__restrict__ Kgl,
                                                       const double *__restrict__ gx1, const double *__restrict__ gx2, int G,
                                                       int ngl2, double ell1, double ell2, double *partials,
                                                       const HpDev *__restrict__ tab) {
    if (tab) {                                  // blockIdx.y = hyper-parameter set: M, Kgl G*G apart, partials GR_MAXV*1024 apart
        ell1 = tab[blockIdx.y].ell_s[0];
        ell2 = tab[blockIdx.y].ell_s[1];
        M += (long)blockIdx.y * G * G;
        Kgl += (long)blockIdx.y * G * G;
        partials += (long)blockIdx.y * GR_MAXV * 1024;
    }
    double v[2] = {0.0, 0.0};
    const long n2 = (long)G * G;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n2; e += (long)gridDim.x * 256) {
        const int g = (int)(e / G), h = (int)(e % G);
        const double mk = M[e] * Kgl[e];
        if (ngl2 > 0) {
            const double d1 = gx1[g / ngl2] - gx1[h / ngl2], d2 = gx2[g % ngl2] - gx2[h % ngl2];
            v[0] += mk * (d1 * d1) / (ell1 * ell1 * ell1);
            v[1] += mk * (d2 * d2) / (ell2 * ell2 * ell2);
        } else {
            const double d1 = gx1[g] - gx1[h];
            v[0] += mk * (d1 * d1) / (ell1 * ell1 * ell1);
        }
    }
    block_partials<2>(v, partials, gridDim.x, blockIdx.x);
}

void k_kgl_grad(gpcsd_ctx *c, const double *M, const double *Kgl, const double *gx1, const double *gx2, int G, int ngl2,
                double ell1, double ell2, double *out2, hipStream_t s, const HpDev *tab, int B, long s_out) {
    const int nblocks = 512, nb = tab ? B : 1;
    double *part = c->buf<double>("grad_partials", (size_t)GR_MAXV * 1024 * nb);
    hipLaunchKernelGGL(kgl_grad_kernel, dim3(nblocks, nb), dim3(256), 0, s, M, Kgl, gx1, gx2, G, ngl2, ell1, ell2, part, tab);
    hipLaunchKernelGGL(final_sums_kernel, dim3(ngl2 > 0 ? 2 : 1, nb), dim3(256), 0, s, (const double *)part, nblocks, out2,
                       (long)GR_MAXV * 1024, s_out);
    GP_HIP(hipGetLastError());
}

// ---- spatial length scales on the 2D tensor grid: Kgl = K1 (x) K2 (covariances.py:216), so
//   <A^T Gs A, dKgl/dell1> = <Gs A, A (dK1 (x) K2)>,   <A^T Gs A, dKgl/dell2> = <Gs A, A (K1 (x) dK2)>:
// two elementwise inner products of nx x G matrices (the factors A (dK1 (x) K2), A (K1 (x) dK2) are Kronecker-structured small
// products, capi_grad.inl) instead of the G x G contraction of A^T (Gs A) with dKgl (1.1 GF + 2 x 1.44 M kernel terms at 384 x 1200).
__global__ __launch_bounds__(256) void frob_pair_kernel(const double *__restrict__ P, const double *__restrict__ T1,
                                                        const double *__restrict__ T2, long n, double *partials) {
    {                                           // blockIdx.y = hyper-parameter set: operands n apart, partials GR_MAXV*1024 apart
        const long b = blockIdx.y;
        P += b * n; T1 += b * n; T2 += b * n;
        partials += b * GR_MAXV * 1024;
    }
    double v[2] = {0.0, 0.0};
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const double p = P[e];
        v[0] += p * T1[e];
        v[1] += p * T2[e];
    }
    block_partials<2>(v, partials, gridDim.x, blockIdx.x);
}

void k_frob_pair(gpcsd_ctx *c, const double *P, const double *T1, const double *T2, long n, double *out2, hipStream_t s, int B,
                 long s_out) {
    const int nblocks = 256;
    double *part = c->buf<double>("grad_partials", (size_t)GR_MAXV * 1024 * B);
    hipLaunchKernelGGL(frob_pair_kernel, dim3(nblocks, B), dim3(256), 0, s, P, T1, T2, n, part);
    hipLaunchKernelGGL(final_sums_kernel, dim3(2, B), dim3(256), 0, s, (const double *)part, nblocks, out2, (long)GR_MAXV * 1024, s_out);
    GP_HIP(hipGetLastError());
}

// ---- forward-model radius: 2 sum_xg S_xg dA_xg/dR                      (forward_models.py:16, :53)
//   1D: A = w_g (sqrt(q+1) - sqrt(q)), q = (r/R)^2   ->  dA/dR = w_g (sqrt(q) - q / sqrt(q+1)) / R
//   2D: A = w_g (log(R+eps+sqrt((R+eps)^2+w^2)) - ...)  ->  dA/dR = w_g / sqrt((R+eps)^2 + w^2)
__global__ __launch_bounds__(256) void fwdR_grad_kernel(const double *__restrict__ S, const double *__restrict__ x, int nx,
                                                        const double *__restrict__ gx1, const double *__restrict__ gw1,
                                                        const double *__restrict__ gx2, const double *__restrict__ gw2, int G,
                                                        int ngl2, double R, double eps, double *partials,
                                                        const HpDev *__restrict__ tab) {
    if (tab) {                                  // blockIdx.y = hyper-parameter set: S nx*G apart, partials GR_MAXV*1024 apart
        R = tab[blockIdx.y].R;
        eps = tab[blockIdx.y].eps;
        S += (long)blockIdx.y * nx * G;
        partials += (long)blockIdx.y * GR_MAXV * 1024;
    }
    double v[1] = {0.0};
    const long n2 = (long)nx * G;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n2; e += (long)gridDim.x * 256) {
        const int i = (int)(e / G), g = (int)(e % G);
        double dA;
        if (ngl2 > 0) {
            const int g1 = g / ngl2, g2 = g % ngl2;
            const double d1 = gx1[g1] - x[2 * i], d2 = gx2[g2] - x[2 * i + 1];
            const double re = R + eps;
            dA = (gw1[g1] * gw2[g2]) / sqrt(re * re + (d1 * d1 + d2 * d2));
        } else {
            const double r = gx1[g] - x[i];
            const double q = (r / R) * (r / R);
            dA = gw1[g] * (sqrt(q) - q / sqrt(q + 1.0)) / R;
        }
        v[0] += 2.0 * S[e] * dA;
    }
    block_partials<1>(v, partials, gridDim.x, blockIdx.x);
}

void k_fwdR_grad(gpcsd_ctx *c, const double *S, const double *x, int nx, const double *gx1, const double *gw1, const double *gx2,
                 const double *gw2, int G, int ngl2, double R, double eps, double *out1, hipStream_t s, const HpDev *tab, int B,
                 long s_out) {
    const int nblocks = 256, nb = tab ? B : 1;
    double *part = c->buf<double>("grad_partials", (size_t)GR_MAXV * 1024 * nb);
    hipLaunchKernelGGL(fwdR_grad_kernel, dim3(nblocks, nb), dim3(256), 0, s, S, x, nx, gx1, gw1, gx2, gw2, G, ngl2, R, eps, part,
                       tab);
    hipLaunchKernelGGL(final_sums_kernel, dim3(1, nb), dim3(256), 0, s, (const double *)part, nblocks, out1,
                       (long)GR_MAXV * 1024, s_out);
    GP_HIP(hipGetLastError());
}

// out[x] = sum_k B[x*rowlen + k]^2 : per eigen-row sums of B^2 over all trials and times (d loglik / d sig2n_x for a
// per-electrode noise list, utility_functions.py:54-63).  One workgroup per x, fixed reduction order.
__global__ __launch_bounds__(256) void rowgroup_sumsq_kernel(const double *__restrict__ B, long rowlen, double *__restrict__ out) {
    __shared__ double red[4];
    const double *__restrict__ b = B + (long)blockIdx.x * rowlen;
    double s = 0.0;
    for (long k = threadIdx.x; k < rowlen; k += 256) s += b[k] * b[k];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

void k_rowgroup_sumsq(gpcsd_ctx *c, const double *B, int nrows, long rowlen, double *out, hipStream_t s) {
    hipLaunchKernelGGL(rowgroup_sumsq_kernel, dim3(nrows), dim3(256), 0, s, B, rowlen, out);
    GP_HIP(hipGetLastError());
}

// Per-electrode noise lists are attached to the EIGEN-index of Ks (utility_functions.py:54-63), so the objective is not a
// function of Ks alone: rotating the eigenvectors moves noise between them.  With B = alpha/D and S = sum_r B_r B_r^T
// the extra first-order term is  -1/2 sum_{x != y} dKhat_yx (sig_y - sig_x)/(es_x - es_y) S_xy, i.e. a symmetric
// correction of Ghat_s (what the reference's autograd obtains from the eigh VJP).  Pairs whose eigenvalues coincide to
// rounding are skipped (the reference produces inf/NaN there).
__global__ __launch_bounds__(256) void siglist_eigvec_term_kernel(double *__restrict__ Ghs, const double *__restrict__ Ssum,
                                                                  const double *__restrict__ es, const double *__restrict__ sig,
                                                                  int nx, double tiny) {
    {                                           // blockIdx.y = hyper-parameter set
        const long b = blockIdx.y;
        Ghs += b * nx * nx; Ssum += b * nx * nx; es += b * nx; sig += b * nx;
    }
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= (long)nx * nx) return;
    const int y = (int)(e / nx), x = (int)(e % nx);
    if (x == y) return;
    const double de = es[x] - es[y];
    if (fabs(de) <= tiny) return;
    Ghs[e] += -0.5 * (sig[y] - sig[x]) / de * Ssum[(long)x * nx + y];
}

void k_siglist_eigvec_term(gpcsd_ctx *c, double *Ghs, const double *Ssum, const double *es, const double *sig, int nx,
                           double tiny, hipStream_t s, int B) {
    hipLaunchKernelGGL(siglist_eigvec_term_kernel, dim3(ceil_div((long)nx * nx, 256), B), dim3(256), 0, s, Ghs, Ssum, es, sig, nx, tiny);
    GP_HIP(hipGetLastError());
}

// out[b] = sum_x sum_i alpha[(x*nb + b)*nt + i]^2 / D[x*nt + i] : per-trial whitened quadratic forms (one workgroup per
// trial, fixed reduction order).  alpha = Qs^T resid_b Qt for every trial b at once.
__global__ __launch_bounds__(256) void per_trial_quad_kernel(const double *__restrict__ alpha, const double *__restrict__ D, int nx,
                                                             int nb, int nt, double *__restrict__ out) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    double s = 0.0;
    for (long e = threadIdx.x; e < (long)nx * nt; e += 256) {
        const int x = (int)(e / nt), i = (int)(e % nt);
        const double a = alpha[((long)x * nb + b) * nt + i];
        s += a * a / D[e];
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) out[b] = s;
}

void k_per_trial_quad(gpcsd_ctx *c, const double *alpha, const double *D, int nx, int nb, int nt, double *out, hipStream_t s) {
    hipLaunchKernelGGL(per_trial_quad_kernel, dim3(nb), dim3(256), 0, s, alpha, D, nx, nb, nt, out);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
