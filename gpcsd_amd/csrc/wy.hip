// Back-transformation of the tridiagonal eigenvectors, Z <- (H_0 H_1 ... H_{n-3}) Z, for several independent problems in
// TWO launches (stage 3 of the large-n eigensolver; LAPACK's dormtr behind numpy.linalg.eigh).
//
// Reflectors are grouped in panels of 64: Q_p = I - V_p^T T_p V_p (V_p holds the reflectors by rows), and
// Q Z = Q_0 (Q_1 (... (Q_{P-1} Z))).  Columns of Z are independent, so one workgroup keeps a 16-column slab of Z in LDS
// and walks ALL panels on it with fp64 MFMA 16x16x4 -- no launch per panel, no traffic for Z between panels:
//     W1 = V_p Zc (64x16),  W2 = T_p W1,  Zc -= V_p^T W2.
// The T factors come from a preparation launch (one workgroup per panel and problem): G = V_p V_p^T on MFMA, then
// T = (diag(1/tau) + striu(G))^{-1} by wave-parallel back substitution (closed form of the compact-WY T factor).
// Every launch-bound GEMM chain this replaces cost ~8 us per panel and problem.
#include <algorithm>

#include "devutil.hpp"
#include "kernels.hpp"
#include "wy_prep.hpp"

namespace gpcsd {

constexpr int WY_ZC = 16;             // columns of Z per workgroup (one MFMA fragment wide)
constexpr int WY_LD = WY_ZC + 2;      // LDS row stride: 18*i mod 32 gives distinct even slots for the b64 fragment reads

// G = V_p V_p^T (16 waves, one 16x16 fragment each), then T by back substitution (4 columns per wave): wy_prep.hpp
__global__ __launch_bounds__(1024) void wy_prep_kernel(WyBatch b) { wy_prep_body(wy_resolve(b, blockIdx.y), blockIdx.x, threadIdx.x); }

// one workgroup = 16 columns of Z resident in LDS, all panels applied in sequence
__global__ __launch_bounds__(256) void wy_apply_kernel(WyBatch b) {
    const WyProb P = wy_resolve(b, blockIdx.y);
    const int n = P.n;
    const int c0 = blockIdx.x * WY_ZC;
    if (c0 >= n) return;
    extern __shared__ double smem[];
    double *Zs = smem;                         // [n][WY_LD]
    double *W1 = Zs + (size_t)n * WY_LD;       // [64][WY_LD]
    double *W2 = W1 + WY_NB * WY_LD;           // [64][WY_LD]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    if (blockIdx.x == 0 && P.w_scale) {        // eigenvalues back to the scale of the input matrix (nobody reads them here)
        const double m = P.amax[0];
        for (int i = tid; i < n; i += 256) P.w_scale[i] *= m;
    }
    for (int idx = tid; idx < n * WY_ZC; idx += 256) {
        const int r = idx / WY_ZC, j = idx % WY_ZC;
        if (P.z_identity) Zs[r * WY_LD + j] = (r == c0 + j) ? 1.0 : 0.0;        // Q itself: the panels applied to the identity
        else Zs[r * WY_LD + j] = (c0 + j < n) ? P.Z[(long)r * n + c0 + j] : 0.0;
    }
    __syncthreads();
    const int nfrag = (n + 15) / 16;
    for (int p = P.npanels - 1; p >= 0; --p) {
        const double *__restrict__ Vp = P.V + (long)p * WY_NB * n;
        const double *__restrict__ Tp = P.T + (long)p * WY_NB * WY_NB;
        const int kstart = (p * WY_NB) & ~3;
        // W1 = V_p Zc : wave w owns panel rows 16w .. 16w+15
        {
            const double *__restrict__ ra = Vp + (long)(16 * wid + fr) * n;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            // the panel rows come straight from L2: batches of eight clamped (branch-free) loads, the NEXT batch issued
            // before the MFMAs of the current one, so the chain never waits for a full L2 round trip
            double a8[8], b8[8];
            auto load8 = [&](double (&dst)[8], int k0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + fq;
                    dst[u] = ra[k < n ? k : n - 1];
                }
            };
            auto mma8 = [&](const double (&src)[8], int k0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + fq;
                    const double bb = Zs[(k < n ? k : n - 1) * WY_LD + fr];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < n ? src[u] : 0.0, bb, acc, 0, 0, 0);
                }
            };
            load8(a8, kstart);
            for (int k0 = kstart; k0 < n; k0 += 64) {
                if (k0 + 32 < n) load8(b8, k0 + 32);
                mma8(a8, k0);
                if (k0 + 32 < n) {
                    if (k0 + 64 < n) load8(a8, k0 + 64);
                    mma8(b8, k0 + 32);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W1[(16 * wid + fq + 4 * r) * WY_LD + fr] = acc[r];
        }
        __syncthreads();
        // W2 = T_p W1
        {
            const double *__restrict__ ta = Tp + (long)(16 * wid + fr) * WY_NB;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < WY_NB; k0 += 4) {
                const int k = k0 + fq;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[k], W1[k * WY_LD + fr], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W2[(16 * wid + fq + 4 * r) * WY_LD + fr] = acc[r];
        }
        __syncthreads();
        // Zc -= V_p^T W2 : rows below the panel's first reflector only
        {
            // the sixteen V_p^T loads of the NEXT row fragment are issued before the MFMAs of the current one
            double a16[16], b16[16];
            auto loadf = [&](double (&dst)[16], int fm) {
                const int m = 16 * fm + fr;
                const double *__restrict__ vm = Vp + (m < n ? m : n - 1);   // clamped column: rows >= n are never stored
#pragma unroll
                for (int u = 0; u < 16; ++u) dst[u] = vm[(long)(4 * u + fq) * n];
            };
            auto applyf = [&](const double (&src)[16], int fm) {
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(src[u], W2[(4 * u + fq) * WY_LD + fr], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * fm + fq + 4 * r;
                    if (row < n) Zs[row * WY_LD + fr] -= acc[r];
                }
            };
            int fm = (p * WY_NB) / 16 + wid;
            if (fm < nfrag) loadf(a16, fm);
            for (; fm < nfrag; fm += 8) {
                if (fm + 4 < nfrag) loadf(b16, fm + 4);
                applyf(a16, fm);
                if (fm + 4 < nfrag) {
                    if (fm + 8 < nfrag) loadf(a16, fm + 8);
                    applyf(b16, fm + 4);
                }
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < n * WY_ZC; idx += 256) {
        const int r = idx / WY_ZC, j = idx % WY_ZC;
        if (c0 + j < n) P.Z[(long)r * n + c0 + j] = Zs[r * WY_LD + j];
    }
}

void wy_prep_device(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s) {
    int maxP = 0;
    for (int i = 0; i < nclass; ++i) maxP = std::max(maxP, b.p[i].npanels);
    if (maxP == 0) return;
    hipLaunchKernelGGL(wy_prep_kernel, dim3(maxP, b.start[MAX_EIG_BATCH]), dim3(1024), 0, s, b);
    GP_HIP(hipGetLastError());
}

bool wy_fused_supported(int nmax) { return ((size_t)nmax * WY_LD + 2 * WY_NB * WY_LD) * sizeof(double) <= 160 * 1024; }

void wy_batch_device(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s, bool prep_done) {
    int maxP = 0, nmax = 0;
    for (int i = 0; i < nclass; ++i) {
        maxP = std::max(maxP, b.p[i].npanels);
        nmax = std::max(nmax, b.p[i].n);
    }
    if (maxP == 0) return;
    const int count = b.start[MAX_EIG_BATCH];          // all replicas of all classes
    const size_t sh = ((size_t)nmax * WY_LD + 2 * WY_NB * WY_LD) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_set = true;
    }
    if (!prep_done) hipLaunchKernelGGL(wy_prep_kernel, dim3(maxP, count), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(wy_apply_kernel, dim3(ceil_div(nmax, WY_ZC), count), dim3(256), sh, s, b);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
