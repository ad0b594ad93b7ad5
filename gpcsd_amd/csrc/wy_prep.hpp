// Preparation stage of the fused compact-WY back-transformation (see wy.hip): T factors of all panels.  A header because
// the same body runs either as its own launch (wy_prep_kernel) or as a role of the D&C leaf launch (stedc.hip:
// dc_leaf_wyprep_kernel) -- it only needs the reflectors, so it hides behind the leaf eigenproblems instead of taking a
// launch of its own in the dependent chain.
#pragma once
#include "devutil.hpp"
#include "kernels.hpp"

namespace gpcsd {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int WY_NB = 64;
constexpr int WY_KC = 64;             // K chunk of a panel staged through LDS by the preparation kernel

// G = V_p V_p^T (16 waves, one 16x16 fragment each), then T by back substitution (4 columns per wave).
// Body of the preparation launch for panel p of problem P; all 1024 threads of the workgroup call it (it has barriers).
__device__ __forceinline__ void wy_prep_body(const WyProb &P, const int p, const int tid) {
    if (p >= P.npanels) return;
    const int n = P.n;
    __shared__ double g[WY_NB][WY_NB + 1];
    __shared__ double st[WY_NB];
    __shared__ double vs[WY_NB][WY_KC + 2];                // one K chunk of the panel, [reflector][k], stride = 2 mod 32
    const int lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const double *__restrict__ Vp = P.V + (long)p * WY_NB * n;
    {
        // G = V_p V_p^T.  The panel is staged through LDS in chunks of WY_KC columns with coalesced loads (the direct
        // version issued one dependent L2 round trip per MFMA step: 63 of them at n = 250).
        const int fa = wid >> 2, fb = wid & 3;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const int kstart = (p * WY_NB) & ~3;               // reflector k is zero up to column k
        for (int kc = kstart; kc < n; kc += WY_KC) {
            __syncthreads();
            for (int idx = tid; idx < WY_NB * WY_KC; idx += 1024) {
                const int r = idx / WY_KC, k = idx % WY_KC;
                vs[r][k] = (kc + k < n) ? Vp[(long)r * n + kc + k] : 0.0;
            }
            __syncthreads();
#pragma unroll 4
            for (int k0 = 0; k0 < WY_KC; k0 += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[16 * fa + fr][k0 + fq], vs[16 * fb + fr][k0 + fq], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) g[16 * fa + fq + 4 * r][16 * fb + fr] = acc[r];
    }
    if (tid < WY_NB) {
        const int kk = p * WY_NB + tid;
        st[tid] = (kk < P.nrefl) ? P.tau[kk] : 0.0;
    }
    __syncthreads();
    // column c of T solves (diag(1/tau) + striu(G)) x = e_c; lane l carries the running right-hand side b_l.  The four
    // columns of a wave are independent chains walked together (j runs over the longest), lane reads stay in the VALU.
    {
        const int cb = wid * 4;
        double bv[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[q] = (lane == cb + q) ? 1.0 : 0.0;
            x[q] = 0.0;
        }
        for (int j = cb + 3; j >= 0; --j) {
            const double tj = st[j], gj = g[lane][j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (j <= cb + q) {                         // wave-uniform
                    const double xj = tj * lane_get(bv[q], j);
                    if (lane == j) x[q] = xj;
                    if (lane < j) bv[q] -= gj * xj;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            P.T[(long)p * WY_NB * WY_NB + (long)lane * WY_NB + cb + q] = (lane <= cb + q) ? x[q] : 0.0;
    }
}


}  // namespace gpcsd
