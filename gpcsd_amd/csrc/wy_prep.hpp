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
// LDS of the caller: vs = WY_NB x (KC + 2) doubles (one K chunk of the panel, [reflector][k], row stride = 2 mod 32), g =
// WY_NB x (WY_NB + 1) doubles (may be the same storage as vs: the chunk is dead when G is stored), st = WY_NB doubles.
// KC = 64 inside the leaf launch (little LDS beside the leaf units); the launch of its own stages a whole 250-row panel at
// once (KC = 256): one round of loads and two barriers instead of four of each, the launch is on the critical path of the
// log-likelihood's tridiagonal form.
template <int KC>
__device__ __forceinline__ void wy_prep_body(const WyProb &P, const int p, const int tid, double *__restrict__ vs,
                                             double *g, double *__restrict__ st) {
    if (p >= P.npanels) return;
    const int n = P.n;
    constexpr int LDV = KC + 2, LDG = WY_NB + 1;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const double *__restrict__ Vp = P.V + (long)p * WY_NB * n;
    {
        const int fa = wid >> 2, fb = wid & 3;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const int kstart = (p * WY_NB) & ~3;               // reflector k is zero up to column k
        for (int kc = kstart; kc < n; kc += KC) {
            const int kn = min(KC, (n - kc + 3) & ~3);     // columns of this chunk that are not padding
            __syncthreads();
#pragma unroll
            for (int u = 0; u < WY_NB * KC / 1024; ++u) {
                const int idx = tid + 1024 * u;
                const int r = idx / KC, k = idx % KC;
                if (k < kn) vs[r * LDV + k] = (kc + k < n) ? Vp[(long)r * n + kc + k] : 0.0;
            }
            __syncthreads();
            const double *__restrict__ va = vs + (16 * fa + fr) * LDV + fq, *__restrict__ vb = vs + (16 * fb + fr) * LDV + fq;
#pragma unroll 8
            for (int k0 = 0; k0 < kn; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[k0], vb[k0], acc, 0, 0, 0);
        }
        __syncthreads();                                   // g may live where the chunk was
#pragma unroll
        for (int r = 0; r < 4; ++r) g[(16 * fa + fq + 4 * r) * LDG + 16 * fb + fr] = acc[r];
    }
    if (tid < WY_NB) {
        const int kk = p * WY_NB + tid;
        st[tid] = (kk < P.nrefl) ? P.tau[kk] : 0.0;
    }
    __syncthreads();
    // column c of T solves (diag(1/tau) + striu(G)) x = e_c; lane l carries the running right-hand side b_l.  The four
    // columns of a wave are independent chains walked together (j runs over the longest), lane reads stay in the VALU; the
    // LDS reads of step j - 1 are issued before the arithmetic of step j.
    {
        const int cb = wid * 4;
        double bv[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[q] = (lane == cb + q) ? 1.0 : 0.0;
            x[q] = 0.0;
        }
        const double *__restrict__ grow = g + lane * LDG;
        double tj = st[cb + 3], gj = grow[cb + 3];
        for (int j = cb + 3; j >= 0; --j) {
            const int jn = j > 0 ? j - 1 : 0;
            const double tn = st[jn], gn = grow[jn];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (j <= cb + q) {                         // wave-uniform
                    const double xj = tj * lane_get(bv[q], j);
                    if (lane == j) x[q] = xj;
                    if (lane < j) bv[q] -= gj * xj;
                }
            }
            tj = tn;
            gj = gn;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            P.T[(long)p * WY_NB * WY_NB + (long)lane * WY_NB + cb + q] = (lane <= cb + q) ? x[q] : 0.0;
    }
}

// the body inside another launch (the D&C leaf launch): static LDS, 64-column chunks
__device__ __forceinline__ void wy_prep_role(const WyProb &P, const int p, const int tid) {
    __shared__ double g[WY_NB * (WY_NB + 1)];
    __shared__ double st[WY_NB];
    __shared__ double vs[WY_NB * (WY_KC + 2)];
    wy_prep_body<WY_KC>(P, p, tid, vs, g, st);
}

constexpr int WY_PREP_KC = 256;       // chunk of the launch of its own
constexpr size_t WY_PREP_LDS = ((size_t)WY_NB * (WY_PREP_KC + 2) + WY_NB) * sizeof(double);

}  // namespace gpcsd
