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

// T factor of one compact-WY panel: G = V_p V_p^T, then T = (diag(1 / tau) + striu(G))^-1 (LAPACK's dlarft, written as an
// inverse).  Body of the preparation launch for panel p of problem P; all 1024 threads of the workgroup call it (barriers).
//  * G: the panel is staged through LDS in chunks of KC columns; wave (kq, bi, bj) accumulates the 32 x 32 block (bi, bj) of G
//    over the kq-th quarter of each chunk as 2 x 2 MFMA fragments (four LDS operand reads per four MFMAs: with one fragment
//    per wave over the whole K range the sixteen waves were LDS-bandwidth-bound, 8 us of a 30 us launch), the four partial
//    sums are added in a fixed order.
//  * the inverse, blocked: the four 16 x 16 diagonal blocks by back substitution (four columns per wave, sixteen steps at most
//    instead of 64: the serial walk over a whole 64-column T was the other half of the launch), then the off-diagonal blocks
//    by doubling, T_ab = -T_aa M_ab T_bb at block size 16 and 32, MFMA products through LDS.
// LDS of the caller: vs = WY_NB x (KC + 2) doubles (chunk, [reflector][k], row stride = 2 mod 32); g, tl = WY_NB x WY_LDG doubles
// each (G, T); pl = 32 x WY_LDP doubles (products); st = WY_NB doubles.  g, tl and pl may lie inside vs (the chunk is dead when G
// is stored).  KC = 64 inside the leaf launch, 256 in the launch of its own (a whole 250-row panel at once: the launch is on
// the critical path of the tridiagonal forms).
constexpr int WY_LDG = WY_NB + 1;                          // row stride of G and T in LDS
constexpr int WY_LDP = 33;                                 // row stride of the 32-row product block

template <int KC>
__device__ __forceinline__ void wy_prep_body(const WyProb &P, const int p, const int tid, double *__restrict__ vs, double *g,
                                             double *tl, double *pl, double *__restrict__ st, unsigned long long *clk = nullptr) {
    if (p >= P.npanels) return;
    auto stamp = [&](int k) { if (clk && tid == 0) clk[k] = wall_clock64(); };
    stamp(0);
    const int n = P.n;
    constexpr int LDV = KC + 2, LDG = WY_LDG;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const double *__restrict__ Vp = P.V + (long)p * WY_NB * n;
    {
        const int kq = wid >> 2, bi = (wid >> 1) & 1, bj = wid & 1;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
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
            stamp(1);
            const int kl = ((kn / 4 + 3) / 4) * 4;         // MFMA steps of four columns: a quarter of them per kq
            const int k_lo = min(kn, kq * kl), k_hi = min(kn, k_lo + kl);
            const double *__restrict__ va = vs + (32 * bi + fr) * LDV + fq, *__restrict__ vb = vs + (32 * bj + fr) * LDV + fq;
#pragma unroll 4
            for (int k0 = k_lo; k0 < k_hi; k0 += 4) {
                const double a0 = va[k0], a1 = va[16 * LDV + k0], b0 = vb[k0], b1 = vb[16 * LDV + k0];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();                                   // g, tl, pl may live where the chunk was
        // the four partial sums one after the other (fixed order); T starts as zero (its lower blocks are never written otherwise)
        for (int idx = tid; idx < WY_NB * LDG; idx += 1024) tl[idx] = 0.0;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            if (kq == q) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double *o = g + (32 * bi + 16 * i + fq + 4 * r) * LDG + 32 * bj + 16 * j + fr;
                            *o = (q == 0) ? acc[i][j][r] : *o + acc[i][j][r];
                        }
            }
            if (q == 0 && tid < WY_NB) {
                const int kk = p * WY_NB + tid;
                st[tid] = (kk < P.nrefl) ? P.tau[kk] : 0.0;
            }
            __syncthreads();
        }
    }
    stamp(2);
    // diagonal blocks: column c of block b solves (diag(1/tau) + striu(G))_bb x = e_c; lane l < 16 carries the running right-hand
    // side b_l.  The four columns of a wave are independent chains walked together, lane reads stay in the VALU; the LDS reads of
    // step j - 1 are issued before the arithmetic of step j.
    {
        const int b0 = 16 * (wid >> 2), cb = 4 * (wid & 3);
        double bv[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[q] = (lane == cb + q) ? 1.0 : 0.0;
            x[q] = 0.0;
        }
        const double *__restrict__ grow = g + (b0 + (lane & 15)) * LDG + b0;
        double tj = st[b0 + cb + 3], gj = grow[cb + 3];
        for (int j = cb + 3; j >= 0; --j) {
            const int jn = j > 0 ? j - 1 : 0;
            const double tn = st[b0 + jn], gn = grow[jn];
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
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tl[(b0 + lane) * LDG + b0 + cb + q] = (lane <= cb + q) ? x[q] : 0.0;
        }
    }
    __syncthreads();
    // off-diagonal blocks by doubling: T_ab = -T_aa (G_ab T_bb), first (a, b) = (0, 1) and (2, 3) at block size 16, then the
    // 32 x 32 block above the diagonal
    auto frag = [&](const double *A, int lda, const double *B, int ldb, int K) {      // A[fr][k] B[k][fr] over k < K
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[fr * lda + k0 + fq], B[(k0 + fq) * ldb + fr], acc, 0, 0, 0);
        return acc;
    };
    if (wid < 2) {
        const int a = 32 * wid, b = a + 16;
        const d4 pr = frag(g + a * LDG + b, LDG, tl + b * LDG + b, LDG, 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[(16 * wid + fq + 4 * r) * WY_LDP + fr] = pr[r];
    }
    __syncthreads();
    if (wid < 2) {
        const int a = 32 * wid, b = a + 16;
        const d4 tr = frag(tl + a * LDG + a, LDG, pl + 16 * wid * WY_LDP, WY_LDP, 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[(a + fq + 4 * r) * LDG + b + fr] = -tr[r];
    }
    __syncthreads();
    if (wid < 4) {
        const int fi = wid >> 1, fj = wid & 1;
        const d4 pr = frag(g + (16 * fi) * LDG + 32, LDG, tl + 32 * LDG + 32 + 16 * fj, LDG, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[(16 * fi + fq + 4 * r) * WY_LDP + 16 * fj + fr] = pr[r];
    }
    __syncthreads();
    if (wid < 4) {
        const int fi = wid >> 1, fj = wid & 1;
        const d4 tr = frag(tl + (16 * fi) * LDG, LDG, pl + 16 * fj, WY_LDP, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[(16 * fi + fq + 4 * r) * LDG + 32 + 16 * fj + fr] = -tr[r];
    }
    __syncthreads();
    double *__restrict__ Tg = P.T + (long)p * WY_NB * WY_NB;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 1024 * u, r = idx >> 6, cc = idx & 63;
        Tg[idx] = tl[r * LDG + cc];
    }
    stamp(3);
}

// the body inside another launch (the D&C leaf launch): static LDS, 64-column chunks, T on the chunk's storage
__device__ __forceinline__ void wy_prep_role(const WyProb &P, const int p, const int tid) {
    __shared__ double g[WY_NB * WY_LDG];
    __shared__ double pl[32 * WY_LDP];
    __shared__ double st[WY_NB];
    __shared__ double vs[WY_NB * (WY_KC + 2)];
    static_assert(WY_NB * (WY_KC + 2) >= WY_NB * WY_LDG, "T does not fit the chunk's storage");
    wy_prep_body<WY_KC>(P, p, tid, vs, g, /*tl=*/vs, pl, st);
}

constexpr int WY_PREP_KC = 256;       // chunk of the launch of its own: G, T and the products all on the chunk's storage
constexpr size_t WY_PREP_LDS = ((size_t)WY_NB * (WY_PREP_KC + 2) + WY_NB) * sizeof(double);
static_assert(WY_NB * (WY_PREP_KC + 2) >= 2 * WY_NB * WY_LDG + 32 * WY_LDP, "G, T, products do not fit the chunk's storage");

}  // namespace gpcsd
