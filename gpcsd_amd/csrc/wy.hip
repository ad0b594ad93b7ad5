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
__global__ __launch_bounds__(1024) void wy_prep_kernel(WyBatch b) {
    extern __shared__ double psm[];
    double *vs = psm, *st = psm + WY_NB * (WY_PREP_KC + 2);
    wy_prep_body<WY_PREP_KC>(wy_resolve(b, blockIdx.y), blockIdx.x, threadIdx.x, vs, /*g, tl, pl on the chunk's storage*/ vs,
                             vs + WY_NB * WY_LDG, vs + 2 * WY_NB * WY_LDG, st, (blockIdx.x == 0 && blockIdx.y == 0) ? b.clk : nullptr);
}

constexpr int WY_NT = 1024;           // threads of an apply workgroup
constexpr int WY_KS = 4;              // the K range of W1 = V_p Zc split over this many wave groups
constexpr int WY_LDT = WY_NB + 1;     // LDS row stride of T_p

// one workgroup = 16 columns of Z resident in LDS, all panels applied in sequence.  Sixteen waves: the 64 x 16 product
// W1 = V_p Zc has four fragments only, so its K range is split four ways (partial sums in LDS, added up by the readers);
// the update of Zc has one row fragment per wave at 250 rows.  (Four waves, each walking the whole K range and four row
// fragments: 47 us per launch at 4 x 250 rows, this form 1/3 less; the launch forms Q on the critical path of the
// log-likelihood's tridiagonal form)  KS = 1 when the partial sums would not fit beside a long slab.
template <int KS>
__global__ __launch_bounds__(WY_NT) void wy_apply_kernel(WyBatch b) {
    const WyProb P = wy_resolve(b, blockIdx.y);
    const int n = P.n;
    const int c0 = blockIdx.x * WY_ZC;
    if (c0 >= n) return;
    extern __shared__ double smem[];
    double *Zs = smem;                         // [n][WY_LD]
    double *W1 = Zs + (size_t)n * WY_LD;       // [KS][64][WY_LD]
    double *W2 = W1 + KS * WY_NB * WY_LD;      // [64][WY_LD]
    double *Ts = W2 + WY_NB * WY_LD;           // [64][WY_LDT]: T_p (KS > 1 only)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const bool stamping = b.clk && blockIdx.y == 0 && blockIdx.x == gridDim.x - 1 && tid == 0;
    int nstamp = 8;
    auto stamp = [&]() { if (stamping && nstamp < 32) b.clk[nstamp++] = wall_clock64(); };
    stamp();
    if (blockIdx.x == 0 && P.w_scale) {        // eigenvalues back to the scale of the input matrix (nobody reads them here)
        const double m = P.amax[0];
        for (int i = tid; i < n; i += WY_NT) P.w_scale[i] *= m;
    }
    for (int idx = tid; idx < n * WY_ZC; idx += WY_NT) {
        const int r = idx / WY_ZC, j = idx % WY_ZC;
        if (P.z_identity) Zs[r * WY_LD + j] = (r == c0 + j) ? 1.0 : 0.0;        // Q itself: the panels applied to the identity
        else Zs[r * WY_LD + j] = (c0 + j < n) ? P.Z[(long)r * n + c0 + j] : 0.0;
    }
    __syncthreads();
    stamp();
    const int nfrag = (n + 15) / 16;
    struct W1Range { int fa, ks, kb, ke; };
    auto w1_range = [&](int p) {
        const int kstart = (p * WY_NB) & ~3;
        const int klen = (((n - kstart + KS - 1) / KS) + 3) & ~3;
        W1Range w;
        w.fa = wid & 3; w.ks = wid >> 2;
        w.kb = kstart + w.ks * klen; w.ke = min(n, w.kb + klen);
        return w;
    };
    auto load8 = [&](double (&dst)[8], const double *__restrict__ ra, int k0, int ke) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + fq;
            dst[u] = ra[k < ke ? k : n - 1];
        }
    };
    double a8[8];                              // first batch of the W1 operand of the panel about to be applied
    auto preload_w1 = [&](int p) {
        if (wid < 4 * KS) {
            const W1Range w = w1_range(p);
            if (w.kb < w.ke) load8(a8, P.V + (long)p * WY_NB * n + (long)(16 * w.fa + fr) * n, w.kb, w.ke);
        }
    };
    if (P.npanels > 0) preload_w1(P.npanels - 1);
    for (int p = P.npanels - 1; p >= 0; --p) {
        const double *__restrict__ Vp = P.V + (long)p * WY_NB * n;
        const double *__restrict__ Tp = P.T + (long)p * WY_NB * WY_NB;
        const int kstart = (p * WY_NB) & ~3;
        // T_p on its way to LDS (KS > 1: there is room): four coalesced loads per thread issued here, in front of the W1 product
        // they do not depend on -- as strided fragment loads in the W2 phase they were a 2 us round trip per panel
        double tl4[4];
        if (KS > 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) tl4[u] = Tp[tid + WY_NT * u];
        }
        // W1 = V_p Zc : wave (ks, fa) owns panel rows 16 fa .. 16 fa + 15 over the ks-th part of the K range.  The panel rows come
        // straight from L2: batches of eight clamped (branch-free) loads, the NEXT batch issued before the MFMAs of the current
        // one; the first batch of a panel was issued before the previous panel's update phase (a8 is carried across panels).
        if (wid < 4 * KS) {
            const W1Range w = w1_range(p);
            const double *__restrict__ ra = Vp + (long)(16 * w.fa + fr) * n;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            double b8[8];
            auto mma8 = [&](const double (&src)[8], int k0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 4 * u + fq;
                    const double bb = Zs[(k < w.ke ? k : n - 1) * WY_LD + fr];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < w.ke ? src[u] : 0.0, bb, acc, 0, 0, 0);
                }
            };
            for (int k0 = w.kb; k0 < w.ke; k0 += 64) {
                if (k0 + 32 < w.ke) load8(b8, ra, k0 + 32, w.ke);
                mma8(a8, k0);
                if (k0 + 32 < w.ke) {
                    if (k0 + 64 < w.ke) load8(a8, ra, k0 + 64, w.ke);
                    mma8(b8, k0 + 32);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W1[(w.ks * WY_NB + 16 * w.fa + fq + 4 * r) * WY_LD + fr] = acc[r];
        }
        if (KS > 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + WY_NT * u;
                Ts[(idx >> 6) * WY_LDT + (idx & 63)] = tl4[u];
            }
        }
        lds_barrier();
        stamp();
        // W2 = T_p W1 (four waves; the others go on to the loads of the update)
        if (wid < 4) {
            const double *__restrict__ ta = Tp + (long)(16 * wid + fr) * WY_NB;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = 4 * u + fq;
                double w = W1[k * WY_LD + fr];
#pragma unroll
                for (int q = 1; q < KS; ++q) w += W1[(q * WY_NB + k) * WY_LD + fr];
                const double tv = (KS > 1) ? Ts[(16 * wid + fr) * WY_LDT + k] : ta[k];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv, w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W2[(16 * wid + fq + 4 * r) * WY_LD + fr] = acc[r];
        }
        // Zc -= V_p^T W2 : rows below the panel's first reflector only, a row fragment in two halves of eight MFMA steps:
        // the loads of the next half are issued before the MFMAs of the current one, the first before the barrier that
        // publishes W2.
        {
            double ua[8], ub[8];
            auto loadh = [&](double (&dst)[8], int fm, int h) {
                const int m = 16 * fm + fr;
                const double *__restrict__ vm = Vp + (m < n ? m : n - 1) + (long)(32 * h + fq) * n;   // clamped column
#pragma unroll
                for (int u = 0; u < 8; ++u) dst[u] = vm[(long)(4 * u) * n];
            };
            auto mmah = [&](const double (&src)[8], int h, d4 &acc) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(src[u], W2[(32 * h + 4 * u + fq) * WY_LD + fr], acc, 0, 0, 0);
            };
            constexpr int NW = WY_NT / 64;
            int fm = (p * WY_NB) / 16 + wid;
            if (fm < nfrag) loadh(ua, fm, 0);
            if (p > 0) preload_w1(p - 1);                     // (reads V only: independent of this panel's update)
            lds_barrier();
            stamp();
            for (; fm < nfrag; fm += NW) {
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                loadh(ub, fm, 1);
                mmah(ua, 0, acc);
                if (fm + NW < nfrag) loadh(ua, fm + NW, 0);
                mmah(ub, 1, acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * fm + fq + 4 * r;
                    if (row < n) Zs[row * WY_LD + fr] -= acc[r];
                }
            }
        }
        lds_barrier();
        stamp();
    }
    for (int idx = tid; idx < n * WY_ZC; idx += WY_NT) {
        const int r = idx / WY_ZC, j = idx % WY_ZC;
        if (c0 + j < n) P.Z[(long)r * n + c0 + j] = Zs[r * WY_LD + j];
    }
    stamp();
}

static bool wy_clk_on() {
    static const bool on = getenv("GPCSD_WY_CLK") && getenv("GPCSD_WY_CLK")[0] == '1';
    return on;
}
static void wy_clk_print(gpcsd_ctx *c, unsigned long long *d, const char *what, int first, int n, hipStream_t s) {
    unsigned long long h[32];
    GP_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[%s] phases (10 ns ticks):", what);
    for (int i = first + 1; i < first + n; ++i) fprintf(stderr, " %llu", h[i] >= h[i - 1] ? h[i] - h[i - 1] : 0ull);
    fprintf(stderr, "\n");
}

static void wy_prep_launch(const WyBatch &b, int maxP, int count, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)WY_PREP_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(wy_prep_kernel, dim3(maxP, count), dim3(1024), WY_PREP_LDS, s, b);
}
static WyBatch wy_with_clk(gpcsd_ctx *c, const WyBatch &b, hipStream_t s) {
    WyBatch w = b;
    if (wy_clk_on()) {
        w.clk = c->buf<unsigned long long>("wy_clk", 32);
        GP_HIP(hipMemsetAsync(w.clk, 0, 32 * sizeof(unsigned long long), s));
    }
    return w;
}

void wy_prep_device(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s) {
    int maxP = 0;
    for (int i = 0; i < nclass; ++i) maxP = std::max(maxP, b.p[i].npanels);
    if (maxP == 0) return;
    const WyBatch w = wy_with_clk(c, b, s);
    wy_prep_launch(w, maxP, b.start[MAX_EIG_BATCH], s);
    if (w.clk) wy_clk_print(c, w.clk, "wy_prep: load, G, back substitution", 0, 4, s);
    GP_HIP(hipGetLastError());
}

static size_t wy_apply_lds(int nmax, int ks) {
    return ((size_t)nmax * WY_LD + (size_t)(ks + 1) * WY_NB * WY_LD + (ks > 1 ? (size_t)WY_NB * WY_LDT : 0)) * sizeof(double);
}
bool wy_fused_supported(int nmax) { return wy_apply_lds(nmax, 1) <= 160 * 1024; }

void wy_batch_device(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s, bool prep_done) {
    int maxP = 0, nmax = 0;
    for (int i = 0; i < nclass; ++i) {
        maxP = std::max(maxP, b.p[i].npanels);
        nmax = std::max(nmax, b.p[i].n);
    }
    if (maxP == 0) return;
    const int count = b.start[MAX_EIG_BATCH];          // all replicas of all classes
    const bool split = wy_apply_lds(nmax, WY_KS) <= 160 * 1024;
    const size_t sh = wy_apply_lds(nmax, split ? WY_KS : 1);
    static bool attr_set = false;
    if (!attr_set) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_apply_kernel<WY_KS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_apply_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_set = true;
    }
    const WyBatch w = wy_with_clk(c, b, s);
    if (!prep_done) wy_prep_launch(w, maxP, count, s);
    if (split) hipLaunchKernelGGL(wy_apply_kernel<WY_KS>, dim3(ceil_div(nmax, WY_ZC), count), dim3(WY_NT), sh, s, w);
    else hipLaunchKernelGGL(wy_apply_kernel<1>, dim3(ceil_div(nmax, WY_ZC), count), dim3(WY_NT), sh, s, w);
    if (w.clk) wy_clk_print(c, w.clk, "wy_apply, last column block: init, then per panel W1 | W2 + loads | update, store", 8, 2 + 3 * maxP + 1, s);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
