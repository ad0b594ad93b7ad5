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
#include <functional>

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

// ------------------------------------------------------------------------------------------------------------------
// Q panel by panel, beside the tridiagonalisation that is still producing the later panels (stage 5 of a staged chain).
//
// Reflector i is zero up to column i, so H_i only touches columns > i of whatever it multiplies from the right: the columns
// [0, 64 (p + 1)) of Q = H_0 H_1 .. are final once panel p is.  Q is accumulated FORWARD on its transpose,
//     Zt = Q^T = P_{np-1}^T .. P_1^T P_0^T,   P_p^T = I - V_p^T T_p^T V_p,
// one launch per panel: a workgroup keeps a 16-column slab of Zt -- sixteen rows of Q -- in LDS exactly as the
// back-transformation does, applies ONE panel and writes the rows of Zt from the panel's first column on back to Q (row-major,
// Q[r][j] = Zt[j][r]): the panel's own 64 columns of Q are final, the rest is the state the next panel's launch reads.
// The T factor of the panel comes from a launch of the preparation body that first WAITS for the register tail's progress word
// (sytrd_regtail.hpp: reflectors and tau of the panel are in global memory): the only place where a launch of one stream spins
// on a running kernel of another.  The wait is bounded; when the time is up the launch reports failure 7 and goes on.
// ------------------------------------------------------------------------------------------------------------------
constexpr unsigned WY_PROG_DONE = 0x7fffffffu;        // = RT_PROG_DONE
// (the gate's patience travels in WyBatch::gate_ticks: 0.2 s of the 100 MHz wall clock by default -- a tail takes < 1 ms)

// One launch per panel, FOUR waves per workgroup and ~75 KB of LDS: the launches run beside the previous call's large products,
// under which a CU never drains -- a workgroup that needs a whole CU (the back-transformation's 1024 threads, the preparation
// launch's 130 KB) waits there for as long as the flood of tiles lasts (measured: a 75 us panel stage took 250 us), one the size
// of a GEMM tile's workgroup takes the place of the next tile that retires.  Workgroup (slab, problem):
//   gate   thread 0 waits for the tail's progress word (bounded), everybody takes the acquire fence;
//   T      the panel's T factor, by every workgroup for itself (64 x 64 Gram of the panel on MFMA, ~8 us at 250 rows, 2 us for
//          the last panel; the blocked inverse of wy_prep.hpp on four waves) -- no launch and no dependence between workgroups;
//          slab 0 also stores it (stage 4 of a chain with an eigenvector-form consumer reads it);
//   apply  the slab: rows j >= 64 p of Zt from Q (panel 0: the identity), W1 = V_p Zc, W2 = T_p^T W1, Zc -= V_p^T W2, back to Q.
constexpr int WQ_NT = 256;
constexpr int WQ_KC = 64, WQ_LDV = WQ_KC + 2;
// LDS (doubles): phase T: chunk / G [64][66], T [64][65], products [32][33], tau [64];  phase apply (overlaid): Zs [n][18], W1, W2 [64][18]
inline size_t wq_lds_bytes(int n) {
    const size_t t_phase = (size_t)WY_NB * WQ_LDV + (size_t)WY_NB * WY_LDG + 32 * WY_LDP + WY_NB;
    const size_t a_phase = (size_t)n * WY_LD + 2 * (size_t)WY_NB * WY_LD;
    return std::max(t_phase, a_phase) * sizeof(double);
}

__global__ __launch_bounds__(WQ_NT) void wy_qstage_kernel(WyBatch b, int p0, int np) {
    const WyProb P = wy_resolve(b, blockIdx.y);
    const int n = P.n;
    const int r0 = blockIdx.x * WY_ZC;                 // rows r0 .. r0 + 15 of Q = this slab's columns of Zt
    if (r0 >= n) return;
    // np > 1 (the unpipelined form: every panel in ONE launch -- behind the tail the four launches of the pipelined form cost a
    // launch's scheduling under the main stream's products each, 250 us for 4 x 20): the slab travels through Q between panels,
    // written and read by this workgroup only
    for (int p = p0; p < p0 + np && p < P.npanels; ++p) {
    if (p > p0) {
        __threadfence_block();
        __syncthreads();
    }
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // (GPCSD_QPIPE_CLK=1: phase clocks of workgroup (0, 0) of the last two panels' launches, clk[48 + 8 (p & 1) + k]; the gates' stamps fill [0, 48): three per (panel, problem))
    int nst = 0;
    auto stamp = [&]() {
        if (b.clk && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nst < 8) b.clk[48 + 8 * (p & 1) + nst] = wall_clock64();
        ++nst;
    };
    // ---- gate
    int gate_failed = 0;
    if (tid == 0) {
        const unsigned target = (p + 1 < P.npanels) ? (unsigned)(WY_NB * (p + 1)) : WY_PROG_DONE;
        const unsigned *prog = reinterpret_cast<const unsigned *>(P.tau + n + WY_NB);
        const unsigned long long t0 = wall_clock64();
        const bool stamping = b.clk && blockIdx.x == 0;
        if (stamping) b.clk[3 * (p * 4 + blockIdx.y)] = t0;              // (GPCSD_QPIPE_CLK=1: when the gate started, passed, and the tail's start)
        bool ok = b.gate_ticks != 0;                                     // (0: the test aid -- every gate gives up at once)
        while (ok && __hip_atomic_load(prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > b.gate_ticks) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        // The tail is not running beside this launch (kernels serialised by a profiler or a debugger, an oversubscribed card): not a
        // numerical failure.  Report 7 and do NOTHING with the unfinished reflectors; the call that collects this evaluation runs
        // it again behind the end of the tail (capi_fused.inl: q_pipe_missed).
        if (!ok && b.status) atomicMax(b.status, 7);
        gate_failed = !ok;
        if (stamping) {
            b.clk[3 * (p * 4 + blockIdx.y) + 1] = wall_clock64();
            b.clk[3 * (p * 4 + blockIdx.y) + 2] = *reinterpret_cast<const unsigned long long *>(P.tau + n + WY_NB + 1);
        }
    }
    // thread 0's verdict through the first word of the (not yet used) dynamic LDS: a static flag or __syncthreads_or's own would
    // take static LDS out of the 160 KB the launch asks for as dynamic
    if (tid == 0) reinterpret_cast<volatile int *>(smem)[0] = gate_failed;
    __syncthreads();
    gate_failed = reinterpret_cast<volatile int *>(smem)[0];
    if (gate_failed) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // every wave reads the panel the tail's workgroup has just released
    __builtin_amdgcn_s_setprio(3);                        // (on the log-likelihood's critical path, beside a flood of GEMM tiles)
    stamp();
    const double *__restrict__ Vp = P.V + (long)p * WY_NB * n;
    const int kstart = (p * WY_NB) & ~3;                  // reflector k is zero up to column k
    // ---- T factor (wy_prep.hpp's scheme on four waves): G = V_p V_p^T, upper blocks only
    double *vs = smem, *g = smem, *tl = smem + WY_NB * WQ_LDV, *pl = tl + WY_NB * WY_LDG, *st = pl + 32 * WY_LDP;
    constexpr int LDG = WY_LDG;
    {
        const int bi = (wid >> 1) & 1, bj = wid & 1;      // wave 2 would own the lower block: nobody reads it
        const bool upper = bi <= bj;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        for (int kc = kstart; kc < n; kc += WQ_KC) {
            const int kn = min(WQ_KC, (n - kc + 3) & ~3);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < WY_NB * WQ_KC / WQ_NT; ++u) {
                const int idx = tid + WQ_NT * u;
                const int r = idx / WQ_KC, k = idx % WQ_KC;
                if (k < kn) vs[r * WQ_LDV + k] = (kc + k < n) ? Vp[(long)r * n + kc + k] : 0.0;
            }
            __syncthreads();
            if (upper) {
                const double *__restrict__ va = vs + (32 * bi + fr) * WQ_LDV + fq, *__restrict__ vb = vs + (32 * bj + fr) * WQ_LDV + fq;
#pragma unroll 4
                for (int k0 = 0; k0 < kn; k0 += 4) {
                    const double a0 = va[k0], a1 = va[16 * WQ_LDV + k0], b0 = vb[k0], b1 = vb[16 * WQ_LDV + k0];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                   // G lands where the chunk was
        for (int idx = tid; idx < WY_NB * LDG; idx += WQ_NT) tl[idx] = 0.0;
        if (upper) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[(32 * bi + 16 * i + fq + 4 * r) * LDG + 32 * bj + 16 * j + fr] = acc[i][j][r];
        }
        if (tid < WY_NB) {
            const int kk = p * WY_NB + tid;
            st[tid] = (kk < P.nrefl) ? P.tau[kk] : 0.0;
        }
        __syncthreads();
    }
    stamp();
    // diagonal 16 x 16 blocks of T = (diag(1 / tau) + striu(G))^-1: wave w owns block w, LANE c its column c -- sixteen entries in
    // registers, x_c = tau_c, x_j = -tau_j sum_{l > j} G_jl x_l going up (entries below the diagonal stay zero, so the sums need no
    // masks); the G entries are broadcast LDS reads at compile-time offsets.  (The axpy form of wy_prep.hpp, a lane per ROW and
    // forty dependent lane-read steps, was 8 of the 30 us the last panel's launch took.)
    {
        const int b0 = 16 * wid, cidx = lane & 15;
        const double *__restrict__ gb = g + b0 * LDG + b0;
        double x[16];
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            double sum = 0.0;
#pragma unroll
            for (int l = j + 1; l < 16; ++l) sum = fma(gb[j * LDG + l], x[l], sum);
            const double tj = st[b0 + j];
            x[j] = (j == cidx) ? tj : ((j < cidx) ? -tj * sum : 0.0);
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) tl[(b0 + j) * LDG + b0 + cidx] = x[j];
        }
    }
    __syncthreads();
    stamp();
    // off-diagonal blocks by doubling: T_ab = -T_aa (G_ab T_bb) at block size 16, then 32
    auto frag = [&](const double *A, int lda, const double *B, int ldb, int K) {      // A[fr][k] B[k][fr] over k < K
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[fr * lda + k0 + fq], B[(k0 + fq) * ldb + fr], acc, 0, 0, 0);
        return acc;
    };
    if (wid < 2) {
        const int a = 32 * wid, bb = a + 16;
        const d4 pr = frag(g + a * LDG + bb, LDG, tl + bb * LDG + bb, LDG, 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[(16 * wid + fq + 4 * r) * WY_LDP + fr] = pr[r];
    }
    __syncthreads();
    if (wid < 2) {
        const int a = 32 * wid, bb = a + 16;
        const d4 tr = frag(tl + a * LDG + a, LDG, pl + 16 * wid * WY_LDP, WY_LDP, 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[(a + fq + 4 * r) * LDG + bb + fr] = -tr[r];
    }
    __syncthreads();
    {
        const int fi = wid >> 1, fj = wid & 1;
        const d4 pr = frag(g + (16 * fi) * LDG + 32, LDG, tl + 32 * LDG + 32 + 16 * fj, LDG, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[(16 * fi + fq + 4 * r) * WY_LDP + 16 * fj + fr] = pr[r];
    }
    __syncthreads();
    {
        const int fi = wid >> 1, fj = wid & 1;
        const d4 tr = frag(tl + (16 * fi) * LDG, LDG, pl + 16 * fj, WY_LDP, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[(16 * fi + fq + 4 * r) * LDG + 32 + 16 * fj + fr] = -tr[r];
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        double *__restrict__ Tg = P.T + (long)p * WY_NB * WY_NB;
#pragma unroll
        for (int u = 0; u < WY_NB * WY_NB / WQ_NT; ++u) {
            const int idx = tid + WQ_NT * u;
            Tg[idx] = tl[(idx >> 6) * LDG + (idx & 63)];
        }
    }
    // this wave's operand of W2 = T_p^T W1 (rows 16 wid .. of T^T = columns of T) into registers: the LDS is reused below
    double ta[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) ta[u] = tl[(4 * u + fq) * LDG + 16 * wid + fr];
    __syncthreads();
    stamp();
    // ---- apply
    double *Zs = smem;                         // [n][WY_LD]
    double *W1 = Zs + (size_t)n * WY_LD;       // [64][WY_LD]
    double *W2 = W1 + WY_NB * WY_LD;           // [64][WY_LD]
    const int j0 = (p * WY_NB) & ~15;          // the panel's reflectors are zero in front of column 64 p + 1
    for (int j = j0 + tid; j < n; j += WQ_NT) {
#pragma unroll
        for (int rr = 0; rr < WY_ZC; ++rr) {
            double v;
            if (p == 0) v = (j == r0 + rr) ? 1.0 : 0.0;
            else v = (r0 + rr < n) ? P.Z[(long)(r0 + rr) * n + j] : 0.0;
            Zs[j * WY_LD + rr] = v;
        }
    }
    __syncthreads();
    stamp();
    {   // W1 = V_p Zc: wave w owns panel rows 16 w .. 16 w + 15 over the whole K range
        const double *__restrict__ ra = Vp + (long)(16 * wid + fr) * n;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = kstart; k0 < n; k0 += 32) {
            double a8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 4 * u + fq;
                a8[u] = ra[k < n ? k : n - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 4 * u + fq;
                const double bb = Zs[(k < n ? k : n - 1) * WY_LD + fr];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < n ? a8[u] : 0.0, bb, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) W1[(16 * wid + fq + 4 * r) * WY_LD + fr] = acc[r];
    }
    lds_barrier();
    {   // W2 = T_p^T W1
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[u], W1[(4 * u + fq) * WY_LD + fr], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) W2[(16 * wid + fq + 4 * r) * WY_LD + fr] = acc[r];
    }
    lds_barrier();
    stamp();
    {   // Zc -= V_p^T W2, rows from the panel's first reflector on
        const int nfrag = (n + 15) / 16;
        for (int fm = (p * WY_NB) / 16 + wid; fm < nfrag; fm += WQ_NT / 64) {
            const int m = 16 * fm + fr;
            const double *__restrict__ vm = Vp + (m < n ? m : n - 1) + (long)fq * n;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vm[(long)(4 * u) * n], W2[(4 * u + fq) * WY_LD + fr], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * fm + fq + 4 * r;
                if (row < n) Zs[row * WY_LD + fr] -= acc[r];
            }
        }
    }
    __syncthreads();
    stamp();
    for (int j = j0 + tid; j < n; j += WQ_NT) {
#pragma unroll
        for (int rr = 0; rr < WY_ZC; ++rr)
            if (r0 + rr < n) P.Z[(long)(r0 + rr) * n + j] = Zs[j * WY_LD + rr];
    }
    stamp();
    }
}

static bool wy_clk_on() {
    static const bool on = getenv("GPCSD_WY_CLK") && getenv("GPCSD_WY_CLK")[0] == '1';
    return on;
}
static void wy_clk_print(gpcsd_ctx *c, unsigned long long *d, const char *what, int first, int n, hipStream_t s) {
    unsigned long long h[32];
    c->copy_out(h, d, sizeof(h), s);
    GP_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[%s] phases (10 ns ticks):", what);
    for (int i = first + 1; i < first + n; ++i) fprintf(stderr, " %llu", h[i] >= h[i - 1] ? h[i] - h[i - 1] : 0ull);
    fprintf(stderr, "\n");
}

static void wy_prep_launch(const WyBatch &b, int maxP, int count, hipStream_t s) {
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)WY_PREP_LDS));
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
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_apply_kernel<WY_KS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_apply_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    }
    const WyBatch w = wy_with_clk(c, b, s);
    if (!prep_done) wy_prep_launch(w, maxP, count, s);
    if (split) hipLaunchKernelGGL(wy_apply_kernel<WY_KS>, dim3(ceil_div(nmax, WY_ZC), count), dim3(WY_NT), sh, s, w);
    else hipLaunchKernelGGL(wy_apply_kernel<1>, dim3(ceil_div(nmax, WY_ZC), count), dim3(WY_NT), sh, s, w);
    if (w.clk) wy_clk_print(c, w.clk, "wy_apply, last column block: init, then per panel W1 | W2 + loads | update, store", 8, 2 + 3 * maxP + 1, s);
    GP_HIP(hipGetLastError());
}

// Stage 5 (see above): for every panel the gated T factor, the forward apply, then whatever the caller hangs on the columns the
// panel completes -- chunk(class i: first and one-past-last final column of this stage; empty ranges for a class that has no
// such panel).  Z of the batch = the Q buffers (row-major).
void wy_q_pipeline(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s,
                   const std::function<void(const int *col0, const int *col1)> &chunk, bool one_launch) {
    int maxP = 0, nmax = 0;
    for (int i = 0; i < nclass; ++i) {
        maxP = std::max(maxP, b.p[i].npanels);
        nmax = std::max(nmax, b.p[i].n);
    }
    if (maxP == 0) return;
    const int count = b.start[MAX_EIG_BATCH];
    const size_t sh = wq_lds_bytes(nmax);
    GP_REQUIRE(sh <= 160 * 1024, -3, "wy_q_pipeline: %d rows do not fit the stage kernel", nmax);
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wy_qstage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    WyBatch bc = b;
    // behind the END of the tail (one_launch) a gate finds its word at "done" whatever its patience; beside a running tail it is
    // gpcsd_ctx::q_gate_ticks (GPCSD_QPIPE_GATE_TICKS=0: every gate gives up -- tests/test_q_pipeline.py drives the replay with it)
    bc.gate_ticks = one_launch ? 20000000ull : c->q_gate_ticks;
    static const bool gate_clk = getenv("GPCSD_QPIPE_CLK") && getenv("GPCSD_QPIPE_CLK")[0] == '1';
    if (gate_clk) bc.clk = c->buf<unsigned long long>("wy_clk", 64);        // (tools/qpipe_probe.py reads it)
    if (one_launch) {                      // unpipelined (behind the tail): every panel in one launch, nothing hangs on the panels
        hipLaunchKernelGGL(wy_qstage_kernel, dim3(ceil_div(nmax, WY_ZC), count), dim3(WQ_NT), sh, s, bc, 0, maxP);
        GP_HIP(hipGetLastError());
        return;
    }
    for (int p = 0; p < maxP; ++p) {
        hipLaunchKernelGGL(wy_qstage_kernel, dim3(ceil_div(nmax, WY_ZC), count), dim3(WQ_NT), sh, s, bc, p, 1);
        GP_HIP(hipGetLastError());
        int col0[MAX_EIG_BATCH], col1[MAX_EIG_BATCH];
        for (int i = 0; i < nclass; ++i) {
            const int P = b.p[i].npanels;
            col0[i] = col1[i] = 0;
            if (p < P) {
                col0[i] = WY_NB * p;
                col1[i] = (p + 1 < P) ? WY_NB * (p + 1) : b.p[i].n;
            }
        }
        if (chunk) chunk(col0, col1);
    }
}

}  // namespace gpcsd
