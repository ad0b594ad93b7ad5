// Register-resident tail of the Householder tridiagonalisation (included by eigh_dc.hip after SytrdBatch).
//
// The trailing T <= 256 rows of every problem are reduced by ONE workgroup of 768 threads in one launch.
//
// Block (the last <= 192 rows/columns, in VGPRs): thread (row group g, column part h) keeps the 4 x 12 tile rows 4g..4g+3,
// columns 12h..12h+11 (96 VGPRs).  A wave is four row groups x sixteen column parts, i.e. one 16-lane DPP row per row
// group, so the row sums of y = A v fold with DPP moves and never touch LDS.  Per column:
//   gen   the wave that owns row kk reads that row (= column kk by symmetry) out of its registers, forms the
//         Householder vector and publishes v -- a wave-local section that overlaps the other waves' rank-2 update of
//         the previous column;
//   A     barrier;  y = A v from registers (48 FMAs + a DPP reduce-scatter per thread), v.y partials;
//   B     barrier;  w = tau y - cc v formed on the fly, rank-2 update of the tile (96 FMAs per thread).
// Two barriers per column instead of a dependent launch (4.7 us + 1.8 us graph node gap); dead waves skip the FMA phases,
// dead rows / columns need no masks in the FMA loops: v and y are zero there.
//
// Strip (T > 192: the S = T - 192 leading rows, full length, in LDS -- up to 64 x 258 doubles): the same two-barrier column
// with the strip rows dealt round-robin to the waves: row r belongs to wave r mod 12, lane l holds columns l, l+64, ..
// The wave owning row k generates reflector k straight from LDS (it wrote that row itself in the previous update: LDS is
// in order per wave, no extra barrier); strip rows add their dot products to y, their columns add to the row sums of the
// block (folded into the same DPP reduction), and the update is a read-modify-write of the wave's own rows.  This replaces
// 58 per-column launches of the 250-row temporal half-problems (0.38 ms of a 1.0 ms chain) by work inside the launch.
//
// Indices: "tail-global" g in [0, T) is matrix row/column k0 + g; block-local index i is g = S + i.  The LDS vectors
// (v, y, d, e, tau) keep the block at the FIXED slot RT_SMAX + i and the strip in front of it (slot OFF + g with
// OFF = RT_SMAX - S), so the block columns address them with compile-time offsets whatever S is.
#pragma once
#include <type_traits>

namespace gpcsd {

constexpr int RT_R = 4, RT_C = 12, RT_T = 192, RT_NTH = (RT_T / RT_R) * (RT_T / RT_C), RT_NW = RT_NTH / 64;
constexpr int RT_SMAX = 64, RT_TMAX = RT_T + RT_SMAX;     // strip rows, largest tail
constexpr unsigned RT_PROG_DONE = 0x7fffffffu;            // progress word of a finished launch (see sytrd_rtail_kernel)
static_assert(RT_T / RT_C == 16 && RT_NTH == 768, "one DPP row per row group");

// strip rows are LDT doubles apart: LDT / 2 odd, so that sixteen consecutive rows start in distinct 16-byte bank groups
// (the column sums read a b128 per lane from sixteen different rows at once)
__host__ __device__ inline int rt_strip_rows(int T) { return T > RT_T ? ((T - RT_T + 1) & ~1) : 0; }
__host__ __device__ inline int rt_strip_ld(int T) { return ((T + 1) & ~1) | 2; }
inline size_t rt_strip_bytes(int T) { return ((size_t)rt_strip_rows(T) * rt_strip_ld(T) + 8) * sizeof(double); }

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

// a store that is visible to the other XCDs once it has completed (relaxed, agent scope: written through this XCD's L2)
__device__ __forceinline__ void rt_store_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a reflector entry: written through only when somebody reads it while this launch is still running (pipe, uniform)
__device__ __forceinline__ void rt_store_v(bool pipe, double *p, double v) {
    if (pipe) rt_store_agent(p, v);
    else *p = v;
}

// Householder scalars of a column with pivot alpha and squared norm xnorm2 below it: H = I - tau u u^T with
// u = (alpha - beta, x_2, ..) left UN-normalised: u is known as soon as s = sqrt(alpha^2 + |x|^2) is, and
// tau = 1 / (s (|alpha| + s)) = r / |u_1| is formed by every wave after the barrier, off the generating wave's chain.
// s and r = 1/s come out of one coupled Newton (Goldschmidt) iteration on the hardware rsq seed (6 dependent operations
// instead of the ~25 of an IEEE sqrt and two divisions).  The matrix is scaled to max|a| = 1 and tau grows like 1 / s^2, so
// a column with s < 1e-50 is treated as zero (H = I: r = 0; the entries dropped are 1e-34 of an ulp of the matrix) -- an
// exactly low-rank input (e.g. all ones) leaves columns that are rounding noise of rounding noise, shrinking by 1e-15 per
// column, and 1 / s^2 overflowed on the eighth of them.  `ok` false forces H = I (last column).
__device__ __forceinline__ void rt_house(double alpha, double xnorm2, bool ok, double &r, double &u1, double &beta) {
    const double s2 = fma(alpha, alpha, xnorm2);
    r = 0.0;
    u1 = 1.0;
    beta = alpha;
    if (ok && xnorm2 > 0.0 && s2 > 1e-100) {                         // wave-uniform
        const double y0 = __builtin_amdgcn_rsq(s2);
        double g = s2 * y0, hh = 0.5 * y0;
        double e = fma(-hh, g, 0.5);
        g = fma(g, e, g);
        hh = fma(hh, e, hh);
        e = fma(-hh, g, 0.5);
        g = fma(g, e, g);                                            // sqrt(s2)
        hh = fma(hh, e, hh);                                         // 1 / (2 sqrt(s2))
        beta = -copysign(g, alpha);
        u1 = alpha - beta;                                           // sign(alpha) (|alpha| + s): no cancellation
        r = hh + hh;
    }
}

// Early exit for positive semi-definite inputs (SytrdProb::psd: the Gram matrices of the fused calls).  A Householder step is a
// similarity transformation, so the block still to be reduced stays PSD and its trace -- trace(A) minus the diagonal entries
// already produced, one scalar kept by the generating wave -- bounds its norm.  Once that trace has dropped below 64 unit roundoffs
// of trace(A) the rest of the block is rounding noise of the part already reduced: the remaining reflectors are left out (H = I:
// tau = 0) and the tridiagonal ends with the block's diagonal.  Backward error <= the dropped trace, inside the n eps ||A||
// the reduction itself commits.  The Gram matrices of smooth kernels are numerically low-rank: the 192-row halves of the
// 384-channel Ks have ~58 eigenvalues above 1e-13 of the largest, and two thirds of the serial column steps of their tails
// multiplied noise.  gpcsd_tail_early_exit(ctx, 0) / GPCSD_TAIL_EARLY_EXIT=0 keep every column step (the host then never sets
// SytrdProb::psd): the A/B and the cross-check of tests/test_tail_early_exit.py.

__global__ __launch_bounds__(RT_NTH) void sytrd_rtail_kernel(SytrdBatch b) {
    const SytrdProb P = sy_resolve(b, blockIdx.x);
    const int n = P.n, k0 = P.k_tail;
    if (k0 >= n - 1) return;
    if (b.clk && threadIdx.x == 0) b.clk[2 * blockIdx.x] = wall_clock64();      // (measurement only: see SytrdBatch::clk)
    const int T = n - k0;                          // rows / columns k0 .. n-1, T <= RT_TMAX
    const int S = rt_strip_rows(T), LDT = rt_strip_ld(T);
    const int TB = T - S;                          // live rows of the register block, <= RT_T
    const int OFF = RT_SMAX - S;                   // slot of tail-global index 0 in the LDS vectors
    extern __shared__ __attribute__((aligned(16))) double strip[];   // [S][LDT]
    __shared__ __attribute__((aligned(16))) double sx[RT_T], sv2[2][RT_TMAX], sy[RT_TMAX];
    __shared__ __attribute__((aligned(16))) double red[RT_NW];
    __shared__ double sd[RT_TMAX], se[RT_TMAX], st[RT_TMAX];
    __shared__ double s_r, s_u1;                   // 1 / ||column|| (0: H = I) and |u_1| of the current reflector
    __shared__ double s_trace, s_ttol;             // early exit (below): trace still to be reduced, and the threshold
    __shared__ int s_exit_at;                      // first block column left out (-1: none)
    // v is double-buffered: the wave generating reflector kk+1 writes it while slower waves still read v of column kk
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gg = lane >> 4, h = lane & 15;       // row group inside the wave, column part
    const int row0 = 16 * wid + 4 * gg, c0 = RT_C * h;
    const double *__restrict__ Ain = (k0 & 1) ? P.A1 : P.A0;
    const double *__restrict__ yin = (k0 & 1) ? P.y1 : P.y0;
    // Progress word (SytrdProb::pipe, whole problems only): the number of leading reflectors whose rows of V and whose tau are
    // complete in global memory, published at every multiple of 64 -- one compact-WY panel -- and RT_PROG_DONE behind the last
    // store of the launch.  The consumers of a staged chain (wy.hip: the T factor of a panel, the columns of Q and of X = Y~ Q the
    // panel completes) start on their panel while this workgroup is still reducing the next one.
    unsigned *const prog = sy_progress_word(P);
    const bool pipe = P.pipe != 0 && k0 == 0;
    if (pipe && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(prog)[1] = wall_clock64();      // (measurement: GPCSD_QPIPE_CLK)
    int published = 0;
    // (uniform; called behind a barrier that follows the last column's tau in LDS.)  Every thread fences its own fire-and-forget
    // stores of reflector rows, the barrier collects the fences, one thread releases the word.
    // (The rows of V and tau leave as agent-scope stores -- written through the XCD's L2 -- so that the release needs no write-back
    // of that L2 (a __threadfence here writes back whatever the GEMM tiles of the other streams left dirty in it: the tail of a
    // pipelined cfg3 step measured 35 us longer); every thread waits for its own stores, the barrier collects the waits.)
    // The word itself is published by a RELAXED agent-scope store behind `s_waitcnt vmcnt(0)` + the barrier: that is a release only
    // where stores are counted in vmcnt and agent-scope (sc1) stores write through the L2 -- gfx942 / gfx950.  A target with a
    // separate store counter (vscnt) needs __ATOMIC_RELEASE here (measured on gfx950: +30 us per tail, the fence writes the XCD's
    // whole L2 back under the other streams' GEMM traffic).  build.py compiles for gfx950 only; anything else must not get this far:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "sytrd_regtail.hpp: the progress word's release relies on gfx942 / gfx950 memory-counter semantics (see publish())"
#endif
    auto publish = [&](const int done) {
        const int upto = done & ~63;
        if (!pipe || upto <= published) return;
        for (int g = published + tid; g < upto; g += RT_NTH) rt_store_agent(P.tau + g, st[OFF + g]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(prog, (unsigned)upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = upto;
    };

    // pending rank-2 update of step k0-1 (its reflector and y = A v are in global memory): v -> sv2[0], w -> sy
    {
        double pv = 0.0, py = 0.0, taup = 0.0;
        const int gslot = tid - OFF;                                 // thread tid fills slot tid
        if (k0 > 0) {
            taup = P.tau[k0 - 1];
            if (gslot >= 0 && gslot < T) {
                pv = P.V[(long)(k0 - 1) * n + k0 + gslot];
                py = yin[k0 + gslot];
            }
        }
        const double part = wave_sum(pv * py);
        if (lane == 0) red[wid] = part;
        __syncthreads();
        double dot = 0.0;
#pragma unroll
        for (int q = 0; q < RT_NW; ++q) dot += red[q];
        const double cc = 0.5 * taup * taup * dot;
        if (tid < RT_TMAX) {
            sv2[0][tid] = pv;
            sv2[1][tid] = 0.0;                                       // the strip columns never write the padding slots
            sy[tid] = taup * py - cc * pv;
        }
        __syncthreads();
    }
    double a[RT_R][RT_C];
    {
        const double *svp = sv2[0] + RT_SMAX, *syp = sy + RT_SMAX;
#pragma unroll
        for (int r = 0; r < RT_R; ++r) {
            const int i = row0 + r;
            const bool rok = i < TB;
            const double *__restrict__ arow = Ain + (long)(k0 + S + (rok ? i : 0)) * n + k0 + S;
            const double vi = rok ? svp[i] : 0.0, wi = rok ? syp[i] : 0.0;
#pragma unroll
            for (int j = 0; j < RT_C; ++j) {
                const int c = c0 + j;
                const bool ok = rok && c < TB;
                const double g = ok ? arow[c] : 0.0;
                a[r][j] = ok ? g - vi * syp[c] - wi * svp[c] : 0.0;
            }
        }
        // the pad columns (c >= T) are read by the column sums of padding rows: they must hold zeros, not stale LDS
        for (int idx = tid; idx < S * LDT + 8; idx += RT_NTH) {
            const int r = idx / LDT, c = idx - r * LDT;
            double v = 0.0;
            if (r < S && c < T)
                v = Ain[(long)(k0 + r) * n + k0 + c] - sv2[0][OFF + r] * sy[OFF + c] - sy[OFF + r] * sv2[0][OFF + c];
            strip[idx] = v;
        }
    }
    {
        // trace of the register block as loaded (one entry per thread at most lies on the diagonal... up to RT_R of them)
        double tr = 0.0;
#pragma unroll
        for (int r = 0; r < RT_R; ++r)
#pragma unroll
            for (int j = 0; j < RT_C; ++j) tr += (row0 + r == c0 + j) ? a[r][j] : 0.0;
        tr = wave_sum(tr);
        if (lane == 0) red[wid] = tr;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < RT_NW; ++q) tot += red[q];
        s_trace = tot;
        // (strip rows in front of the block, S > 0: their diagonal is not counted -- the check then only starts once the
        // strip is done, with the block's own trace at that point taken below)
        // 64 unit roundoffs of the trace: what the running difference itself can be off by after a few hundred subtractions, and
        // a dropped PSD block moves no eigenvalue by more than its trace (Weyl) -- the few-eps-||A|| any solver leaves in the
        // small eigenvalues.  (4 n eps, the reduction's own backward-error scale, showed in a log-likelihood: 1D models multiply
        // the temporal spectrum by spatial eigenvalues of 1e8, and 1.2e-9 of it moved.)
        s_ttol = (P.psd && S == 0) ? 64.0 * EPS_U * tot : -1.0;
        s_exit_at = -1;
    }
    if (tid < RT_TMAX) sy[tid] = 0.0;              // from here on sy is y = A v: zero on dead rows and beyond T
    __syncthreads();

    // y = (block) v row sums over the 16 lanes of a row group as a reduce-scatter: each exchange halves the number of sums a
    // lane still carries (2 + 1 DPP adds), two rotations finish them.  Lane h ends with the sum of row 2*(h&1) + ((h>>1)&1)
    // of its group, replicated in its 4 quads.
    auto reduce_rows = [&](const double (&acc)[RT_R]) {
        const bool b0 = h & 1, b1 = h & 2;
        const double t0 = (b0 ? acc[2] : acc[0]) + dpp_mov<0xB1>(b0 ? acc[0] : acc[2]);
        const double t1 = (b0 ? acc[3] : acc[1]) + dpp_mov<0xB1>(b0 ? acc[1] : acc[3]);
        double y = (b1 ? t1 : t0) + dpp_mov<0x4E>(b1 ? t0 : t1);
        y += dpp_mov<0x124>(y);                                      // row_ror:4
        y += dpp_mov<0x128>(y);                                      // row_ror:8
        return y;
    };
    const int myrow = row0 + 2 * (h & 1) + ((h >> 1) & 1);           // block row whose sum this lane ends up with

    // ------------------------------------------------------------------------------------------------------------
    // strip columns k = 0 .. S-1: every block row is live, the matrix is strip rows (k, S) + the block
    // ------------------------------------------------------------------------------------------------------------
    // strip rows: lane l covers the tail-global columns (2l, 2l + 1) and (128 + 2l, 129 + 2l), one 16-byte LDS access each
    const int cA = 2 * lane, cB = 128 + 2 * lane;
    const bool okB = cB < T;                                         // (cA < 128 < T whenever there is a strip)
    for (int k = 0; k < S; ++k) {
        double *sv = sv2[k & 1] + OFF;                               // tail-global view of the vectors in this phase
        double *syg = sy + OFF;
        if (wid == k % RT_NW) {                                      // gen: this wave wrote strip row k itself
            __builtin_amdgcn_s_setprio(3);
            const double *__restrict__ row = strip + k * LDT;
            double x[4], part = 0.0;
            {
                const double2 xa = *reinterpret_cast<const double2 *>(row + cA);          // cA < 128 <= T always
                const double2 xb = okB ? *reinterpret_cast<const double2 *>(row + cB) : double2{0.0, 0.0};
                x[0] = xa.x; x[1] = xa.y; x[2] = xb.x; x[3] = xb.y;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = ((q & 2) ? cB : cA) + (q & 1);
                const double m = (c >= k + 2) ? x[q] : 0.0;
                part = fma(m, m, part);
            }
            const double xnorm2 = wave_sum(part);
            const double dk = row[k], alpha = row[k + 1];
            double r, u1, beta;
            rt_house(alpha, xnorm2, true, r, u1, beta);
            double *__restrict__ vrow = P.V + (long)(k0 + k) * n + k0;      // reflector k (zeros up to k, u_1 at k+1)
            double tv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = ((q & 2) ? cB : cA) + (q & 1);
                double t = (c > k + 1) ? x[q] : 0.0;                       // (a pad column c == T reads 0 and stays 0)
                t = (c == k + 1) ? u1 : t;
                tv[q] = (r != 0.0 || c == k + 1) ? t : 0.0;
                if (c < T) rt_store_v(pipe, vrow + c, tv[q]);              // fire and forget: the barriers wait for LDS only
            }
            *reinterpret_cast<double2 *>(sv + cA) = double2{tv[0], tv[1]};
            if (okB) *reinterpret_cast<double2 *>(sv + cB) = double2{tv[2], tv[3]};
            if (lane == 0) {
                sd[OFF + k] = dk;
                se[OFF + k] = beta;
                s_r = r;
                s_u1 = fabs(u1);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();                                             // ---- A: v, scalars published
        const double rr = s_r, au = s_u1;
        const double tau = rr * fast_rcp(au);
        if (tid == 0) {
            st[OFF + k] = tau;
            syg[k] = 0.0;                                            // row k is dead from now on (nobody reads y before B)
        }
        const bool live = rr != 0.0;                                 // uniform over the workgroup
        const int rfirst = k + 1 + (wid - (k + 1) % RT_NW + RT_NW) % RT_NW;   // this wave's first strip row > k
        if (live) {
            // strip rows of this wave: y_r = strip[r][:] . v
            double dps = 0.0;
            const double2 va = *reinterpret_cast<const double2 *>(sv + cA);
            const double2 vb = okB ? *reinterpret_cast<const double2 *>(sv + cB) : double2{0.0, 0.0};
            for (int r = rfirst; r < S; r += RT_NW) {
                const double *__restrict__ row = strip + r * LDT;
                const double2 ra = *reinterpret_cast<const double2 *>(row + cA);
                const double2 rb = okB ? *reinterpret_cast<const double2 *>(row + cB) : double2{0.0, 0.0};
                const double p = fma(rb.y, vb.y, fma(rb.x, vb.x, fma(ra.y, va.y, ra.x * va.x)));
                const double yr = wave_sum(p);
                if (lane == 0) syg[r] = yr;
                dps = fma(sv[r], yr, dps);
            }
            // block rows: the tile from registers, plus the strip COLUMNS of these rows (strip row rs, lanes h = rs mod 16)
            const double vmy = sv[S + myrow];
            double acc[RT_R] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) {
                const double2 vv = *reinterpret_cast<const double2 *>(sv + S + c0 + j);
#pragma unroll
                for (int r = 0; r < RT_R; ++r) acc[r] = fma(a[r][j + 1], vv.y, fma(a[r][j], vv.x, acc[r]));
            }
            for (int rs = k + 1 + (h - (k + 1) % 16 + 16) % 16; rs < S; rs += 16) {
                const double vr = sv[rs];
                const double2 s01 = *reinterpret_cast<const double2 *>(strip + rs * LDT + S + row0);
                const double2 s23 = *reinterpret_cast<const double2 *>(strip + rs * LDT + S + row0 + 2);
                acc[0] = fma(s01.x, vr, acc[0]);
                acc[1] = fma(s01.y, vr, acc[1]);
                acc[2] = fma(s23.x, vr, acc[2]);
                acc[3] = fma(s23.y, vr, acc[3]);
            }
            const double y = reduce_rows(acc);
            if (h < 4) syg[S + myrow] = y;
            double dp = (h < 4) ? vmy * y : 0.0;
            dp += dpp_mov<0xB1>(dp);
            dp += dpp_mov<0x4E>(dp);
            dp = (lane_get(dp, 0) + lane_get(dp, 16)) + (lane_get(dp, 32) + lane_get(dp, 48));
            if (lane == 0) red[wid] = dp + dps;
        } else if (lane == 0) {
            red[wid] = 0.0;
        }
        const double htt = 0.5 * tau * tau;
        __syncthreads();                                             // ---- B: y, v.y partials published
        if (live) {
            double pr[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const double2 t2 = *reinterpret_cast<const double2 *>(red + 2 * q);
                pr[q] = t2.x + t2.y;
            }
            const double dot = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + (pr[4] + pr[5]);
            const double cc = htt * dot;
            // strip rows of this wave (row k+1 first when it is ours: its owner generates the next reflector from it)
            if (rfirst < S) {
                // two adjacent columns per 16-byte LDS access (half the LDS instructions of one column per access); a pad
                // column (index T when T is odd) has v = w = 0 and is written back unchanged
                const double2 va = *reinterpret_cast<const double2 *>(sv + cA);
                const double2 vb = okB ? *reinterpret_cast<const double2 *>(sv + cB) : double2{0.0, 0.0};
                const double2 ya = *reinterpret_cast<const double2 *>(syg + cA);
                const double2 yb = okB ? *reinterpret_cast<const double2 *>(syg + cB) : double2{0.0, 0.0};
                const double wa0 = tau * ya.x - cc * va.x, wa1 = (cA + 1 < T) ? tau * ya.y - cc * va.y : 0.0;
                const double wb0 = okB ? tau * yb.x - cc * vb.x : 0.0, wb1 = (cB + 1 < T) ? tau * yb.y - cc * vb.y : 0.0;
                for (int r = rfirst; r < S; r += RT_NW) {
                    double *__restrict__ row = strip + r * LDT;
                    const double vr = sv[r], wr = tau * syg[r] - cc * vr;
                    double2 ra = *reinterpret_cast<const double2 *>(row + cA);
                    ra.x = fma(-wr, va.x, fma(-vr, wa0, ra.x));
                    ra.y = fma(-wr, va.y, fma(-vr, wa1, ra.y));
                    *reinterpret_cast<double2 *>(row + cA) = ra;
                    if (okB) {
                        double2 rb = *reinterpret_cast<const double2 *>(row + cB);
                        rb.x = fma(-wr, vb.x, fma(-vr, wb0, rb.x));
                        rb.y = fma(-wr, vb.y, fma(-vr, wb1, rb.y));
                        *reinterpret_cast<double2 *>(row + cB) = rb;
                    }
                }
            }
            double wrow[RT_R], vrow[RT_R];
#pragma unroll
            for (int r = 0; r < RT_R; r += 2) {
                const double2 tv = *reinterpret_cast<const double2 *>(sv + S + row0 + r);
                const double2 ty = *reinterpret_cast<const double2 *>(syg + S + row0 + r);
                vrow[r] = tv.x;
                vrow[r + 1] = tv.y;
                wrow[r] = tau * ty.x - cc * tv.x;
                wrow[r + 1] = tau * ty.y - cc * tv.y;
            }
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) {
                const double2 yy = *reinterpret_cast<const double2 *>(syg + S + c0 + j);
                const double2 vv = *reinterpret_cast<const double2 *>(sv + S + c0 + j);
                const double w0 = tau * yy.x - cc * vv.x, w1 = tau * yy.y - cc * vv.y;
#pragma unroll
                for (int r = 0; r < RT_R; ++r) {
                    a[r][j] = fma(-wrow[r], vv.x, fma(-vrow[r], w0, a[r][j]));
                    a[r][j + 1] = fma(-wrow[r], vv.y, fma(-vrow[r], w1, a[r][j + 1]));
                }
            }
        }
    }
    if (S > 0) {                                   // the reflectors of the block columns are zero over the strip
        __syncthreads();
        if (tid < S) {
            sv2[0][OFF + tid] = 0.0;
            sv2[1][OFF + tid] = 0.0;
        }
        publish(S);                                // (a full strip is one panel)
    }

    // ------------------------------------------------------------------------------------------------------------
    // block columns (local index kk = g - S)
    // ------------------------------------------------------------------------------------------------------------
    double *const syb = sy + RT_SMAX, *const sdb = sd + RT_SMAX, *const seb = se + RT_SMAX, *const stb = st + RT_SMAX;
    // copy row `row` (owned by this wave) into sx.  The register row is picked with selects on the wave-uniform row
    // index: a branchy version is merged by the compiler into a dynamically indexed copy of the tile in scratch memory.
    auto publish_row = [&](int row) {
        const int rk = row & 3, ggk = (row >> 2) & 3;
        double x[RT_C];
#pragma unroll
        for (int j = 0; j < RT_C; ++j) {
            const double lo = (rk & 1) ? a[1][j] : a[0][j];
            const double hi = (rk & 1) ? a[3][j] : a[2][j];
            x[j] = (rk & 2) ? hi : lo;
        }
        if (gg == ggk) {
#pragma unroll
            for (int j = 0; j < RT_C; ++j) sx[c0 + j] = x[j];
        }
    };

    // One column of the block.  RK = kk mod 4 is a compile-time constant (the column loop below is unrolled by four), so
    // the row handed to the gen section is read straight out of its registers: picking it with selects on a run-time
    // index cost 72 v_cndmask per column on the critical path.  S is even: the v buffer (S + kk) & 1 is RK & 1.
    auto column = [&](const int kk, auto RKc) {
        constexpr int RK = decltype(RKc)::value;
        double *const svb = sv2[RK & 1] + RT_SMAX;
        // ---- gen: only the wave owning row kk (wave-uniform branch).  Row kk (= column kk by symmetry) goes to LDS, then
        // all 64 lanes work on three entries each: norm, Householder scalars, v.  Every other wave waits for this section
        // at barrier A, so it is kept short and issues ahead of the waves sharing its SIMD.
        if (wid == (kk >> 4)) {
            __builtin_amdgcn_s_setprio(3);
            if (gg == ((kk >> 2) & 3)) {
#pragma unroll
                for (int j = 0; j < RT_C; ++j) sx[c0 + j] = a[RK][j];
            }
            double x[3], sq[3];                                      // same wave wrote sx: LDS is in order
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                x[q] = sx[64 * q + lane];
                sq[q] = (64 * q + lane >= kk + 2) ? x[q] : 0.0;
            }
            const double xnorm2 = wave_sum(fma(sq[2], sq[2], fma(sq[1], sq[1], sq[0] * sq[0])));
            const double dk = sx[kk], alpha = sx[kk + 1];
            double r, u1, beta;
            rt_house(alpha, xnorm2, TB - kk - 1 >= 2, r, u1, beta);
            // reflector S + kk goes to global memory from here (zero over the strip and up to kk: the rows were cleared by
            // the scaling pass) -- written by the wave that holds it instead of by every thread after the barrier
            double *__restrict__ vrow = P.V + (long)(k0 + S + kk) * n + k0 + S;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int c = 64 * q + lane;
                double t = (c > kk + 1) ? x[q] : 0.0;
                t = (c == kk + 1) ? u1 : t;
                t = (r != 0.0 || c == kk + 1) ? t : 0.0;
                svb[c] = t;
                if (c < TB) rt_store_v(pipe, vrow + c, t);
            }
            if (lane == 0) {
                sdb[kk] = dk;
                seb[kk] = beta;
                s_r = r;
                s_u1 = fabs(u1);
                const double rem = s_trace - dk;                     // trace of the block behind column kk
                s_trace = rem;
                if (s_ttol >= 0.0 && rem <= s_ttol && s_exit_at < 0) s_exit_at = kk + 1;     // (s_ttol < 0: the check is off)
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();                                             // ---- A: v, scalars published
        const double rr = s_r, au = s_u1;
        const bool live = (16 * wid + 15 > kk) && (rr != 0.0);       // wave-uniform: still owns a row > kk
        double tau;
        // v is re-read from LDS pair by pair in both phases (6 ds_read_b128 each) instead of being held in 24 VGPRs:
        // the 4 x 12 tile already takes 96 of the 168 registers a thread may use at three waves per SIMD
        if (live) {
            const double vmy = svb[myrow];
            tau = rr * fast_rcp(au);                                 // independent of the sums below: interleaves with them
            double acc[RT_R] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) {
                const double2 vv = *reinterpret_cast<const double2 *>(svb + c0 + j);
#pragma unroll
                for (int r = 0; r < RT_R; ++r) acc[r] = fma(a[r][j + 1], vv.y, fma(a[r][j], vv.x, acc[r]));
            }
            double y = reduce_rows(acc);
            y = (myrow > kk) ? y : 0.0;
            if (h < 4) syb[myrow] = y;                               // one lane per row publishes y ..
            double dp = (h < 4) ? vmy * y : 0.0;                     // .. and carries its v.y term
            dp += dpp_mov<0xB1>(dp);                                 // the four rows of the group (lanes 0..3 of the DPP row)
            dp += dpp_mov<0x4E>(dp);
            dp = (lane_get(dp, 0) + lane_get(dp, 16)) + (lane_get(dp, 32) + lane_get(dp, 48));
            if (lane == 0) red[wid] = dp;
        } else {
            tau = rr * fast_rcp(au);
            if (lane == 0) red[wid] = 0.0;
        }
        if (tid == 0) stb[kk] = tau;
        const double htt = 0.5 * tau * tau;
        __syncthreads();                                             // ---- B: y, v.y partials published
        if (live) {
            // the wave that generates the next reflector is the critical path of the column: its update goes first
            if (wid == ((kk + 1) >> 4)) __builtin_amdgcn_s_setprio(3);
            // v.y over the twelve waves as a tree: a serial sum is twelve dependent adds in every wave
            static_assert(RT_NW == 12, "reduction tree below");
            double pr[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const double2 t2 = *reinterpret_cast<const double2 *>(red + 2 * q);
                pr[q] = t2.x + t2.y;
            }
            const double dot = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + (pr[4] + pr[5]);
            const double cc = htt * dot;
            double wrow[RT_R], vrow[RT_R];                           // re-read rather than kept live across the barrier
#pragma unroll
            for (int r = 0; r < RT_R; r += 2) {
                const double2 tv = *reinterpret_cast<const double2 *>(svb + row0 + r);
                const double2 ty = *reinterpret_cast<const double2 *>(syb + row0 + r);
                vrow[r] = tv.x;
                vrow[r + 1] = tv.y;
                wrow[r] = tau * ty.x - cc * tv.x;
                wrow[r + 1] = tau * ty.y - cc * tv.y;
            }
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) {
                const double2 yy = *reinterpret_cast<const double2 *>(syb + c0 + j);
                const double2 vv = *reinterpret_cast<const double2 *>(svb + c0 + j);
                const double w0 = tau * yy.x - cc * vv.x, w1 = tau * yy.y - cc * vv.y;
#pragma unroll
                for (int r = 0; r < RT_R; ++r) {
                    a[r][j] = fma(-wrow[r], vv.x, fma(-vrow[r], w0, a[r][j]));
                    a[r][j + 1] = fma(-wrow[r], vv.y, fma(-vrow[r], w1, a[r][j + 1]));
                }
            }
        }
    };
    int exit_at = -1, kb = 0;
    for (; kb < TB - 1; kb += 4) {                                   // TB is the same for every thread: uniform control flow
        column(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < TB - 1) column(kb + 1, std::integral_constant<int, 1>{});
        if (kb + 2 < TB - 1) column(kb + 2, std::integral_constant<int, 2>{});
        if (kb + 3 < TB - 1) column(kb + 3, std::integral_constant<int, 3>{});
        // (read behind the last column's barrier B: the flag was set in front of one of this group's barriers A)
        exit_at = s_exit_at;                                         // uniform: every thread reads the same LDS word
        if (exit_at >= 0) break;
        publish(S + min(kb + 4, TB - 1));                            // columns S .. S + kb + 3 are done (behind their barrier B)
    }
    if (exit_at >= 0) {
        // Columns exit_at .. kb + 3 of the group were still reduced (harmless: noise); from kb + 4 on the block is left as it
        // is -- d = its diagonal, e = tau = 0, the reflector rows stay as the scaling pass cleared them (H = I).
        const int kfirst = min(kb + 4, TB - 1);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RT_R; ++r)
#pragma unroll
            for (int j = 0; j < RT_C; ++j)
                if (row0 + r == c0 + j && row0 + r >= kfirst && row0 + r < TB) sdb[row0 + r] = a[r][j];
        if (tid >= kfirst && tid < TB) {
            seb[tid] = 0.0;
            stb[tid] = 0.0;
        }
    }
    // last diagonal element a[TB-1][TB-1]
    if (wid == ((TB - 1) >> 4)) {
        publish_row(TB - 1);
        if (lane == 0) {
            sdb[TB - 1] = sx[TB - 1];
            seb[TB - 1] = 0.0;
            stb[TB - 1] = 0.0;
        }
    }
    __syncthreads();
    for (int kk = tid; kk < T; kk += RT_NTH) {
        P.d[k0 + kk] = sd[OFF + kk];
        P.e[k0 + kk] = se[OFF + kk];
        if (pipe) rt_store_agent(P.tau + k0 + kk, st[OFF + kk]);
        else P.tau[k0 + kk] = st[OFF + kk];
    }
    if (pipe) {
        // the consumers of the progress word read V and tau (agent-scope stores: complete = visible); d and e are read by launches
        // that are ordered behind the END of this one by an event (capi: ev_t1), as without the progress words
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(prog, RT_PROG_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b.clk && threadIdx.x == 0) b.clk[2 * blockIdx.x + 1] = wall_clock64();
}

}  // namespace gpcsd
