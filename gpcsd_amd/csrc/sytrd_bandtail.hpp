// Band-reducing variant of the register-resident tail (sytrd_regtail.hpp): Q^T A Q = B with HALF-BANDWIDTH 4 instead of a
// tridiagonal matrix (included by eigh_dc.hip behind sytrd_regtail.hpp; round 5).
//
// Why.  The consumers of the temporal side never read its spectrum: the log-likelihood and the prediction work
// in the basis U (x) Q and only need the SHIFTED systems (lam m B + sig2 I) solved and their determinants -- for which a narrow
// band is as good as a tridiagonal matrix.  The tridiagonalisation pays two workgroup barriers, a serial reflector generation
// and a cross-wave reduction PER COLUMN (2.3 us x 250 columns: the period of the cfg3 step).  Reducing to a band of width 4
// takes the same flops but synchronises once per PANEL of four columns:
//
//   QR   ONE wave factors the part of the panel below the band by four Householder reflectors, wave-local (no barrier): V (m x 4),
//        the compact-WY factor T (4 x 4) with Q_p = I - V T V^T, and the panel's band entries (the triangle R);
//   A    barrier;  X = A22 V  (four matrix-vector products in one pass over the tile: 192 FMAs per thread, one reduce-scatter);
//   B    barrier;  wave 0: H = V^T X, M = T^T H T / 2;
//   C    barrier;  thread i: Z_i = X_i T - V_i M;  the reflectors and the band entries go to global memory; the NEXT panel's
//        rows are published (as they are before this panel's update);
//   D    barrier;  A22 -= Z V^T + V Z^T  (rank 8: 384 FMAs per thread).
//
// Who factors the panel.  Strip phase: the wave that owns the panel's four strip rows (it wrote them itself).  Block phase: WAVE 0,
// from the published rows, to which it first applies the previous panel's update itself (a 4 x 192 slice of phase D in the
// factorisation's own lane layout).  Wave 0 owns block rows 0 .. 15: from the fifth block panel on its tile is dead, it skips
// phase D and factors panel p + 1 WHILE the other waves update with panel p -- the one serial section of a panel disappears
// behind the update.
//
// Four barriers per four columns instead of eight.  The reflectors go to SytrdProb::V / tau in the layout of the tridiagonal tail
// (row k = reflector k, support from row k + 4 on), so the compact-WY machinery that forms Q (wy.hip) takes them unchanged; the
// band goes to SytrdProb::bd as bd[j * n + k] = B[k + j][k], j = 0 .. 4.
//
// Data layout as in sytrd_regtail.hpp: the trailing <= 192 rows / columns as 4 x 12 tiles in the registers of 768 threads, the
// S = T - 192 (rounded up to a multiple of 4) leading rows as a strip in LDS; strip rows are dealt to the waves in GROUPS OF FOUR
// (group g -> wave g mod 12) so that a panel's four rows belong to one wave.  Only whole problems (k_tail == 0, n <= bt_max_rows()).
#pragma once

namespace gpcsd {

constexpr int BT_W = 4;                                             // half-bandwidth = columns per panel = rows per thread tile
static_assert(BT_W == RT_R, "a panel is one row group of a wave");

__host__ __device__ inline int bt_strip_rows(int T) { return T > RT_T ? ((T - RT_T + 3) & ~3) : 0; }
// dynamic LDS: the strip, or (block phase, on the same storage) two generations of the published panel rows
inline size_t bt_lds_bytes(int T) {
    const size_t strip = ((size_t)bt_strip_rows(T) * rt_strip_ld(T) + 8) * sizeof(double);
    const size_t pan = ((size_t)2 * BT_W * RT_T + (size_t)64 * RT_R * RT_C) * sizeof(double);     // (+ a parked tile of wave 0)
    return strip > pan ? strip : pan;
}

typedef double bt_d2 __attribute__((ext_vector_type(2)));

// 16 values per lane, id = 4 a + b.  Returns in out[a] the sum over the 16 lanes of the DPP row of value 4 a + (h & 3): two
// reduce-scatter steps over the bits of b (exchange with lane ^ 1, lane ^ 2), then two rotations.  (Fixed association.)
__device__ __forceinline__ void bt_reduce16(const double (&v)[16], int h, double (&out)[4]) {
    const bool b0 = h & 1, b1 = h & 2;
    double w8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        w8[i] = keep + dpp_mov<0xB1>(send);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double keep = b1 ? w8[2 * a + 1] : w8[2 * a], send = b1 ? w8[2 * a] : w8[2 * a + 1];
        double y = keep + dpp_mov<0x4E>(send);
        y += dpp_mov<0x124>(y);                                      // row_ror:4
        y += dpp_mov<0x128>(y);                                      // row_ror:8
        out[a] = y;
    }
}

// three wave-wide sums at once (the DPP stages of independent values interleave)
__device__ __forceinline__ void bt_wave_sum3(double &x, double &y, double &z) {
    x += dpp_mov<0xB1>(x); y += dpp_mov<0xB1>(y); z += dpp_mov<0xB1>(z);
    x += dpp_mov<0x4E>(x); y += dpp_mov<0x4E>(y); z += dpp_mov<0x4E>(z);
    x += dpp_mov<0x141>(x); y += dpp_mov<0x141>(y); z += dpp_mov<0x141>(z);
    x += dpp_mov<0x140>(x); y += dpp_mov<0x140>(y); z += dpp_mov<0x140>(z);
    x = (lane_get(x, 0) + lane_get(x, 16)) + (lane_get(x, 32) + lane_get(x, 48));
    y = (lane_get(y, 0) + lane_get(y, 16)) + (lane_get(y, 32) + lane_get(y, 48));
    z = (lane_get(z, 0) + lane_get(z, 16)) + (lane_get(z, 32) + lane_get(z, 48));
}

__global__ __launch_bounds__(RT_NTH) void sybrd_btail_kernel(SytrdBatch b) {
    const SytrdProb P = sy_resolve(b, blockIdx.x);
    const int n = P.n;
    if (P.k_tail != 0 || n > RT_TMAX || n < 2) return;             // (the host only launches whole problems; see sytrd_batch_launch)
    if (b.clk && threadIdx.x == 0) b.clk[2 * blockIdx.x] = wall_clock64();
    const int T = n;
#ifdef BT_NOSTRIP                                                    // (tuning builds: the block phase alone, T <= 192)
    const int S = 0, LDT = 2;
#else
    const int S = bt_strip_rows(T), LDT = rt_strip_ld(T);
#endif
    const int TB = T - S;                                            // live rows of the register block, <= RT_T
    const int OFF = RT_SMAX - S;                                     // slot of tail-global index 0 in the LDS vectors
    extern __shared__ __attribute__((aligned(16))) double strip[];   // [S][LDT]; block phase: panA[2][BT_W][RT_T]
    // [j][slot], vector-major: a lane reads its 12 consecutive columns of one vector as 16-byte pairs 96 bytes from its neighbour's
    // (2-way bank conflicts, as the tridiagonal tail's vectors).  Interleaved [slot][j] -- one 32-byte record per index, lanes
    // 384 bytes apart -- put every other lane on the same four banks: 8-way conflicts on every operand read of phases A and D.
    __shared__ __attribute__((aligned(16))) double sV[2][BT_W][RT_TMAX];      // the panel's reflectors; double-buffered
    __shared__ __attribute__((aligned(16))) double sX[BT_W][RT_TMAX];         // X = A V, then Z in place
    __shared__ __attribute__((aligned(16))) double part[RT_NW][4][16];        // cross-row partial sums of a wave's strip groups
    __shared__ __attribute__((aligned(16))) double sT[BT_W][BT_W], sM[BT_W][BT_W], sH[BT_W * BT_W];
    __shared__ double panb[2][BT_W][BT_W];                           // R of the panel: panb[.][j][i] = R[i][j], i <= j (R[j][j] = beta_j)
    __shared__ double stau[RT_TMAX];
    __shared__ int s_live;                                           // any reflector of the current panel with tau != 0
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gg = lane >> 4, h = lane & 15;
    const int row0 = 16 * wid + 4 * gg, c0 = RT_C * h;
    const double *__restrict__ Ain = P.A0;
    double (*const panA)[BT_W][RT_T] = reinterpret_cast<double (*)[BT_W][RT_T]>(strip);

    // ---- load: the block into registers, the strip into LDS, vectors cleared
    double a[RT_R][RT_C];
#pragma unroll
    for (int r = 0; r < RT_R; ++r) {
        const int i = row0 + r;
        const bool rok = i < TB;
        const double *__restrict__ arow = Ain + (long)(S + (rok ? i : 0)) * n + S;
#pragma unroll
        for (int j = 0; j < RT_C; ++j) {
            const int c = c0 + j;
            a[r][j] = (rok && c < TB) ? arow[c] : 0.0;
        }
    }
    for (int idx = tid; idx < S * LDT + 8; idx += RT_NTH) {
        const int r = idx / LDT, c = idx - r * LDT;
        strip[idx] = (r < S && c < T) ? Ain[(long)r * n + c] : 0.0;
    }
    for (int idx = tid; idx < 2 * RT_TMAX * BT_W; idx += RT_NTH) (&sV[0][0][0])[idx] = 0.0;
    for (int idx = tid; idx < RT_TMAX * BT_W; idx += RT_NTH) (&sX[0][0])[idx] = 0.0;
    if (tid < RT_TMAX) stau[tid] = 0.0;
    __syncthreads();

    // ------------------------------------------------------------------------------------------------------------
    // Householder QR of a panel held as w[j][q]: row j of the panel (= matrix column first + j), the lane's NQ entries at the
    // (phase-specific) column indices colq[q]; pv0 = index of the first pivot (first + 4), nlim = columns that exist.  The wave
    // works alone.  On return w[j][q] holds the reflectors (zero outside their support: the rows' registers are reused), tau[j];
    // lane 0 has written T to sT, R to panb[gen] and s_live.  pick(x, c): the entry at column c, every lane (uniform).
    // Per column: the finished entries left of and at the pivot are read out (R) and zeroed, so the norm and the products below
    // need no masks; the products with the later columns (for their update) and with the earlier reflectors (for T) are three
    // values however far the panel is, reduced together.
    // ------------------------------------------------------------------------------------------------------------
    auto house4 = [&](auto &w, double (&tau)[BT_W], const auto &colq, auto NQc, int pv0, int nlim, int gen, auto pick) {
        constexpr int NQ = decltype(NQc)::value;
        double Tm[BT_W][BT_W];                                       // upper triangle, built column by column (LAPACK dlarft)
#pragma unroll
        for (int j = 0; j < BT_W; ++j) {
            const int pv = pv0 + j;
            double rj[BT_W];                                         // R[i][j], i < j: the column's entries at the earlier pivots
#pragma unroll
            for (int i = 0; i < j; ++i) rj[i] = pick(w[j], pv0 + i);
            const double alpha = pick(w[j], pv);
            double sq = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                w[j][q] = (colq[q] > pv && colq[q] < nlim) ? w[j][q] : 0.0;
                sq = fma(w[j][q], w[j][q], sq);
            }
            const double xnorm2 = wave_sum(sq);
            double r, u1, beta;
            rt_house(alpha, xnorm2, pv < nlim - 1, r, u1, beta);
            tau[j] = (r != 0.0) ? r * fast_rcp(fabs(u1)) : 0.0;
            if (r != 0.0) {                                          // uniform
#pragma unroll
                for (int q = 0; q < NQ; ++q) w[j][q] = (colq[q] == pv) ? u1 : w[j][q];
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) w[j][q] = 0.0;
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < j; ++i) panb[gen][j][i] = rj[i];
                panb[gen][j][j] = beta;                              // (= alpha when the reflector is the identity)
            }
            // d[g] = v_j . (vector of row g): g > j -> the later column (update), g < j -> the earlier reflector (T)
            double d3[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int g = t < j ? t : t + 1;                     // the three rows other than j
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) s = fma(w[j][q], w[g][q], s);
                d3[t] = s;
            }
            bt_wave_sum3(d3[0], d3[1], d3[2]);
#pragma unroll
            for (int t = j; t < 3; ++t) {                            // later columns g = t + 1
                const double f = tau[j] * d3[t];
#pragma unroll
                for (int q = 0; q < NQ; ++q) w[t + 1][q] = fma(-f, w[j][q], w[t + 1][q]);
            }
            // T[:j, j] = -tau_j T[:j, :j] (V_{:j}^T v_j),  T[j][j] = tau_j
#pragma unroll
            for (int l = 0; l < BT_W; ++l) Tm[l][j] = 0.0;
            Tm[j][j] = tau[j];
#pragma unroll
            for (int l = 0; l < j; ++l) {
                double s = 0.0;
#pragma unroll
                for (int m2 = l; m2 < j; ++m2) s = fma(Tm[l][m2], d3[m2], s);
                Tm[l][j] = -tau[j] * s;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < BT_W; ++i)
#pragma unroll
                for (int jp = 0; jp < BT_W; ++jp) sT[i][jp] = (jp >= i) ? Tm[i][jp] : 0.0;
            s_live = (tau[0] != 0.0 || tau[1] != 0.0 || tau[2] != 0.0 || tau[3] != 0.0) ? 1 : 0;
        }
    };

    int pc = 0;                                                      // panel counter: generation pc & 1 of sV / panb / panA
#ifdef BT_PHASE_CLK
    // tuning aid (a build with -DBT_PHASE_CLK): thread 0 accumulates the device clock between the barriers of a panel -- [0] up to A
    // (the factorisation, or the wait for it), [1] A..B (X), [2] B..C (H, M), [3] C..D (Z), [4] D..end (update); [5] wave 0's
    // own factorisations -- and leaves the sums in the unused corner of the band array
    unsigned long long pclk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, plast = wall_clock64();
#define BT_STAMP(k) do { const unsigned long long t_ = wall_clock64(); pclk[k] += t_ - plast; plast = t_; } while (0)
#else
#define BT_STAMP(k) do { } while (0)
#endif
    // ------------------------------------------------------------------------------------------------------------
    // phases A .. D of one panel for everybody.  `first` = tail-global index of the panel's first column
    // ------------------------------------------------------------------------------------------------------------
    // (Lane-dependent LDS addresses are re-derived per phase from these few values behind an opaque barrier: left to itself the
    // compiler hoists every address of every phase out of the panel loop, ~25 registers it then spills and reloads each phase.)
#define BT_OPAQUE(x) asm volatile("" : "+v"(x))
    auto panel_rest = [&](const int first, auto in_strip_c, const bool publish_next) {
        constexpr bool in_strip = decltype(in_strip_c)::value;
        double (*const sv)[RT_TMAX] = sV[pc & 1];
        int tid = threadIdx.x;
        BT_OPAQUE(tid);
        int lane = tid & 63, gg = lane >> 4, h = lane & 15, row0 = 16 * wid + 4 * gg, c0 = RT_C * h;
        lds_barrier();                                               // ---- A: V, T, tau published
        BT_STAMP(0);
        const bool plive = s_live != 0;                              // uniform
        const int lo = first + BT_W;                                 // first live tail-global index
        const int blk_lo = lo - S;                                   // ... as a block-local index (<= 0 in the strip phase)
        const bool wlive = 16 * wid + 15 >= blk_lo;                  // this wave still owns a live block row
        double hs[BT_W] = {0.0, 0.0, 0.0, 0.0};                      // this lane's share of H = V^T X (row k1 = 0 .. 3, column h & 3)
        if (plive) {
            // ---- X = A22 V.  Strip rows of this wave (groups g = wid, wid + 12, .. behind the panel's): lane l covers the columns
            // (2l, 2l + 1), (128 + 2l, 129 + 2l); a group's 16 partial sums (row a, vector k) are reduced over the wave together.
            if (in_strip) {
                const int cA = 2 * lane, cB = 128 + 2 * lane;
                const bool okB = cB < T;
                const int gfirst = first / BT_W + 1;
                for (int g = gfirst + (wid - gfirst % RT_NW + RT_NW) % RT_NW; g < S / BT_W; g += RT_NW) {
                    double pr[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) pr[i] = 0.0;
#pragma unroll 1
                    for (int half = 0; half < 2; ++half) {           // the lane's column pair (cA | cB), one at a time (registers)
                        if (half == 1 && !okB) break;
                        const int cc = half ? cB : cA;
                        bt_d2 vk[BT_W];                               // (V[k][cc], V[k][cc + 1])
#pragma unroll
                        for (int k = 0; k < BT_W; ++k) vk[k] = *reinterpret_cast<const bt_d2 *>(&sv[k][OFF + cc]);
#pragma unroll
                        for (int ar = 0; ar < 4; ++ar) {
                            const bt_d2 ra = *reinterpret_cast<const bt_d2 *>(strip + (BT_W * g + ar) * LDT + cc);
#pragma unroll
                            for (int k = 0; k < BT_W; ++k) pr[4 * ar + k] = fma(ra.y, vk[k].y, fma(ra.x, vk[k].x, pr[4 * ar + k]));
                        }
                    }
                    double o4[4];
                    bt_reduce16(pr, h, o4);                          // o4[a]: DPP-row sum of (row a, vector h & 3)
                    if (h < 4) {
#pragma unroll
                        for (int ar = 0; ar < 4; ++ar) part[wid][gg][4 * ar + h] = o4[ar];
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 16) {
                        const double tot = (part[wid][0][lane] + part[wid][1][lane]) + (part[wid][2][lane] + part[wid][3][lane]);
                        const int sl = OFF + BT_W * g + (lane >> 2);
                        sX[lane & 3][sl] = tot;
#pragma unroll
                        for (int k1 = 0; k1 < BT_W; ++k1) hs[k1] = fma(sv[k1][sl], tot, hs[k1]);      // H[k1][lane & 3] += V[row][k1] X[row][lane & 3]
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // block rows: the tile from registers plus (strip phase) the strip COLUMNS of these rows
#ifndef BT_NO_X
            if (wlive) {
                double acc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0;
#ifndef BT_X_NOFMA
#pragma unroll
                for (int j = 0; j < RT_C; j += 2) {
#pragma unroll
                    for (int k = 0; k < BT_W; ++k) {
                        const bt_d2 v2 = *reinterpret_cast<const bt_d2 *>(&sv[k][RT_SMAX + c0 + j]);
#pragma unroll
                        for (int r = 0; r < RT_R; ++r) acc[4 * r + k] = fma(a[r][j + 1], v2.y, fma(a[r][j], v2.x, acc[4 * r + k]));
                    }
                }
#else
                acc[0] = a[0][0]; acc[5] = a[1][1]; acc[10] = a[2][2]; acc[15] = a[3][3];
#endif
                if (in_strip) {
                    for (int rs = lo + (h - lo % 16 + 16) % 16; rs < S; rs += 16) {
                        const double vr4[4] = {sv[0][OFF + rs], sv[1][OFF + rs], sv[2][OFF + rs], sv[3][OFF + rs]};
                        const bt_d2 s01 = *reinterpret_cast<const bt_d2 *>(strip + rs * LDT + S + row0);
                        const bt_d2 s23 = *reinterpret_cast<const bt_d2 *>(strip + rs * LDT + S + row0 + 2);
                        const double sr[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
                        for (int r = 0; r < RT_R; ++r)
#pragma unroll
                            for (int k = 0; k < BT_W; ++k) acc[4 * r + k] = fma(sr[r], vr4[k], acc[4 * r + k]);
                    }
                }
                double o4[4];
#ifndef BT_X_NORED
                bt_reduce16(acc, h, o4);
#else
                o4[0] = acc[0] + acc[4]; o4[1] = acc[5] + acc[1]; o4[2] = acc[10] + acc[2]; o4[3] = acc[15] + acc[3];
#endif
                const int ar = h >> 2;
                const double lo01 = (ar & 1) ? o4[1] : o4[0], hi23 = (ar & 1) ? o4[3] : o4[2];
                const double xv = (ar & 2) ? hi23 : lo01;
                const int i = row0 + ar;                             // block row of this lane's value, vector h & 3
                const double xm = (i >= blk_lo && i < TB) ? xv : 0.0;
                sX[h & 3][RT_SMAX + i] = xm;
#pragma unroll
                for (int k1 = 0; k1 < BT_W; ++k1) hs[k1] = fma(sv[k1][RT_SMAX + i], xm, hs[k1]);
            }
#endif
            // H = V^T X: the lane holds products for column h & 3 -- summed over the four lanes of the DPP row that share it (two
            // rotations), the row groups' and the waves' partial sums go through LDS to wave 0
#pragma unroll
            for (int k1 = 0; k1 < BT_W; ++k1) {
                hs[k1] += dpp_mov<0x124>(hs[k1]);                    // row_ror:4
                hs[k1] += dpp_mov<0x128>(hs[k1]);                    // row_ror:8
            }
            if (h < 4) {
#pragma unroll
                for (int k1 = 0; k1 < BT_W; ++k1) part[wid][gg][4 * k1 + h] = hs[k1];
            }
        }
        lds_barrier();                                               // ---- B: X published
        BT_STAMP(1);
        BT_OPAQUE(tid);
        lane = tid & 63; gg = lane >> 4; h = lane & 15; row0 = 16 * wid + 4 * gg; c0 = RT_C * h;
#ifndef BT_NO_H
        if (plive && wid == 0) {
            // H: entry e = lane & 15 (k1 = e >> 2, k2 = e & 3), the 48 partial sums (12 waves x 4 row groups) a quarter per lane group
            // in a fixed order, then M = T^T H T / 2
            const int e = lane & 15, q = lane >> 4;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < RT_NW; ++w) t += part[w][q][e];
            __builtin_amdgcn_wave_barrier();
            part[0][q][e] = t;
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) sH[lane] = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) {
                const int ia = lane >> 2, ib = lane & 3;             // M[ia][ib] = 1/2 sum_{k,l} T[k][ia] H[k][l] T[l][ib]
                double s2 = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double u = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) u = fma(sH[4 * k + l], sT[l][ib], u);
                    s2 = fma(sT[k][ia], u, s2);
                }
                sM[ia][ib] = 0.5 * s2;
            }
        }
#endif
        lds_barrier();                                               // ---- C: M published
        BT_STAMP(2);
        BT_OPAQUE(tid);
        lane = tid & 63; gg = lane >> 4; h = lane & 15; row0 = 16 * wid + 4 * gg; c0 = RT_C * h;
        if (tid < RT_TMAX) {                                         // Z_i = X_i T - V_i M, in place of X (dead / padding slots: zero)
            const int gidx = tid - OFF;
            const double vv[4] = {sv[0][tid], sv[1][tid], sv[2][tid], sv[3][tid]};
            double z[4] = {0.0, 0.0, 0.0, 0.0};
            if (plive && gidx >= lo && gidx < T) {
                const double xx[4] = {sX[0][tid], sX[1][tid], sX[2][tid], sX[3][tid]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) s = fma(xx[l], sT[l][k], s);
#pragma unroll
                    for (int l = 0; l < 4; ++l) s = fma(-vv[l], sM[l][k], s);
                    z[k] = s;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sX[k][tid] = z[k];
            // the panel's reflectors to global memory (row first + j of V, tail-global column gidx)
            if (gidx >= 0 && gidx < T) {
                double *__restrict__ vg = P.V + (long)first * n + gidx;
                vg[0] = vv[0]; vg[n] = vv[1]; vg[2 * (long)n] = vv[2]; vg[3 * (long)n] = vv[3];
            }
        } else if (tid < RT_TMAX + BT_W * (BT_W + 1)) {
            // the band entries of the panel's four columns: bd[d][first + j] = B[first + j + d][first + j]
            const int e = tid - RT_TMAX, j = e / (BT_W + 1), d = e - j * (BT_W + 1);
            if (first + j + d < T) {
                double val;
                if (d < BT_W - j) {                                  // inside the diagonal block: the panel row's own entry
                    if (in_strip) val = strip[(first + j) * LDT + first + j + d];
                    else val = panA[pc & 1][j][first - S + j + d];
                } else {
                    val = panb[pc & 1][j][j + d - BT_W];             // R[i][j] with first + 4 + i = first + j + d
                }
                P.bd[(long)d * n + first + j] = val;
            }
        }
        if (publish_next) {                                          // the next block panel's rows, as they are before this panel's update
            const int nk = blk_lo;                                   // its first block row
            if (wid == (nk >> 4) && gg == ((nk >> 2) & 3)) {
#pragma unroll
                for (int r = 0; r < RT_R; ++r)
#pragma unroll
                    for (int j = 0; j < RT_C; j += 2) *reinterpret_cast<bt_d2 *>(&panA[(pc + 1) & 1][r][c0 + j]) = bt_d2{a[r][j], a[r][j + 1]};
            }
        }
        lds_barrier();                                               // ---- D: Z published
        BT_STAMP(3);
        BT_OPAQUE(tid);
        lane = tid & 63; gg = lane >> 4; h = lane & 15; row0 = 16 * wid + 4 * gg; c0 = RT_C * h;
        if (plive) {
            // ---- A22 -= Z V^T + V Z^T
            if (in_strip) {
                const int cA = 2 * lane, cB = 128 + 2 * lane;
                const bool okB = cB < T;
                const int gfirst = first / BT_W + 1;
                const int g0 = gfirst + (wid - gfirst % RT_NW + RT_NW) % RT_NW;
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {               // the lane's column pair (cA | cB), one at a time (registers)
                    if (half == 1 && !okB) break;
                    const int cc = half ? cB : cA;
                    double vc[2][4], zc[2][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bt_d2 v2 = *reinterpret_cast<const bt_d2 *>(&sv[k][OFF + cc]), z2 = *reinterpret_cast<const bt_d2 *>(&sX[k][OFF + cc]);
                        vc[0][k] = v2.x; vc[1][k] = v2.y;
                        zc[0][k] = z2.x; zc[1][k] = z2.y;
                    }
                    for (int g = g0; g < S / BT_W; g += RT_NW) {
#pragma unroll
                        for (int ar = 0; ar < 4; ++ar) {
                            const int r = BT_W * g + ar;
                            double *__restrict__ row = strip + r * LDT;
                            const double zr[4] = {sX[0][OFF + r], sX[1][OFF + r], sX[2][OFF + r], sX[3][OFF + r]};
                            const double vr[4] = {sv[0][OFF + r], sv[1][OFF + r], sv[2][OFF + r], sv[3][OFF + r]};
                            bt_d2 e = *reinterpret_cast<const bt_d2 *>(row + cc);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                e.x = fma(-vr[k], zc[0][k], fma(-zr[k], vc[0][k], e.x));
                                e.y = fma(-vr[k], zc[1][k], fma(-zr[k], vc[1][k], e.y));
                            }
                            *reinterpret_cast<bt_d2 *>(row + cc) = e;
                        }
                    }
                }
            }
#ifndef BT_NO_UPD
            if (wlive) {
#pragma unroll
                for (int kh = 0; kh < 4; kh += 2) {                  // two vectors at a time (registers)
                    double zr[RT_R][2], vr[RT_R][2];              // [row][vector kh, kh + 1]
#pragma unroll
                    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
                        for (int r = 0; r < RT_R; r += 2) {
                            const bt_d2 zz = *reinterpret_cast<const bt_d2 *>(&sX[kh + kq][RT_SMAX + row0 + r]);
                            const bt_d2 vv = *reinterpret_cast<const bt_d2 *>(&sv[kh + kq][RT_SMAX + row0 + r]);
                            zr[r][kq] = zz.x; zr[r + 1][kq] = zz.y; vr[r][kq] = vv.x; vr[r + 1][kq] = vv.y;
                        }
#pragma unroll
                    for (int j = 0; j < RT_C; j += 2) {
                        const bt_d2 zc0 = *reinterpret_cast<const bt_d2 *>(&sX[kh][RT_SMAX + c0 + j]), zc1 = *reinterpret_cast<const bt_d2 *>(&sX[kh + 1][RT_SMAX + c0 + j]);
                        const bt_d2 vc0 = *reinterpret_cast<const bt_d2 *>(&sv[kh][RT_SMAX + c0 + j]), vc1 = *reinterpret_cast<const bt_d2 *>(&sv[kh + 1][RT_SMAX + c0 + j]);
#pragma unroll
                        for (int r = 0; r < RT_R; ++r) {
                            double e = a[r][j], f = a[r][j + 1];
                            e = fma(-zr[r][0], vc0.x, e); f = fma(-zr[r][0], vc0.y, f);
                            e = fma(-vr[r][0], zc0.x, e); f = fma(-vr[r][0], zc0.y, f);
                            e = fma(-zr[r][1], vc1.x, e); f = fma(-zr[r][1], vc1.y, f);
                            e = fma(-vr[r][1], zc1.x, e); f = fma(-vr[r][1], zc1.y, f);
                            a[r][j] = e;
                            a[r][j + 1] = f;
                        }
                    }
                }
            }
#endif
        }
        ++pc;
        BT_STAMP(4);
    };

    // ------------------------------------------------------------------------------------------------------------
    // strip panels: first = 0, 4, .. < S.  The panel's rows are strip rows of group first / 4, written by their owner wave itself
    // ------------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int first = 0; first < S; first += BT_W) {
        if (wid == (first / BT_W) % RT_NW) {
            __builtin_amdgcn_s_setprio(3);
            const int cA = 2 * lane, cB = 128 + 2 * lane;
            const bool okB = cB < T;
            const int colq[4] = {cA, cA + 1, cB, cB + 1};
            double w[BT_W][4], tau[BT_W];
#pragma unroll
            for (int j = 0; j < BT_W; ++j) {
                const double *__restrict__ row = strip + (first + j) * LDT;
                const bt_d2 xa = *reinterpret_cast<const bt_d2 *>(row + cA);
                const bt_d2 xb = okB ? *reinterpret_cast<const bt_d2 *>(row + cB) : bt_d2{0.0, 0.0};
                w[j][0] = xa.x; w[j][1] = xa.y; w[j][2] = xb.x; w[j][3] = xb.y;
            }
            // entry at tail-global column c (uniform).  All four candidates are read and the VALUE is selected: a select between the
            // array's elements is turned into a run-time index by the compiler, and the array then lives in scratch memory.
            auto pick = [&](const double (&x)[4], int c) {
                const int l = (c < 128 ? c : c - 128) >> 1;
                const double t0 = lane_get(x[0], l), t1 = lane_get(x[1], l), t2 = lane_get(x[2], l), t3 = lane_get(x[3], l);
                const double lo2 = (c & 1) ? t1 : t0, hi2 = (c & 1) ? t3 : t2;
                return c < 128 ? lo2 : hi2;
            };
            house4(w, tau, colq, std::integral_constant<int, 4>{}, first + BT_W, T, pc & 1, pick);
            double (*const sv)[RT_TMAX] = sV[pc & 1];
#pragma unroll
            for (int k = 0; k < BT_W; ++k) {                         // (the pad column of an odd T holds zeros)
                *reinterpret_cast<bt_d2 *>(&sv[k][OFF + cA]) = bt_d2{w[k][0], w[k][1]};
                if (okB) *reinterpret_cast<bt_d2 *>(&sv[k][OFF + cB]) = bt_d2{w[k][2], w[k][3]};
            }
            if (lane < BT_W) stau[OFF + first + lane] = tau[lane];
            __builtin_amdgcn_s_setprio(0);
        }
        panel_rest(first, std::true_type{}, false);
    }
    // ------------------------------------------------------------------------------------------------------------
    // block panels (block-local first column kk = 0, 4, ..).  Wave 0 factors every one of them from the published rows.
    // ------------------------------------------------------------------------------------------------------------
    const int npan = TB >= BT_W + 2 ? (TB - BT_W - 2) / BT_W + 1 : 0;              // panels kk = 0, 4, .. with kk + 4 < TB - 1
    __syncthreads();                                                 // the strip is dead: its storage becomes panA
    if (S > 0) {                                                     // the reflectors of the block panels are zero over the strip
        for (int idx = tid; idx < 2 * RT_SMAX * BT_W; idx += RT_NTH) {
            const int bsel = idx / (RT_SMAX * BT_W), rem = idx - bsel * RT_SMAX * BT_W;
            sV[bsel][rem / RT_SMAX][rem % RT_SMAX] = 0.0;
        }
        for (int idx = tid; idx < RT_SMAX * BT_W; idx += RT_NTH) sX[idx / RT_SMAX][idx % RT_SMAX] = 0.0;
    }
    if (npan > 0 && wid == 0 && gg == 0) {                           // the first block panel's rows: current (nothing pending)
#pragma unroll
        for (int r = 0; r < RT_R; ++r)
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) *reinterpret_cast<bt_d2 *>(&panA[pc & 1][r][c0 + j]) = bt_d2{a[r][j], a[r][j + 1]};
    }
    __syncthreads();
    int kk = 0;
#pragma unroll 1
    for (int ip = 0; ip < npan; ++ip, kk += BT_W) {
        if (wid == 0) {
#ifdef BT_PHASE_CLK
            const unsigned long long tq0 = wall_clock64();
#endif
            __builtin_amdgcn_s_setprio(3);
            // The factorisation wants the registers the wave's 4 x 12 tile occupies.  While the tile is still needed (the first
            // three block panels; the last ones of a tiny block, whose band entries come out of the registers) it is parked in
            // LDS behind panA for the duration -- afterwards it is dead and simply redefined: either way the compiler sees no
            // live tile across this section.
            const bool tile_live = (kk + BT_W <= 15) || (npan <= 3);
            int lane = threadIdx.x & 63;
            BT_OPAQUE(lane);
            const int h = lane & 15, c0 = RT_C * h;
            (void)c0;
            bt_d2 *const tsave = reinterpret_cast<bt_d2 *>(strip + 2 * BT_W * RT_T) + lane;
            if (tile_live) {
#pragma unroll
                for (int r = 0; r < RT_R; ++r)
#pragma unroll
                    for (int j = 0; j < RT_C; j += 2) tsave[(r * (RT_C / 2) + j / 2) * 64] = bt_d2{a[r][j], a[r][j + 1]};
            }
            const int colq[3] = {lane, 64 + lane, 128 + lane};
            double w[BT_W][3], tau[BT_W];
#pragma unroll
            for (int j = 0; j < BT_W; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) w[j][q] = panA[pc & 1][j][64 * q + lane];
            if (ip > 0) {
                // the rows were published before the previous panel's update: apply it here (rows kk + j, the lane's three columns)
                double (*const svp)[RT_TMAX] = sV[(pc - 1) & 1];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int cs = RT_SMAX + 64 * q + lane;
                    double vcq[4], zcq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { vcq[k] = svp[k][cs]; zcq[k] = sX[k][cs]; }
                    double upd[BT_W];
#pragma unroll
                    for (int j = 0; j < BT_W; ++j) {
                        const int rs = RT_SMAX + kk + j;
                        double e = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            e = fma(sX[k][rs], vcq[k], e);
                            e = fma(svp[k][rs], zcq[k], e);
                        }
                        upd[j] = e;
                    }
#pragma unroll
                    for (int j = 0; j < BT_W; ++j) w[j][q] -= upd[j];
                }
#pragma unroll
                for (int j = 0; j < BT_W; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q) panA[pc & 1][j][64 * q + lane] = w[j][q];     // (the band entries are read from here)
            }
#ifdef BT_PHASE_CLK
            const unsigned long long tq1 = wall_clock64();
            pclk[6] += tq1 - tq0;
#endif
            auto pick = [&](const double (&x)[3], int c) {           // (values selected, not elements: see the strip phase)
                const int q = c >> 6, l = c & 63;
                const double t0 = lane_get(x[0], l), t1 = lane_get(x[1], l), t2 = lane_get(x[2], l);
                return (q == 0) ? t0 : (q == 1) ? t1 : t2;
            };
#ifndef BT_NO_QR
            house4(w, tau, colq, std::integral_constant<int, 3>{}, kk + BT_W, TB, pc & 1, pick);
#endif
#ifdef BT_PHASE_CLK
            const unsigned long long tq2 = wall_clock64();
            pclk[7] += tq2 - tq1;
#endif
            double (*const sv)[RT_TMAX] = sV[pc & 1];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int k = 0; k < BT_W; ++k) sv[k][RT_SMAX + 64 * q + lane] = w[k][q];
            if (lane < BT_W) stau[RT_SMAX + kk + lane] = tau[lane];
            if (tile_live) {
#pragma unroll
                for (int r = 0; r < RT_R; ++r)
#pragma unroll
                    for (int j = 0; j < RT_C; j += 2) {
                        const bt_d2 t2 = tsave[(r * (RT_C / 2) + j / 2) * 64];
                        a[r][j] = t2.x;
                        a[r][j + 1] = t2.y;
                    }
            } else {
#pragma unroll
                for (int r = 0; r < RT_R; ++r)
#pragma unroll
                    for (int j = 0; j < RT_C; ++j) a[r][j] = 0.0;
            }
            __builtin_amdgcn_s_setprio(0);
#ifdef BT_PHASE_CLK
            pclk[5] += wall_clock64() - tq0;
#endif
        }
        panel_rest(S + kk, std::false_type{}, ip + 1 < npan);
    }
    // what is left of the block (columns kk ..) lies inside the band: straight out of the registers
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RT_R; ++r)
#pragma unroll
        for (int j = 0; j < RT_C; ++j) {
            const int i = row0 + r, c = c0 + j;
            if (c >= kk && i >= c && i - c <= BT_W && i < TB) P.bd[(long)(i - c) * n + S + c] = a[r][j];
        }
    for (int g = tid; g < T; g += RT_NTH) P.tau[g] = stau[OFF + g];
#ifdef BT_PHASE_CLK
    if (tid == 0 && n >= 8) {
        P.bd[(long)4 * n + n - 4] = (double)pclk[0]; P.bd[(long)4 * n + n - 3] = (double)pclk[1];
        P.bd[(long)4 * n + n - 2] = (double)pclk[2]; P.bd[(long)4 * n + n - 1] = (double)pclk[3];
        P.bd[(long)3 * n + n - 3] = (double)pclk[4]; P.bd[(long)3 * n + n - 2] = (double)pclk[5];
        P.bd[(long)3 * n + n - 1] = (double)pclk[6]; P.bd[(long)2 * n + n - 2] = (double)pclk[7];
    }
#endif
    if (b.clk && threadIdx.x == 0) b.clk[2 * blockIdx.x + 1] = wall_clock64();
}

}  // namespace gpcsd
