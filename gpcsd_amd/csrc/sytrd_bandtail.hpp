// Band-reducing variant of the register-resident tail (sytrd_regtail.hpp): Q^T A Q = B with HALF-BANDWIDTH 4 instead of a
// tridiagonal matrix (included by eigh_dc.hip behind sytrd_regtail.hpp; round 5).
//
// Why.  The consumers of the temporal side never read its spectrum: the log-likelihood and the prediction work
// in the basis U (x) Q and only need the SHIFTED systems (lam m B + sig2 I) solved and their determinants -- for which a narrow
// band is as good as a tridiagonal matrix.  The tridiagonalisation pays two workgroup barriers, a serial reflector generation
// and a cross-wave reduction PER COLUMN (2.3 us x 250 columns: the period of the cfg3 step).  Reducing to a band of width 4
// takes the same flops but synchronises once per PANEL of four columns:
//
//   QR   the wave that owns the panel's four rows (= columns, by symmetry) factors the part of the panel below the band by
//        four Householder reflectors, wave-local (no barrier): V (m x 4), the compact-WY factor T (4 x 4) with
//        Q_p = I - V T V^T, and the panel's band entries (the 4 x 4 diagonal block and the triangle R);
//   A    barrier;  X = A22 V  (four matrix-vector products in one pass over the tile: 192 FMAs per thread, one reduce-scatter);
//   B    barrier;  wave 0: H = V^T X, M = T^T H T;
//   C    barrier;  thread i: Z_i = X_i T - V_i M / 2;
//   D    barrier;  A22 -= Z V^T + V Z^T  (rank 8: 384 FMAs per thread).
//
// Four barriers and one serial section per four columns instead of eight and four.  The reflectors go to SytrdProb::V / tau in
// the layout of the tridiagonal tail (row k = reflector k, support from row k + 4 on), so the compact-WY machinery that forms
// Q (wy.hip) takes them unchanged; the band goes to SytrdProb::bd as bd[j * n + k] = B[k + j][k], j = 0 .. 4.
//
// Data layout as in sytrd_regtail.hpp: the trailing <= 192 rows / columns as 4 x 12 tiles in the registers of 768 threads, the
// S = T - 192 (rounded up to a multiple of 4) leading rows as a strip in LDS; strip rows are dealt to the waves in GROUPS OF FOUR
// (group g -> wave g mod 12) so that a panel's four rows belong to one wave.  Only whole problems (k_tail == 0, n <= 256).
#pragma once

namespace gpcsd {

constexpr int BT_W = 4;                                             // half-bandwidth = columns per panel = rows per thread tile
static_assert(BT_W == RT_R, "a panel is one row group of a wave");

__host__ __device__ inline int bt_strip_rows(int T) { return T > RT_T ? ((T - RT_T + 3) & ~3) : 0; }
inline size_t bt_lds_bytes(int T) { return ((size_t)bt_strip_rows(T) * rt_strip_ld(T) + 8) * sizeof(double); }

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

__global__ __launch_bounds__(RT_NTH) void sybrd_btail_kernel(SytrdBatch b) {
    const SytrdProb P = sy_resolve(b, blockIdx.x);
    const int n = P.n;
    if (P.k_tail != 0 || n > RT_TMAX || n < 2) return;             // (the host only launches whole problems; see sytrd_batch_launch)
    if (b.clk && threadIdx.x == 0) b.clk[2 * blockIdx.x] = wall_clock64();
    const int T = n;
    const int S = bt_strip_rows(T), LDT = rt_strip_ld(T);
    const int TB = T - S;                                            // live rows of the register block, <= RT_T
    const int OFF = RT_SMAX - S;                                     // slot of tail-global index 0 in the LDS vectors
    extern __shared__ __attribute__((aligned(16))) double strip[];   // [S][LDT]
    __shared__ __attribute__((aligned(16))) double sV[2][RT_TMAX][BT_W];      // the panel's reflectors, [slot][j]; double-buffered
    __shared__ __attribute__((aligned(16))) double sX[RT_TMAX][BT_W];         // X = A V, then Z in place
    __shared__ __attribute__((aligned(16))) double pan[BT_W][RT_TMAX];        // the panel's rows (block phase) / its final values
    __shared__ __attribute__((aligned(16))) double part[RT_NW][4][16];        // cross-row partial sums of a wave's strip groups
    __shared__ __attribute__((aligned(16))) double sT[BT_W][BT_W], sM[BT_W][BT_W], sH[BT_W * BT_W];
    __shared__ double stau[RT_TMAX];
    __shared__ int s_live;                                           // any reflector of the current panel with tau != 0
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gg = lane >> 4, h = lane & 15;
    const int row0 = 16 * wid + 4 * gg, c0 = RT_C * h;
    const double *__restrict__ Ain = P.A0;

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
    // works alone.  On return: pan[j][slot0 + c] holds the panel's final values up to the pivots (R on and above them), w[j][q] the
    // reflectors (zero outside their support: the rows' registers are reused), tau[j], and lane 0 has written T to sT.
    // pick(x, c): the entry at column c, every lane.
    // ------------------------------------------------------------------------------------------------------------
    auto house4 = [&](auto &w, double (&tau)[BT_W], const auto &colq, auto NQc, int pv0, int nlim, int slot0, int clim, auto pick) {
        constexpr int NQ = decltype(NQc)::value;
#pragma unroll
        for (int j = 0; j < BT_W; ++j) {
            const int pv = pv0 + j;
            double sq = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double m = (colq[q] > pv && colq[q] < nlim) ? w[j][q] : 0.0;
                sq = fma(m, m, sq);
            }
            const double alpha = pick(w[j], pv);
            const double xnorm2 = wave_sum(sq);
            double r, u1, beta;
            rt_house(alpha, xnorm2, pv < nlim - 1, r, u1, beta);
            tau[j] = (r != 0.0) ? r * fast_rcp(fabs(u1)) : 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = colq[q];
                const double x = w[j][q];
                // the row's final values go to pan (left of the pivot untouched, beta at it; nobody reads beyond it) and its
                // registers become the reflector
                if (c < clim) pan[j][slot0 + c] = (c == pv) ? beta : x;
                double t = (c > pv && c < nlim) ? x : 0.0;
                t = (c == pv) ? u1 : t;
                w[j][q] = (r != 0.0) ? t : 0.0;
            }
            double dp[BT_W];
#pragma unroll
            for (int jp = j + 1; jp < BT_W; ++jp) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) s = fma(w[j][q], w[jp][q], s);
                dp[jp] = s;
            }
#pragma unroll
            for (int jp = j + 1; jp < BT_W; ++jp) dp[jp] = wave_sum(dp[jp]);
#pragma unroll
            for (int jp = j + 1; jp < BT_W; ++jp) {
                const double f = tau[j] * dp[jp];
#pragma unroll
                for (int q = 0; q < NQ; ++q) w[jp][q] = fma(-f, w[j][q], w[jp][q]);
            }
        }
        // T = (diag(1 / tau) + striu(V^T V))^-1, column by column (LAPACK dlarft): T[:i, i] = -tau_i T[:i, :i] (V_{:i}^T v_i)
        double g[BT_W][BT_W];
#pragma unroll
        for (int i = 0; i < BT_W; ++i)
#pragma unroll
            for (int jp = i + 1; jp < BT_W; ++jp) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) s = fma(w[i][q], w[jp][q], s);
                g[i][jp] = s;
            }
#pragma unroll
        for (int i = 0; i < BT_W; ++i)
#pragma unroll
            for (int jp = i + 1; jp < BT_W; ++jp) g[i][jp] = wave_sum(g[i][jp]);
        double Tm[BT_W][BT_W];
#pragma unroll
        for (int i = 0; i < BT_W; ++i)
#pragma unroll
            for (int jp = 0; jp < BT_W; ++jp) Tm[i][jp] = 0.0;
#pragma unroll
        for (int i = 0; i < BT_W; ++i) {
            Tm[i][i] = tau[i];
#pragma unroll
            for (int l = 0; l < i; ++l) {
                double s = 0.0;
#pragma unroll
                for (int m2 = l; m2 < i; ++m2) s = fma(Tm[l][m2], g[m2][i], s);
                Tm[l][i] = -tau[i] * s;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < BT_W; ++i)
#pragma unroll
                for (int jp = 0; jp < BT_W; ++jp) sT[i][jp] = Tm[i][jp];
            s_live = (tau[0] != 0.0 || tau[1] != 0.0 || tau[2] != 0.0 || tau[3] != 0.0) ? 1 : 0;
        }
    };

    // the band entries of the panel's four columns, from pan[j][slot] (final values): bd[d][first + j] = B[first + j + d][first + j]
    auto emit_band = [&](int first, int slot_first, int nlim) {      // first: tail-global column of panel row 0; nlim in the same units
        if (lane < BT_W * (BT_W + 1)) {
            const int j = lane / (BT_W + 1), d = lane - j * (BT_W + 1);
            if (first + j + d < nlim) P.bd[(long)d * n + first + j] = pan[j][slot_first + j + d];
        }
    };

    int pc = 0;                                                      // panel counter: V buffer pc & 1
    // one panel's phases A .. D for everybody; `first` = tail-global index of the panel's first column
    auto panel_rest = [&](const int first, const bool in_strip) {
        double (*const sv)[BT_W] = sV[pc & 1];
        __syncthreads();                                             // ---- A: V, T, tau published
        const bool plive = s_live != 0;                              // uniform
        const int lo = first + BT_W;                                 // first live tail-global index
        const int blk_lo = lo - S;                                   // ... as a block-local index (<= 0 in the strip phase)
        const bool wlive = 16 * wid + 15 >= blk_lo;                  // this wave still owns a live block row
        if (plive) {
            // ---- X = A22 V.  Strip rows of this wave (groups g = wid, wid + 12, .. behind the panel's): lane l covers the columns
            // (2l, 2l + 1), (128 + 2l, 129 + 2l); a group's 16 partial sums (row a, vector k) are reduced over the wave together.
            if (in_strip) {
                const int cA = 2 * lane, cB = 128 + 2 * lane;
                const bool okB = cB < T;
                double vc[4][BT_W];                                  // V at the lane's four columns
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = ((q & 2) ? cB : cA) + (q & 1);
                    const bool ok = (q & 2) ? okB : true;
                    const bt_d2 v01 = ok ? *reinterpret_cast<const bt_d2 *>(&sv[OFF + c][0]) : bt_d2{0.0, 0.0};
                    const bt_d2 v23 = ok ? *reinterpret_cast<const bt_d2 *>(&sv[OFF + c][2]) : bt_d2{0.0, 0.0};
                    vc[q][0] = v01.x; vc[q][1] = v01.y; vc[q][2] = v23.x; vc[q][3] = v23.y;
                }
                const int gfirst = first / BT_W + 1;
                for (int g = gfirst + (wid - gfirst % RT_NW + RT_NW) % RT_NW; g < S / BT_W; g += RT_NW) {
                    double pr[16];
#pragma unroll
                    for (int ar = 0; ar < 4; ++ar) {
                        const double *__restrict__ row = strip + (BT_W * g + ar) * LDT;
                        const bt_d2 ra = *reinterpret_cast<const bt_d2 *>(row + cA);
                        const bt_d2 rb = okB ? *reinterpret_cast<const bt_d2 *>(row + cB) : bt_d2{0.0, 0.0};
#pragma unroll
                        for (int k = 0; k < BT_W; ++k)
                            pr[4 * ar + k] = fma(rb.y, vc[3][k], fma(rb.x, vc[2][k], fma(ra.y, vc[1][k], ra.x * vc[0][k])));
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
                        sX[OFF + BT_W * g + (lane >> 2)][lane & 3] = tot;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // block rows: the tile from registers plus (strip phase) the strip COLUMNS of these rows
            if (wlive) {
                double acc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0;
#pragma unroll
                for (int j = 0; j < RT_C; ++j) {
                    const bt_d2 v01 = *reinterpret_cast<const bt_d2 *>(&sv[RT_SMAX + c0 + j][0]);
                    const bt_d2 v23 = *reinterpret_cast<const bt_d2 *>(&sv[RT_SMAX + c0 + j][2]);
#pragma unroll
                    for (int r = 0; r < RT_R; ++r) {
                        acc[4 * r + 0] = fma(a[r][j], v01.x, acc[4 * r + 0]);
                        acc[4 * r + 1] = fma(a[r][j], v01.y, acc[4 * r + 1]);
                        acc[4 * r + 2] = fma(a[r][j], v23.x, acc[4 * r + 2]);
                        acc[4 * r + 3] = fma(a[r][j], v23.y, acc[4 * r + 3]);
                    }
                }
                if (in_strip) {
                    for (int rs = lo + (h - lo % 16 + 16) % 16; rs < S; rs += 16) {
                        const bt_d2 v01 = *reinterpret_cast<const bt_d2 *>(&sv[OFF + rs][0]);
                        const bt_d2 v23 = *reinterpret_cast<const bt_d2 *>(&sv[OFF + rs][2]);
                        const bt_d2 s01 = *reinterpret_cast<const bt_d2 *>(strip + rs * LDT + S + row0);
                        const bt_d2 s23 = *reinterpret_cast<const bt_d2 *>(strip + rs * LDT + S + row0 + 2);
                        const double sr[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
                        for (int r = 0; r < RT_R; ++r) {
                            acc[4 * r + 0] = fma(sr[r], v01.x, acc[4 * r + 0]);
                            acc[4 * r + 1] = fma(sr[r], v01.y, acc[4 * r + 1]);
                            acc[4 * r + 2] = fma(sr[r], v23.x, acc[4 * r + 2]);
                            acc[4 * r + 3] = fma(sr[r], v23.y, acc[4 * r + 3]);
                        }
                    }
                }
                double o4[4];
                bt_reduce16(acc, h, o4);
                const int ar = h >> 2;
                const double lo01 = (ar & 1) ? o4[1] : o4[0], hi23 = (ar & 1) ? o4[3] : o4[2];
                const double xv = (ar & 2) ? hi23 : lo01;
                const int i = row0 + ar;                             // block row of this lane's value, vector h & 3
                sX[RT_SMAX + i][h & 3] = (i >= blk_lo && i < TB) ? xv : 0.0;
            }
        }
        __syncthreads();                                             // ---- B: X published
        if (plive && wid == 0) {
            // H = V^T X over the live indices (lane l: slots l, l + 64, ..), then M = T^T H T / 2
            double hp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) hp[i] = 0.0;
#pragma unroll
            for (int q = 0; q < RT_TMAX / 64; ++q) {
                const int sl = 64 * q + lane;
                const bt_d2 v01 = *reinterpret_cast<const bt_d2 *>(&sv[sl][0]), v23 = *reinterpret_cast<const bt_d2 *>(&sv[sl][2]);
                const bt_d2 x01 = *reinterpret_cast<const bt_d2 *>(&sX[sl][0]), x23 = *reinterpret_cast<const bt_d2 *>(&sX[sl][2]);
                const double vv[4] = {v01.x, v01.y, v23.x, v23.y}, xx[4] = {x01.x, x01.y, x23.x, x23.y};
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) hp[4 * k1 + k2] = fma(vv[k1], xx[k2], hp[4 * k1 + k2]);
            }
            double o4[4];
            bt_reduce16(hp, h, o4);                                  // o4[k1]: DPP-row sum of H[k1][h & 3]
            if (h < 4) {
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) part[0][gg][4 * k1 + h] = o4[k1];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) sH[lane] = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) {
                const int ia = lane >> 2, ib = lane & 3;             // M[ia][ib] = 1/2 sum_{k,l} T[k][ia] H[k][l] T[l][ib]
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double t = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) t = fma(sH[4 * k + l], sT[l][ib], t);
                    s = fma(sT[k][ia], t, s);
                }
                sM[ia][ib] = 0.5 * s;
            }
        }
        __syncthreads();                                             // ---- C: M published
        if (tid < RT_TMAX) {                                         // Z_i = X_i T - V_i M, in place of X (dead / padding slots: zero)
            const int gidx = tid - OFF;
            double z[4] = {0.0, 0.0, 0.0, 0.0};
            if (plive && gidx >= lo && gidx < T) {
                const double xx[4] = {sX[tid][0], sX[tid][1], sX[tid][2], sX[tid][3]};
                const double vv[4] = {sv[tid][0], sv[tid][1], sv[tid][2], sv[tid][3]};
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
            for (int k = 0; k < 4; ++k) sX[tid][k] = z[k];
        }
        __syncthreads();                                             // ---- D: Z published
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
                    for (int q = 0; q < 2; ++q) {
                        const bt_d2 v01 = *reinterpret_cast<const bt_d2 *>(&sv[OFF + cc + q][0]), v23 = *reinterpret_cast<const bt_d2 *>(&sv[OFF + cc + q][2]);
                        const bt_d2 z01 = *reinterpret_cast<const bt_d2 *>(&sX[OFF + cc + q][0]), z23 = *reinterpret_cast<const bt_d2 *>(&sX[OFF + cc + q][2]);
                        vc[q][0] = v01.x; vc[q][1] = v01.y; vc[q][2] = v23.x; vc[q][3] = v23.y;
                        zc[q][0] = z01.x; zc[q][1] = z01.y; zc[q][2] = z23.x; zc[q][3] = z23.y;
                    }
                    for (int g = g0; g < S / BT_W; g += RT_NW) {
#pragma unroll
                        for (int ar = 0; ar < 4; ++ar) {
                            const int r = BT_W * g + ar;
                            double *__restrict__ row = strip + r * LDT;
                            const bt_d2 zr01 = *reinterpret_cast<const bt_d2 *>(&sX[OFF + r][0]), zr23 = *reinterpret_cast<const bt_d2 *>(&sX[OFF + r][2]);
                            const bt_d2 vr01 = *reinterpret_cast<const bt_d2 *>(&sv[OFF + r][0]), vr23 = *reinterpret_cast<const bt_d2 *>(&sv[OFF + r][2]);
                            const double zr[4] = {zr01.x, zr01.y, zr23.x, zr23.y}, vr[4] = {vr01.x, vr01.y, vr23.x, vr23.y};
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
            if (wlive) {
                // the wave that factors the next panel is the critical path: its update goes first
                if (!in_strip && wid == ((blk_lo) >> 4)) __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int kh = 0; kh < 4; kh += 2) {                  // two vectors at a time (registers)
                    double zr[RT_R][2], vr[RT_R][2];
#pragma unroll
                    for (int r = 0; r < RT_R; ++r) {
                        const bt_d2 zz = *reinterpret_cast<const bt_d2 *>(&sX[RT_SMAX + row0 + r][kh]);
                        const bt_d2 vv = *reinterpret_cast<const bt_d2 *>(&sv[RT_SMAX + row0 + r][kh]);
                        zr[r][0] = zz.x; zr[r][1] = zz.y; vr[r][0] = vv.x; vr[r][1] = vv.y;
                    }
#pragma unroll
                    for (int j = 0; j < RT_C; ++j) {
                        const bt_d2 zc = *reinterpret_cast<const bt_d2 *>(&sX[RT_SMAX + c0 + j][kh]);
                        const bt_d2 vc = *reinterpret_cast<const bt_d2 *>(&sv[RT_SMAX + c0 + j][kh]);
#pragma unroll
                        for (int r = 0; r < RT_R; ++r) {
                            double e = a[r][j];
                            e = fma(-zr[r][0], vc.x, e);
                            e = fma(-vr[r][0], zc.x, e);
                            e = fma(-zr[r][1], vc.y, e);
                            e = fma(-vr[r][1], zc.y, e);
                            a[r][j] = e;
                        }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        ++pc;
    };

    // ------------------------------------------------------------------------------------------------------------
    // strip panels: first = 0, 4, .. < S.  The panel's rows are strip rows of group first / 4, written by their owner wave itself
    // ------------------------------------------------------------------------------------------------------------
    for (int first = 0; first < S; first += BT_W) {
        if (first + BT_W >= T - 1) break;                            // (nothing below the band any more)
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
            auto pick = [&](const double (&x)[4], int c) {           // entry at tail-global column c (uniform)
                const int l = (c < 128 ? c : c - 128) >> 1;
                const double lo2 = (c & 1) ? x[1] : x[0], hi2 = (c & 1) ? x[3] : x[2];
                return lane_get(c < 128 ? lo2 : hi2, l);
            };
            house4(w, tau, colq, std::integral_constant<int, 4>{}, first + BT_W, T, OFF, T, pick);
            double (*const sv)[BT_W] = sV[pc & 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = colq[q];
                if (c < T + (T & 1)) {                               // (the pad column of an odd T holds zeros)
                    *reinterpret_cast<bt_d2 *>(&sv[OFF + c][0]) = bt_d2{w[0][q], w[1][q]};
                    *reinterpret_cast<bt_d2 *>(&sv[OFF + c][2]) = bt_d2{w[2][q], w[3][q]};
                }
#pragma unroll
                for (int j = 0; j < BT_W; ++j)
                    if (c < T) P.V[(long)(first + j) * n + c] = w[j][q];
            }
            if (lane < BT_W) stau[OFF + first + lane] = tau[lane];
            __builtin_amdgcn_wave_barrier();
            emit_band(first, OFF + first, T);
            __builtin_amdgcn_s_setprio(0);
        }
        panel_rest(first, true);
    }
    if (S > 0) {                                                     // the reflectors of the block panels are zero over the strip
        __syncthreads();
        for (int idx = tid; idx < 2 * RT_SMAX * BT_W; idx += RT_NTH) {
            const int bsel = idx / (RT_SMAX * BT_W), rem = idx - bsel * RT_SMAX * BT_W;
            (&sV[bsel][0][0])[rem] = 0.0;
        }
        for (int idx = tid; idx < RT_SMAX * BT_W; idx += RT_NTH) (&sX[0][0])[idx] = 0.0;
    }

    // ------------------------------------------------------------------------------------------------------------
    // block panels (block-local first column kk = 0, 4, ..): the panel's rows are tile rows of one row group of wave kk >> 4
    // ------------------------------------------------------------------------------------------------------------
    int kk = 0;
    for (; kk + BT_W < TB - 1; kk += BT_W) {
        if (wid == (kk >> 4)) {
            __builtin_amdgcn_s_setprio(3);
            if (gg == ((kk >> 2) & 3)) {
#pragma unroll
                for (int r = 0; r < RT_R; ++r)
#pragma unroll
                    for (int j = 0; j < RT_C; j += 2) *reinterpret_cast<bt_d2 *>(&pan[r][RT_SMAX + c0 + j]) = bt_d2{a[r][j], a[r][j + 1]};
            }
            __builtin_amdgcn_wave_barrier();
            const int colq[3] = {lane, 64 + lane, 128 + lane};
            double w[BT_W][3], tau[BT_W];
#pragma unroll
            for (int j = 0; j < BT_W; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) w[j][q] = pan[j][RT_SMAX + 64 * q + lane];
            auto pick = [&](const double (&x)[3], int c) {
                const int q = c >> 6;
                const double t = (q == 0) ? x[0] : (q == 1) ? x[1] : x[2];
                return lane_get(t, c & 63);
            };
            __builtin_amdgcn_wave_barrier();
            house4(w, tau, colq, std::integral_constant<int, 3>{}, kk + BT_W, TB, RT_SMAX, RT_T, pick);
            double (*const sv)[BT_W] = sV[pc & 1];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int c = 64 * q + lane;
                *reinterpret_cast<bt_d2 *>(&sv[RT_SMAX + c][0]) = bt_d2{w[0][q], w[1][q]};
                *reinterpret_cast<bt_d2 *>(&sv[RT_SMAX + c][2]) = bt_d2{w[2][q], w[3][q]};
#pragma unroll
                for (int j = 0; j < BT_W; ++j)
                    if (c < TB) P.V[(long)(S + kk + j) * n + S + c] = w[j][q];
            }
            if (lane < BT_W) stau[RT_SMAX + kk + lane] = tau[lane];
            __builtin_amdgcn_wave_barrier();
            emit_band(S + kk, RT_SMAX + kk, T);
            __builtin_amdgcn_s_setprio(0);
        }
        panel_rest(S + kk, false);
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
    if (b.clk && threadIdx.x == 0) b.clk[2 * blockIdx.x + 1] = wall_clock64();
}

}  // namespace gpcsd
