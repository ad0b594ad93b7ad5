// Single-workgroup tail of the tridiagonalisation, 512-thread layout (included by eigh_dc.hip after sytrd_regtail.hpp).
// EXPERIMENT, not the default (GPCSD_TAIL_V=6 selects it): correct on every size the default handles, but 8 % slower per
// launch at T = 192 / 250 on MI355X -- see the note at its launch site.
//
// Same algorithm, strip and LDS vector layout as sytrd_rtail_kernel (sytrd_regtail.hpp), but the 192 x 192 register block is
// tiled 6 x 12 per thread over 512 threads (32 row groups x 16 column parts; a wave = 4 row groups = 24 rows) instead of
// 4 x 12 over 768.  Two waves per SIMD may use 256 VGPRs each, so besides the 144-VGPR tile the twelve v entries of the
// thread's columns stay in registers between y = A v and the rank-2 update.  The column costs the same VALU issue slots
// (the FMA count is fixed) but a third fewer LDS instructions per element (the 768-thread kernel spends ~25 % of its time on
// the LDS pipe: every wave re-reads its v / y entries as ds_read_b128, 8 cycles each whatever the duplication) and one
// third fewer waves at the barriers.
#pragma once
#include <type_traits>

namespace gpcsd {

constexpr int R6_R = 6, R6_C = 12, R6_NTH = (RT_T / R6_R) * (RT_T / R6_C), R6_NW = R6_NTH / 64, R6_WR = 4 * R6_R;
static_assert(RT_T / R6_C == 16 && R6_NTH == 512 && R6_NW == 8 && R6_WR == 24, "one DPP row per row group");

__global__ __launch_bounds__(R6_NTH) void sytrd_rtail6_kernel(SytrdBatch b) {
    const SytrdProb &P = b.p[blockIdx.x];
    const int n = P.n, k0 = P.k_tail;
    if (k0 >= n - 1) return;
    const int T = n - k0;                          // rows / columns k0 .. n-1, T <= RT_TMAX
    const int S = rt_strip_rows(T), LDT = rt_strip_ld(T);
    const int TB = T - S;                          // live rows of the register block, <= RT_T
    const int OFF = RT_SMAX - S;                   // slot of tail-global index 0 in the LDS vectors
    extern __shared__ __attribute__((aligned(16))) double strip[];   // [S][LDT]
    __shared__ __attribute__((aligned(16))) double sx[RT_T], sv2[2][RT_TMAX], sy[RT_TMAX];
    __shared__ __attribute__((aligned(16))) double red[R6_NW];
    __shared__ double sd[RT_TMAX], se[RT_TMAX], st[RT_TMAX];
    __shared__ double s_r, s_u1;                   // 1 / ||column|| (0: H = I) and |u_1| of the current reflector
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gg = lane >> 4, h = lane & 15;       // row group inside the wave, column part
    const int row0 = R6_WR * wid + R6_R * gg, c0 = R6_C * h;
    const double *__restrict__ Ain = (k0 & 1) ? P.A1 : P.A0;
    const double *__restrict__ yin = (k0 & 1) ? P.y1 : P.y0;

    // pending rank-2 update of step k0-1 (its reflector and y = A v are in global memory): v -> sv2[0], w -> sy
    {
        double pv = 0.0, py = 0.0, taup = 0.0;
        const int gslot = tid - OFF;                                 // thread tid fills slot tid
        if (k0 > 0) {
            taup = P.tau[k0 - 1];
            if (tid < RT_TMAX && gslot >= 0 && gslot < T) {
                pv = P.V[(long)(k0 - 1) * n + k0 + gslot];
                py = yin[k0 + gslot];
            }
        }
        const double part = wave_sum(pv * py);
        if (lane == 0) red[wid] = part;
        __syncthreads();
        double dot = 0.0;
#pragma unroll
        for (int q = 0; q < R6_NW; ++q) dot += red[q];
        const double cc = 0.5 * taup * taup * dot;
        if (tid < RT_TMAX) {
            sv2[0][tid] = pv;
            sv2[1][tid] = 0.0;                                       // the strip columns never write the padding slots
            sy[tid] = taup * py - cc * pv;
        }
        __syncthreads();
    }
    double a[R6_R][R6_C];
    {
        const double *svp = sv2[0] + RT_SMAX, *syp = sy + RT_SMAX;
#pragma unroll
        for (int r = 0; r < R6_R; ++r) {
            const int i = row0 + r;
            const bool rok = i < TB;
            const double *__restrict__ arow = Ain + (long)(k0 + S + (rok ? i : 0)) * n + k0 + S;
            const double vi = rok ? svp[i] : 0.0, wi = rok ? syp[i] : 0.0;
#pragma unroll
            for (int j = 0; j < R6_C; ++j) {
                const int c = c0 + j;
                const bool ok = rok && c < TB;
                const double g = ok ? arow[c] : 0.0;
                a[r][j] = ok ? g - vi * syp[c] - wi * svp[c] : 0.0;
            }
        }
        // the pad columns (c >= T) are read by the column sums of padding rows: they must hold zeros, not stale LDS
        for (int idx = tid; idx < S * LDT + 8; idx += R6_NTH) {
            const int r = idx / LDT, c = idx - r * LDT;
            double v = 0.0;
            if (r < S && c < T)
                v = Ain[(long)(k0 + r) * n + k0 + c] - sv2[0][OFF + r] * sy[OFF + c] - sy[OFF + r] * sv2[0][OFF + c];
            strip[idx] = v;
        }
    }
    __syncthreads();
    if (tid < RT_TMAX) sy[tid] = 0.0;              // from here on sy is y = A v: zero on dead rows and beyond T
    __syncthreads();

    // Six row sums over the 16 lanes of a row group as a reduce-scatter.  Exchange 1 (lane ^ 1): a lane keeps rows 0,1,2 or
    // 3,4,5; exchange 2 (lane ^ 2): lanes with bit 1 clear keep two of those, the others one; two rotations add the four
    // quads.  9 DPP transfers instead of the 24 of six butterflies.  On return A is the sum of row rowA(h) and, in lanes
    // with bit 1 clear, B the sum of row rowA(h) + 1:   h & 3 = 0: rows 0, 1;  1: rows 3, 4;  2: row 2;  3: row 5.
    const bool hb0 = h & 1, hb1 = h & 2;
    const int rowA = (hb0 ? 3 : 0) + (hb1 ? 2 : 0);
    auto reduce_rows = [&](const double (&acc)[R6_R], double &A, double &B) {
        const double s0 = (hb0 ? acc[3] : acc[0]) + dpp_mov<0xB1>(hb0 ? acc[0] : acc[3]);
        const double s1 = (hb0 ? acc[4] : acc[1]) + dpp_mov<0xB1>(hb0 ? acc[1] : acc[4]);
        const double s2 = (hb0 ? acc[5] : acc[2]) + dpp_mov<0xB1>(hb0 ? acc[2] : acc[5]);
        A = (hb1 ? s2 : s0) + dpp_mov<0x4E>(hb1 ? s0 : s2);
        B = s1 + dpp_mov<0x4E>(s1);
        A += dpp_mov<0x124>(A);                                      // row_ror:4
        B += dpp_mov<0x124>(B);
        A += dpp_mov<0x128>(A);                                      // row_ror:8
        B += dpp_mov<0x128>(B);
    };
    // publish the sums of this row group (lanes h < 4), return this lane's part of v . y
    auto publish_y = [&](double A, double B, const double *svb, double *syb, int kk_local) {
        const int rA = row0 + rowA, rB = rA + 1;
        A = (rA > kk_local) ? A : 0.0;
        B = (rB > kk_local) ? B : 0.0;
        double dp = 0.0;
        if (h < 4) {
            syb[rA] = A;
            dp = svb[rA] * A;
            if (h < 2) {
                syb[rB] = B;
                dp = fma(svb[rB], B, dp);
            }
        }
        dp += dpp_mov<0xB1>(dp);                                     // lanes 0..3 of the DPP row
        dp += dpp_mov<0x4E>(dp);
        return (lane_get(dp, 0) + lane_get(dp, 16)) + (lane_get(dp, 32) + lane_get(dp, 48));
    };
    auto dot_tree = [&]() {
        static_assert(R6_NW == 8, "reduction tree below");
        double pr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 t2 = *reinterpret_cast<const double2 *>(red + 2 * q);
            pr[q] = t2.x + t2.y;
        }
        return (pr[0] + pr[1]) + (pr[2] + pr[3]);
    };
    // rank-2 update of the tile: a -= v_r w_c + w_r v_c with w = tau y - cc v; vv = the thread's v columns (kept from A v)
    auto update_tile = [&](const double (&vv)[R6_C], const double *svb, const double *syb, double tau, double cc) {
        double wrow[R6_R], vrow[R6_R];
#pragma unroll
        for (int r = 0; r < R6_R; r += 2) {
            const double2 tv = *reinterpret_cast<const double2 *>(svb + row0 + r);
            const double2 ty = *reinterpret_cast<const double2 *>(syb + row0 + r);
            vrow[r] = tv.x;
            vrow[r + 1] = tv.y;
            wrow[r] = tau * ty.x - cc * tv.x;
            wrow[r + 1] = tau * ty.y - cc * tv.y;
        }
#pragma unroll
        for (int j = 0; j < R6_C; j += 2) {
            const double2 yy = *reinterpret_cast<const double2 *>(syb + c0 + j);
            const double w0 = tau * yy.x - cc * vv[j], w1 = tau * yy.y - cc * vv[j + 1];
#pragma unroll
            for (int r = 0; r < R6_R; ++r) {
                a[r][j] = fma(-wrow[r], vv[j], fma(-vrow[r], w0, a[r][j]));
                a[r][j + 1] = fma(-wrow[r], vv[j + 1], fma(-vrow[r], w1, a[r][j + 1]));
            }
        }
    };

    // ------------------------------------------------------------------------------------------------------------
    // strip columns k = 0 .. S-1: every block row is live, the matrix is strip rows (k, S) + the block
    // ------------------------------------------------------------------------------------------------------------
    for (int k = 0; k < S; ++k) {
        double *sv = sv2[k & 1] + OFF;                               // tail-global view of the vectors in this phase
        double *syg = sy + OFF;
        if (wid == k % R6_NW) {                                      // gen: this wave wrote strip row k itself
            __builtin_amdgcn_s_setprio(3);
            const double *__restrict__ row = strip + k * LDT;
            double x[4], part = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lane + 64 * q;
                x[q] = (c < T) ? row[c] : 0.0;
                const double m = (c >= k + 2) ? x[q] : 0.0;
                part = fma(m, m, part);
            }
            const double xnorm2 = wave_sum(part);
            const double dk = row[k], alpha = row[k + 1];
            double r, u1, beta;
            rt_house(alpha, xnorm2, true, r, u1, beta);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lane + 64 * q;
                double t = (c > k + 1) ? x[q] : 0.0;
                t = (c == k + 1) ? u1 : t;
                if (c < T) sv[c] = (r != 0.0 || c == k + 1) ? t : 0.0;
            }
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
        if (tid < T) P.V[(long)(k0 + k) * n + k0 + tid] = sv[tid];   // reflector k (zeros up to k, u_1 at k+1)
        const double tau = rr * fast_rcp(au);
        if (tid == 0) {
            st[OFF + k] = tau;
            syg[k] = 0.0;                                            // row k is dead from now on (nobody reads y before B)
        }
        const bool live = rr != 0.0;                                 // uniform over the workgroup
        const int rfirst = k + 1 + (wid - (k + 1) % R6_NW + R6_NW) % R6_NW;   // this wave's first strip row > k
        const double *svb = sv + S;
        double *syb = syg + S;
        double vv[R6_C];
        double vq[4];
        if (live) {
            // strip rows of this wave: y_r = strip[r][:] . v
            double dps = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) vq[q] = (lane + 64 * q < T) ? sv[lane + 64 * q] : 0.0;
            for (int r = rfirst; r < S; r += R6_NW) {
                const double *__restrict__ row = strip + r * LDT;
                double p = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = lane + 64 * q;
                    p = fma((c < T) ? row[c] : 0.0, vq[q], p);
                }
                const double yr = wave_sum(p);
                if (lane == 0) syg[r] = yr;
                dps = fma(sv[r], yr, dps);
            }
            // block rows: the tile from registers, plus the strip COLUMNS of these rows (strip row rs, lanes h = rs mod 16)
            double acc[R6_R] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < R6_C; j += 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(svb + c0 + j);
                vv[j] = t2.x;
                vv[j + 1] = t2.y;
#pragma unroll
                for (int r = 0; r < R6_R; ++r) acc[r] = fma(a[r][j + 1], t2.y, fma(a[r][j], t2.x, acc[r]));
            }
            for (int rs = k + 1 + (h - (k + 1) % 16 + 16) % 16; rs < S; rs += 16) {
                const double vr = sv[rs];
                const double *__restrict__ sp = strip + rs * LDT + S + row0;
#pragma unroll
                for (int r = 0; r < R6_R; r += 2) {
                    const double2 s2 = *reinterpret_cast<const double2 *>(sp + r);
                    acc[r] = fma(s2.x, vr, acc[r]);
                    acc[r + 1] = fma(s2.y, vr, acc[r + 1]);
                }
            }
            double A, B;
            reduce_rows(acc, A, B);
            const double dp = publish_y(A, B, svb, syb, -1);
            if (lane == 0) red[wid] = dp + dps;
        } else if (lane == 0) {
            red[wid] = 0.0;
        }
        const double htt = 0.5 * tau * tau;
        __syncthreads();                                             // ---- B: y, v.y partials published
        if (live) {
            const double cc = htt * dot_tree();
            // strip rows of this wave (row k+1 first when it is ours: its owner generates the next reflector from it)
            if (rfirst < S) {
                double wq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = lane + 64 * q;
                    wq[q] = (c < T) ? tau * syg[c] - cc * vq[q] : 0.0;
                }
                for (int r = rfirst; r < S; r += R6_NW) {
                    double *__restrict__ row = strip + r * LDT;
                    const double vr = sv[r], wr = tau * syg[r] - cc * vr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = lane + 64 * q;
                        if (c < T) row[c] = fma(-wr, vq[q], fma(-vr, wq[q], row[c]));
                    }
                }
            }
            update_tile(vv, svb, syb, tau, cc);
        }
    }
    if (S > 0) {                                   // the reflectors of the block columns are zero over the strip
        __syncthreads();
        if (tid < S) {
            sv2[0][OFF + tid] = 0.0;
            sv2[1][OFF + tid] = 0.0;
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // block columns (local index kk = g - S)
    // ------------------------------------------------------------------------------------------------------------
    double *const syb = sy + RT_SMAX, *const sdb = sd + RT_SMAX, *const seb = se + RT_SMAX, *const stb = st + RT_SMAX;
    // One column of the block.  RK = kk mod 6 is a compile-time constant (the column loop below is unrolled by six), so the
    // row handed to the gen section is read straight out of its registers.  S is even: the v buffer (S + kk) & 1 is RK & 1.
    auto column = [&](const int kk, auto RKc) {
        constexpr int RK = decltype(RKc)::value;
        double *const svb = sv2[RK & 1] + RT_SMAX;
        // ---- gen: only the wave owning row kk (wave-uniform branch).  Row kk (= column kk by symmetry) goes to LDS, then
        // all 64 lanes work on three entries each: norm, Householder scalars, v.
        if (wid == kk / R6_WR) {
            __builtin_amdgcn_s_setprio(3);
            if (gg == (kk % R6_WR) / R6_R) {
#pragma unroll
                for (int j = 0; j < R6_C; ++j) sx[c0 + j] = a[RK][j];
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
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int c = 64 * q + lane;
                double t = (c > kk + 1) ? x[q] : 0.0;
                t = (c == kk + 1) ? u1 : t;
                svb[c] = (r != 0.0 || c == kk + 1) ? t : 0.0;
            }
            if (lane == 0) {
                sdb[kk] = dk;
                seb[kk] = beta;
                s_r = r;
                s_u1 = fabs(u1);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();                                             // ---- A: v, scalars published
        const double rr = s_r, au = s_u1;
        if (tid < T) P.V[(long)(k0 + S + kk) * n + k0 + tid] = svb[tid - S];   // reflector S + kk (zeros up to it, u_1 next)
        const bool live = (R6_WR * wid + R6_WR - 1 > kk) && (rr != 0.0);       // wave-uniform: still owns a row > kk
        const double tau = rr * fast_rcp(au);
        double vv[R6_C];
        if (live) {
            double acc[R6_R] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < R6_C; j += 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(svb + c0 + j);
                vv[j] = t2.x;
                vv[j + 1] = t2.y;
#pragma unroll
                for (int r = 0; r < R6_R; ++r) acc[r] = fma(a[r][j + 1], t2.y, fma(a[r][j], t2.x, acc[r]));
            }
            double A, B;
            reduce_rows(acc, A, B);
            const double dp = publish_y(A, B, svb, syb, kk);
            if (lane == 0) red[wid] = dp;
        } else if (lane == 0) {
            red[wid] = 0.0;
        }
        if (tid == 0) stb[kk] = tau;
        const double htt = 0.5 * tau * tau;
        __syncthreads();                                             // ---- B: y, v.y partials published
        if (live) {
            // the wave that generates the next reflector is the critical path of the column: its update goes first
            if (wid == (kk + 1) / R6_WR) __builtin_amdgcn_s_setprio(3);
            const double cc = htt * dot_tree();
            update_tile(vv, svb, syb, tau, cc);
        }
    };
    for (int kb = 0; kb < TB - 1; kb += 6) {                         // TB is the same for every thread: uniform control flow
        column(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < TB - 1) column(kb + 1, std::integral_constant<int, 1>{});
        if (kb + 2 < TB - 1) column(kb + 2, std::integral_constant<int, 2>{});
        if (kb + 3 < TB - 1) column(kb + 3, std::integral_constant<int, 3>{});
        if (kb + 4 < TB - 1) column(kb + 4, std::integral_constant<int, 4>{});
        if (kb + 5 < TB - 1) column(kb + 5, std::integral_constant<int, 5>{});
    }
    // last diagonal element a[TB-1][TB-1] (run-time row pick, once)
    {
        const int row = TB - 1;
        if (wid == row / R6_WR) {
            const int rk = row % R6_R;
            if (gg == (row % R6_WR) / R6_R) {
#pragma unroll
                for (int j = 0; j < R6_C; ++j) {
                    double x = a[0][j];
#pragma unroll
                    for (int r = 1; r < R6_R; ++r) x = (rk == r) ? a[r][j] : x;
                    sx[c0 + j] = x;
                }
            }
            if (lane == 0) {
                sdb[row] = sx[row];
                seb[row] = 0.0;
                stb[row] = 0.0;
            }
        }
    }
    __syncthreads();
    for (int kk = tid; kk < T; kk += R6_NTH) {
        P.d[k0 + kk] = sd[OFF + kk];
        P.e[k0 + kk] = se[OFF + kk];
        P.tau[k0 + kk] = st[OFF + kk];
    }
}

}  // namespace gpcsd
