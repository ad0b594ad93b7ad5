// Register-resident tail of the Householder tridiagonalisation (included by eigh_dc.hip after SytrdBatch).
//
// The trailing block (T <= 192 rows) of every problem lives in the VGPRs of ONE workgroup of 768 threads: thread
// (row group g, column part h) keeps the 4 x 12 tile rows 4g..4g+3, columns 12h..12h+11 (96 VGPRs).  A wave is four
// row groups x sixteen column parts, i.e. one 16-lane DPP row per row group, so the row sums of y = A v fold with four
// DPP stages and never touch LDS.  Per column:
//   gen   the wave that owns row kk copies that row (= column kk by symmetry) out of its registers, forms the
//         Householder vector and publishes v (length T) -- a wave-local section that overlaps the other waves' rank-2
//         update of the previous column;
//   A     barrier;  y = A v from registers (48 FMAs + 4 DPP folds per thread), v.y partials;
//   B     barrier;  w = tau y - cc v formed on the fly, rank-2 update of the tile (96 FMAs per thread).
// Two barriers and ~0.3 k VALU instructions per thread and column instead of a dependent launch (4.6 us) or an LDS-resident
// sweep (3.1 us at T = 113).  Dead rows / columns need no masks in the FMA loops: v and y are zero there.
#pragma once
#include <type_traits>

namespace gpcsd {

constexpr int RT_R = 4, RT_C = 12, RT_T = 192, RT_NTH = (RT_T / RT_R) * (RT_T / RT_C), RT_NW = RT_NTH / 64;
static_assert(RT_T / RT_C == 16 && RT_NTH == 768, "one DPP row per row group");

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

__global__ __launch_bounds__(RT_NTH) void sytrd_rtail_kernel(SytrdBatch b) {
    const SytrdProb &P = b.p[blockIdx.x];
    const int n = P.n, k0 = P.k_tail;
    if (k0 >= n - 1) return;
    const int T = n - k0;                          // rows / columns k0 .. n-1, T <= RT_T
    __shared__ __attribute__((aligned(16))) double sx[RT_T], sv2[2][RT_T], sy[RT_T];
    __shared__ __attribute__((aligned(16))) double red[RT_NW];
    __shared__ double sd[RT_T], se[RT_T], st[RT_T];
    __shared__ double s_r, s_u1;                   // 1 / ||column|| (0: H = I) and |u_1| of the current reflector
    // v is double-buffered: the wave generating reflector kk+1 writes it while slower waves still read v of column kk
    double *sv = sv2[0];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gg = lane >> 4, h = lane & 15;       // row group inside the wave, column part
    const int row0 = 16 * wid + 4 * gg, c0 = RT_C * h;
    const double *__restrict__ Ain = (k0 & 1) ? P.A1 : P.A0;
    const double *__restrict__ yin = (k0 & 1) ? P.y1 : P.y0;

    // pending rank-2 update of step k0-1 (its reflector and y = A v are in global memory): v -> sv, w -> sy
    {
        double pv = 0.0, py = 0.0, taup = 0.0;
        if (k0 > 0) {
            taup = P.tau[k0 - 1];
            if (tid < T) {
                pv = P.V[(long)(k0 - 1) * n + k0 + tid];
                py = yin[k0 + tid];
            }
        }
        const double part = wave_sum(pv * py);
        if (lane == 0) red[wid] = part;
        __syncthreads();
        double dot = 0.0;
#pragma unroll
        for (int q = 0; q < RT_NW; ++q) dot += red[q];
        const double cc = 0.5 * taup * taup * dot;
        if (tid < RT_T) {
            sv[tid] = pv;
            sy[tid] = taup * py - cc * pv;
        }
        __syncthreads();
    }
    double a[RT_R][RT_C];
#pragma unroll
    for (int r = 0; r < RT_R; ++r) {
        const int i = row0 + r;
        const bool rok = i < T;
        const double *__restrict__ arow = Ain + (long)(k0 + (rok ? i : 0)) * n + k0;
        const double vi = sv[i], wi = sy[i];
#pragma unroll
        for (int j = 0; j < RT_C; ++j) {
            const int c = c0 + j;
            const bool ok = rok && c < T;
            const double g = ok ? arow[c] : 0.0;
            a[r][j] = ok ? g - vi * sy[c] - wi * sv[c] : 0.0;
        }
    }
    __syncthreads();

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

    // One column of the reduction.  RK = kk mod 4 is a compile-time constant (the column loop below is unrolled by four), so
    // the row handed to the gen section is read straight out of its registers: picking it with selects on a run-time
    // index cost 72 v_cndmask per column on the critical path.
    auto column = [&](const int kk, auto RKc) {
        constexpr int RK = decltype(RKc)::value;
        sv = sv2[RK & 1];                                            // = kk & 1: a constant LDS offset, no address register
        // ---- gen: only the wave owning row kk (wave-uniform branch).  Row kk (= column kk by symmetry) goes to LDS, then
        // all 64 lanes work on three entries each: norm, Householder scalars, v.  Every other wave waits for this section
        // at barrier A, so it is kept short (~200 instructions) and issues ahead of the waves sharing its SIMD.
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
            const int m = T - kk - 1;
            // Reflector H = I - tau u u^T with u = (alpha - beta, x_2, ..) left UN-normalised: u is known as soon as
            // s = sqrt(alpha^2 + |x|^2) is, and tau = 1 / (s (|alpha| + s)) is formed by every wave after the barrier, off
            // this chain.  A dependent fp64 operation costs ~40 cycles here, so the chain is counted in operations: s and
            // 1/s come out of one coupled Newton (Goldschmidt) iteration on the hardware rsq seed, 6 deep, instead of
            // the ~25 of an IEEE sqrt and two divisions.  The matrix is scaled to max|a| = 1, so s^2 < 1e-290 is a zero column.
            const double s2 = fma(alpha, alpha, xnorm2);
            double r = 0.0, u1 = 1.0, beta = alpha;
            if (m >= 2 && xnorm2 > 0.0 && s2 > 1e-290) {             // wave-uniform
                const double y0 = __builtin_amdgcn_rsq(s2);
                double g = s2 * y0, hh = 0.5 * y0;
                double e = fma(-hh, g, 0.5);
                g = fma(g, e, g);
                hh = fma(hh, e, hh);
                e = fma(-hh, g, 0.5);
                g = fma(g, e, g);                                    // sqrt(s2)
                hh = fma(hh, e, hh);                                 // 1 / (2 sqrt(s2))
                beta = -copysign(g, alpha);
                u1 = alpha - beta;                                   // sign(alpha) (|alpha| + s): no cancellation
                r = hh + hh;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int c = 64 * q + lane;
                double t = (c > kk + 1) ? x[q] : 0.0;
                t = (c == kk + 1) ? u1 : t;
                sv[c] = (r != 0.0 || c == kk + 1) ? t : 0.0;
            }
            if (lane == 0) {
                sd[kk] = dk;
                se[kk] = beta;
                s_r = r;
                s_u1 = fabs(u1);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();                                             // ---- A: v, tau published
        const double rr = s_r, au = s_u1;
        if (tid < T) P.V[(long)(k0 + kk) * n + k0 + tid] = sv[tid];  // reflector kk (zeros up to kk, u_1 at kk+1)
        const bool live = (16 * wid + 15 > kk) && (rr != 0.0);       // wave-uniform: still owns a row > kk
        double tau;
        // v is re-read from LDS pair by pair in both phases (6 ds_read_b128 each) instead of being held in 24 VGPRs:
        // the 4 x 12 tile already takes 96 of the 168 registers a thread may use at three waves per SIMD
        if (live) {
            const int myrow = row0 + 2 * (h & 1) + ((h >> 1) & 1);   // the row whose sum this lane ends up with
            const double vmy = sv[myrow];
            tau = rr * fast_rcp(au);                                 // independent of the sums below: interleaves with them
            double acc[RT_R] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) {
                const double2 vv = *reinterpret_cast<const double2 *>(sv + c0 + j);
#pragma unroll
                for (int r = 0; r < RT_R; ++r) acc[r] = fma(a[r][j + 1], vv.y, fma(a[r][j], vv.x, acc[r]));
            }
            // Four row sums over the 16 lanes of the row group as a reduce-scatter: each exchange halves the number of
            // sums a lane still carries (2 + 1 DPP adds), two rotations finish them: 27 instructions instead of the 48 of
            // four full butterflies.  Lane h ends with the sum of row rl = 2*(h&1) + ((h>>1)&1), replicated in its 4 quads.
            const bool b0 = h & 1, b1 = h & 2;
            const double t0 = (b0 ? acc[2] : acc[0]) + dpp_mov<0xB1>(b0 ? acc[0] : acc[2]);
            const double t1 = (b0 ? acc[3] : acc[1]) + dpp_mov<0xB1>(b0 ? acc[1] : acc[3]);
            double y = (b1 ? t1 : t0) + dpp_mov<0x4E>(b1 ? t0 : t1);
            y += dpp_mov<0x124>(y);                                  // row_ror:4
            y += dpp_mov<0x128>(y);                                  // row_ror:8
            y = (myrow > kk) ? y : 0.0;
            if (h < 4) sy[myrow] = y;                                // one lane per row publishes y ..
            double dp = (h < 4) ? vmy * y : 0.0;                     // .. and carries its v.y term
            dp += dpp_mov<0xB1>(dp);                                 // the four rows of the group (lanes 0..3 of the DPP row)
            dp += dpp_mov<0x4E>(dp);
            dp = (lane_get(dp, 0) + lane_get(dp, 16)) + (lane_get(dp, 32) + lane_get(dp, 48));
            if (lane == 0) red[wid] = dp;
        } else {
            tau = rr * fast_rcp(au);
            if (lane == 0) red[wid] = 0.0;
        }
        if (tid == 0) st[kk] = tau;
        const double htt = 0.5 * tau * tau;
        __syncthreads();                                             // ---- B: y, v.y partials published
        if (live) {
            // v.y over the twelve waves as a tree: a serial sum is twelve dependent adds (~0.4 k cycles) in every wave
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
                const double2 tv = *reinterpret_cast<const double2 *>(sv + row0 + r);
                const double2 ty = *reinterpret_cast<const double2 *>(sy + row0 + r);
                vrow[r] = tv.x;
                vrow[r + 1] = tv.y;
                wrow[r] = tau * ty.x - cc * tv.x;
                wrow[r + 1] = tau * ty.y - cc * tv.y;
            }
#pragma unroll
            for (int j = 0; j < RT_C; j += 2) {
                const double2 yy = *reinterpret_cast<const double2 *>(sy + c0 + j);
                const double2 vv = *reinterpret_cast<const double2 *>(sv + c0 + j);
                const double w0 = tau * yy.x - cc * vv.x, w1 = tau * yy.y - cc * vv.y;
#pragma unroll
                for (int r = 0; r < RT_R; ++r) {
                    a[r][j] = fma(-wrow[r], vv.x, fma(-vrow[r], w0, a[r][j]));
                    a[r][j + 1] = fma(-wrow[r], vv.y, fma(-vrow[r], w1, a[r][j + 1]));
                }
            }
        }
    };
    for (int kb = 0; kb < T - 1; kb += 4) {                          // T is the same for every thread: uniform control flow
        column(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < T - 1) column(kb + 1, std::integral_constant<int, 1>{});
        if (kb + 2 < T - 1) column(kb + 2, std::integral_constant<int, 2>{});
        if (kb + 3 < T - 1) column(kb + 3, std::integral_constant<int, 3>{});
    }
    // last diagonal element a[T-1][T-1]
    if (wid == ((T - 1) >> 4)) {
        publish_row(T - 1);
        if (lane == 0) {
            sd[T - 1] = sx[T - 1];
            se[T - 1] = 0.0;
            st[T - 1] = 0.0;
        }
    }
    __syncthreads();
    for (int kk = tid; kk < T; kk += RT_NTH) {
        P.d[k0 + kk] = sd[kk];
        P.e[k0 + kk] = se[kk];
        P.tau[k0 + kk] = st[kk];
    }
}

}  // namespace gpcsd
