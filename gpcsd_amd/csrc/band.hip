// Shifted BANDED systems of the basis U (x) Q (round 5): the counterparts of gram.hip's ll_tridiag_* / tridiag_solve_kernel for a
// temporal side that was reduced to half-bandwidth 4 instead of a tridiagonal matrix (sytrd_bandtail.hpp).
//
// With Kt_p = m_p Q_p B_p Q_p^T (B_p banded: bd[j * ld + k] = B[k + j][k], j = 0 .. 4) the block of Ks (x) Kt + sig2 I of spatial
// eigen-row x' and temporal parity p is Q_p (lam B_p + sig2 I) Q_p^T with lam = es[x'] m_p: symmetric positive definite (lam >= 0,
// sig2 > 0) with the band of B_p, so A = L D L^T without pivoting, L unit lower triangular with four sub-diagonals.  The
// log-likelihood (gpcsd1d.py:113-128) needs sum log D_k and z^T D^-1 z with z = L^-1 w per row w of W = U^T Y Q; the posterior mean
// (gpcsd1d.py:262-265) the solutions x = L^-T D^-1 L^-1 w.  Same algebra as the tridiagonal form, four multipliers per column
// instead of one.
//
// Factorisation (band_factor): a serial recurrence over the columns -- per column ten FMAs, one reciprocal -- that every lane of
// a wave walks redundantly (one item per wave, all items of a launch in parallel); its dependent chain is reciprocal ->
// multiplier -> next pivot.  Sweeps: lane = trial; a column step is four FMAs of which ONE is on the dependent chain (the terms
// with the older values are summed first).  Coefficients of a column: (l1, l2, l3, l4, 1 / D, D) = L[k+1..k+4][k], the pivot's
// reciprocal and the pivot.
#include "devutil.hpp"
#include "kernels.hpp"

namespace gpcsd {

constexpr int BD_W = 4;                     // half-bandwidth (sytrd_bandtail.hpp: BT_W)
constexpr int BD_NC = 6;                    // doubles per column of a factor
constexpr int BD_KMAX = 256;                // columns of a block at most
typedef double bd_d2 __attribute__((ext_vector_type(2)));

// In: cf[k][0..4] = the band of A (diagonal, four sub-diagonals) of column k, k < np.  Out: cf[k] = (l1, l2, l3, l4, 1 / D_k, D_k).
// Called by every lane of one wave with the same arguments (cf in LDS, wave-private); lane 0 stores.
__device__ __forceinline__ void band_factor(double (*cf)[BD_NC], int np, int lane) {
    // history of the last four columns c = k-1 .. k-4:  l_c[i] = L[c+i][c],  u_c[i] = l_c[i] D_c
    double l1[4] = {0, 0, 0, 0}, l2[4] = {0, 0, 0, 0}, l3[4] = {0, 0, 0, 0}, l4[4] = {0, 0, 0, 0};
    double u1[4] = {0, 0, 0, 0}, u2[4] = {0, 0, 0, 0}, u3[4] = {0, 0, 0, 0}, u4[4] = {0, 0, 0, 0};
    // (index [i-1]: sub-diagonal i;  l1 = column k-1, l2 = column k-2, ..)
    for (int k = 0; k < np; ++k) {
        const bd_d2 a01 = *reinterpret_cast<const bd_d2 *>(&cf[k][0]), a23 = *reinterpret_cast<const bd_d2 *>(&cf[k][2]);
        const double a4 = cf[k][4];
        // row k of L: L[k][k-j] = l_{k-j}[j];  row k+i: L[k+i][k-j] = l_{k-j}[i+j]
        // the terms of the older columns first, the newest column (whose multipliers come out of the previous step's reciprocal) last
        double dk = a01.x;
        dk = fma(-l4[3], u4[3], dk);
        dk = fma(-l3[2], u3[2], dk);
        dk = fma(-l2[1], u2[1], dk);
        dk = fma(-l1[0], u1[0], dk);
        double n1 = a01.y, n2 = a23.x, n3 = a23.y;
        const double n4 = a4;
        n1 = fma(-l3[3], u3[2], n1);
        n1 = fma(-l2[2], u2[1], n1);
        n1 = fma(-l1[1], u1[0], n1);
        n2 = fma(-l2[3], u2[1], n2);
        n2 = fma(-l1[2], u1[0], n2);
        n3 = fma(-l1[3], u1[0], n3);
        const double ri = fast_rcp(dk);
        const double m1 = n1 * ri, m2 = n2 * ri, m3 = n3 * ri, m4 = n4 * ri;
        if (lane == 0) {
            *reinterpret_cast<bd_d2 *>(&cf[k][0]) = bd_d2{m1, m2};
            *reinterpret_cast<bd_d2 *>(&cf[k][2]) = bd_d2{m3, m4};
            *reinterpret_cast<bd_d2 *>(&cf[k][4]) = bd_d2{ri, dk};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            l4[i] = l3[i]; l3[i] = l2[i]; l2[i] = l1[i];
            u4[i] = u3[i]; u3[i] = u2[i]; u2[i] = u1[i];
        }
        l1[0] = m1; l1[1] = m2; l1[2] = m3; l1[3] = m4;
        u1[0] = n1; u1[1] = n2; u1[2] = n3; u1[3] = n4;
    }
}

// cf[k][0..4] <- lam * band column k + sig on the diagonal; entries of rows beyond the block are zero.  All lanes of the wave.
__device__ __forceinline__ void band_shift_load(double (*cf)[BD_NC], const double *__restrict__ bd, int ld, int np, double lam, double sig,
                                                int lane) {
    for (int k = lane; k < np; k += 64) {
        double a[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] = (k + j < np) ? lam * bd[(long)j * ld + k] : 0.0;
        a[0] += sig;
        *reinterpret_cast<bd_d2 *>(&cf[k][0]) = bd_d2{a[0], a[1]};
        *reinterpret_cast<bd_d2 *>(&cf[k][2]) = bd_d2{a[2], a[3]};
        *reinterpret_cast<bd_d2 *>(&cf[k][4]) = bd_d2{a[4], 0.0};
    }
}

struct BandArgs {
    const double *W;             // (U^T Y Q) in the layout [x'][r][t~], rows of nt doubles
    double *B;                   // solve: the solutions, same layout (may be W)
    const double *es;            // spatial eigenvalues, fold order (nx)
    const double *bd[2];         // band of the scaled temporal blocks: bd[p][j * ld[p] + k]
    int ld[2];
    const double *amax[2];       // their scales m_p
    const double *sig;           // scalar noise variance (device)
    int nx, R, nt, np[2], c0[2];
    double *partials;            // ll: [0, nitems) quadratic forms, [nitems, 2 nitems) log-determinants
    double *coef;                // factor / solve: [item][BD_KMAX + BD_PAD][BD_NC]
    int npad;
};
constexpr int BD_PAD = 8;                   // zero columns in front of / behind a factor (the sweeps read four columns back / ahead)

// ------------------------------------------------------------------------------------------------
// log-likelihood: one wave per item, factor in the shadow of the first loads, forward sweep with the quadratic form
// ------------------------------------------------------------------------------------------------
constexpr int BL_CK = 16, BL_WAVES = 4;     // columns per chunk; waves (items) per workgroup
__global__ __launch_bounds__(64 * BL_WAVES) void ll_band_kernel(BandArgs g) {
    __shared__ double tile[BL_WAVES][2][64][BL_CK + 1];
    __shared__ __attribute__((aligned(16))) double coef[BL_WAVES][BD_PAD + BD_KMAX + BL_CK][BD_NC];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * BL_WAVES + wid, nitems = 2 * g.nx;
    if (item >= nitems) return;                                   // (whole waves leave: no workgroup barrier below)
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    if (np <= 0) {
        if (lane == 0) {
            g.partials[item] = 0.0;
            g.partials[nitems + item] = 0.0;
        }
        return;
    }
    const double lam_m = g.es[xr] * g.amax[p][0], sig = g.sig[0];
    const double *__restrict__ Wx = g.W + (long)xr * g.R * g.nt + g.c0[p];
    const int quarter = lane >> 4, kk_l = lane & 15;              // staging: four rows of 16 columns per load instruction
    double stg[16];
    auto load_chunk = [&](int r0, int nr, int k0) {               // all loads of a chunk are issued before any of them is used
        const int nk = min(BL_CK, np - k0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = quarter + 4 * i;
            stg[i] = (rr < nr && kk_l < nk) ? Wx[(long)(r0 + rr) * g.nt + k0 + kk_l] : 0.0;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[wid][buf][quarter + 4 * i][kk_l] = stg[i];
    };
    load_chunk(0, min(64, g.R), 0);                               // the first chunk's rows fly while the factor is formed
    double (*cf)[BD_NC] = coef[wid] + BD_PAD;
    for (int i = lane; i < BD_PAD * BD_NC; i += 64) (&coef[wid][0][0])[i] = 0.0;
    for (int i = lane; i < BL_CK * BD_NC; i += 64) (&cf[np][0])[i] = 0.0;                  // columns behind the block: l = 0, 1 / D = 0
    band_shift_load(cf, g.bd[p], g.ld[p], np, lam_m, sig, lane);
    __builtin_amdgcn_wave_barrier();
    band_factor(cf, np, lane);
    __builtin_amdgcn_wave_barrier();
    double lg = 0.0;
    for (int k = lane; k < np; k += 64) lg += log(cf[k][5]);
    const double logsum = wave_sum(lg);
    // ---- the sweeps: the item's rows in passes of 64, chunks of BL_CK columns
    double quad = 0.0;
    for (int r0 = 0; r0 < g.R; r0 += 64) {
        const int nr = min(64, g.R - r0);
        if (r0 > 0) load_chunk(r0, nr, 0);
        __builtin_amdgcn_wave_barrier();
        store_chunk(0);
        double z1 = 0.0, z2 = 0.0, z3 = 0.0, z4 = 0.0, q[2] = {0.0, 0.0};
        int buf = 0;
        for (int k0 = 0; k0 < np; k0 += BL_CK, buf ^= 1) {
            if (k0 + BL_CK < np) load_chunk(r0, nr, k0 + BL_CK);  // the next chunk's loads fly during this chunk's recurrence
            __builtin_amdgcn_wave_barrier();
            double wv[BL_CK];
            bd_d2 c01[BL_CK + 4], c23[BL_CK + 4];                  // multipliers of columns k0 - 4 .. k0 + BL_CK - 1
            double ri[BL_CK];
#pragma unroll
            for (int kk = 0; kk < BL_CK; ++kk) {
                wv[kk] = tile[wid][buf][lane][kk];
                ri[kk] = cf[k0 + kk][4];
            }
#pragma unroll
            for (int kk = 0; kk < BL_CK + 4; ++kk) {
                c01[kk] = *reinterpret_cast<const bd_d2 *>(&cf[k0 + kk - 4][0]);
                c23[kk] = *reinterpret_cast<const bd_d2 *>(&cf[k0 + kk - 4][2]);
            }
#pragma unroll
            for (int kk = 0; kk < BL_CK; ++kk) {                  // row k = k0 + kk: L[k][k-j] = l_{k-j}[j] = (column kk + 4 - j of the window)[j]
                double t = wv[kk];
                t = fma(-c23[kk].y, z4, t);                        // l_{k-4}[4]
                t = fma(-c23[kk + 1].x, z3, t);                    // l_{k-3}[3]
                t = fma(-c01[kk + 2].y, z2, t);                    // l_{k-2}[2]
                const double z = fma(-c01[kk + 3].x, z1, t);       // l_{k-1}[1]: the one dependent operation of the step
                q[kk & 1] = fma(z * z, ri[kk], q[kk & 1]);
                z4 = z3; z3 = z2; z2 = z1; z1 = z;
            }
            __builtin_amdgcn_wave_barrier();
            if (k0 + BL_CK < np) store_chunk(buf ^ 1);
        }
        quad += wave_sum(lane < nr ? q[0] + q[1] : 0.0);
    }
    if (lane == 0) {
        g.partials[item] = quad;
        g.partials[nitems + item] = logsum;
    }
}

// ------------------------------------------------------------------------------------------------
// prediction: factors of all items by one launch (a wave each), then the solve kernel reads them
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * BL_WAVES) void band_factor_kernel(BandArgs g) {
    __shared__ __attribute__((aligned(16))) double coef[BL_WAVES][BD_KMAX][BD_NC];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * BL_WAVES + wid, nitems = 2 * g.nx;
    if (item >= nitems) return;
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    double *__restrict__ out = g.coef + (long)item * (g.npad + 2 * BD_PAD) * BD_NC;
    // zero columns in front of and behind the block: multipliers 0, reciprocal pivot 0 (the sweeps pass through them unchanged)
    for (int i = lane; i < BD_PAD * BD_NC; i += 64) out[i] = 0.0;
    for (int i = BD_PAD * BD_NC + np * BD_NC + lane; i < (g.npad + 2 * BD_PAD) * BD_NC; i += 64) out[i] = 0.0;
    if (np <= 0) return;
    const double lam_m = g.es[xr] * g.amax[p][0], sig = g.sig[0];
    double (*cf)[BD_NC] = coef[wid];
    band_shift_load(cf, g.bd[p], g.ld[p], np, lam_m, sig, lane);
    __builtin_amdgcn_wave_barrier();
    band_factor(cf, np, lane);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < np * BD_NC; i += 64) out[BD_PAD * BD_NC + i] = (&cf[0][0])[i];
}

// The solve: the layout and the load / store phases of tridiag_solve_kernel (gram.hip) -- an item's rows come into LDS in one burst,
// four waves a batch of 64 columns each, the sweeps run in place on wave 0 -- with the factor read from band_factor_kernel's
// output instead of being formed here, and four multipliers per column.
constexpr int BS_BATCH = 64, BS_P = 64, BS_HB = 16;
__global__ __launch_bounds__(256) void band_solve_kernel(BandArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zbuf = smem;                               // [npad / 2][64][2]: w, then z, then x, in place
    double (*cfs)[BD_NC] = reinterpret_cast<double (*)[BD_NC]>(zbuf + (long)g.npad * BS_P);     // [npad + 2 BD_PAD][BD_NC]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ncf = (g.npad + 2 * BD_PAD) * BD_NC;
    for (int item = blockIdx.x; item < 2 * g.nx; item += gridDim.x) {
        const int xr = item >> 1, p = item & 1;
        const int np = g.np[p];
        if (np <= 0) continue;                             // (the whole workgroup: no barrier is skipped by part of it)
        const int nbatch = (np + BS_BATCH - 1) / BS_BATCH, npad = nbatch * BS_BATCH;
        const long rowbase = (long)xr * g.R * g.nt + g.c0[p];
        const int quarter = lane >> 4, kk_l = lane & 15;   // a global access: four rows of 16 columns
        const double *__restrict__ cin = g.coef + (long)item * ncf;
        for (int r0 = 0; r0 < g.R; r0 += BS_P) {
            const int nr = min(BS_P, g.R - r0);
            const double *const wl = g.W + rowbase + (long)(r0 + quarter) * g.nt + kk_l;
            const int b0 = wid * BS_BATCH;
            double stg[4][16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kq = b0 + 16 * q;
                const int kc = min(kk_l, max(np - 1 - kq, 0));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    double v = 0.0;
                    if (4 * i < nr && kq < np) {                         // wave-uniform: some row / column of the piece exists
                        const int rc = min(quarter + 4 * i, nr - 1) - quarter;
                        v = wl[(long)rc * g.nt + kq + (kc - kk_l)];
                    }
                    stg[q][i] = v;
                }
            }
            if (r0 == 0)
                for (int i = tid; i < ncf; i += 256) (&cfs[0][0])[i] = cin[i];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (b0 < npad) zbuf[((b0 + 16 * q + kk_l) >> 1) * (2 * BS_P) + (quarter + 4 * i) * 2 + (kk_l & 1)] = stg[q][i];
            __syncthreads();
            if (wid == 0) {
                bd_d2 *const zl = reinterpret_cast<bd_d2 *>(zbuf) + lane;     // this lane's row: columns 2j, 2j + 1 at zl[j * 64]
                double (*const cf)[BD_NC] = cfs + BD_PAD;                      // column k of the block
                // ---- forward: z_k = w_k - sum_j l_{k-j}[j] z_{k-j}   (padding columns: multipliers 0 -- z passes through)
                double z1 = 0.0, z2 = 0.0, z3 = 0.0, z4 = 0.0;
                for (int h0 = 0; h0 < npad; h0 += BS_HB) {
                    bd_d2 *const zb = zl + (h0 >> 1) * BS_P;
                    bd_d2 v[BS_HB / 2], c01[BS_HB + 4], c23[BS_HB + 4];
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) v[j] = zb[j * BS_P];
#pragma unroll
                    for (int kk = 0; kk < BS_HB + 4; ++kk) {
                        c01[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk - 4][0]);
                        c23[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk - 4][2]);
                    }
#pragma unroll
                    for (int kk = 0; kk < BS_HB; ++kk) {
                        double t = (kk & 1) ? v[kk >> 1].y : v[kk >> 1].x;
                        t = fma(-c23[kk].y, z4, t);
                        t = fma(-c23[kk + 1].x, z3, t);
                        t = fma(-c01[kk + 2].y, z2, t);
                        const double z = fma(-c01[kk + 3].x, z1, t);
                        if (kk & 1) v[kk >> 1].y = z; else v[kk >> 1].x = z;
                        z4 = z3; z3 = z2; z2 = z1; z1 = z;
                    }
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) zb[j * BS_P] = v[j];
                }
                // ---- backward: x_k = z_k / D_k - sum_j l_k[j] x_{k+j}
                double x1 = 0.0, x2 = 0.0, x3 = 0.0, x4 = 0.0;
                for (int h0 = npad - BS_HB; h0 >= 0; h0 -= BS_HB) {
                    bd_d2 *const zb = zl + (h0 >> 1) * BS_P;
                    bd_d2 v[BS_HB / 2], c01[BS_HB], c23[BS_HB];
                    double ri[BS_HB];
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) v[j] = zb[j * BS_P];
#pragma unroll
                    for (int kk = 0; kk < BS_HB; ++kk) {
                        c01[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][0]);
                        c23[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][2]);
                        ri[kk] = cf[h0 + kk][4];
                    }
#pragma unroll
                    for (int kk = BS_HB - 1; kk >= 0; --kk) {
                        double t = ((kk & 1) ? v[kk >> 1].y : v[kk >> 1].x) * ri[kk];
                        t = fma(-c23[kk].y, x4, t);
                        t = fma(-c23[kk].x, x3, t);
                        t = fma(-c01[kk].y, x2, t);
                        const double x = fma(-c01[kk].x, x1, t);
                        if (kk & 1) v[kk >> 1].y = x; else v[kk >> 1].x = x;
                        x4 = x3; x3 = x2; x2 = x1; x1 = x;
                    }
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) zb[j * BS_P] = v[j];
                }
            }
            __syncthreads();
            // ---- the solutions out: 16-column pieces of a row, read transposed; wave w its batches
            double *const bl = g.B + rowbase + (long)(r0 + quarter) * g.nt + kk_l;
            for (int bi = wid; bi < nbatch; bi += 4) {
                const int bb = bi * BS_BATCH;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = bb + 16 * q;
                    if (kq >= np) continue;                              // wave-uniform
                    const bool kfull = kq + 15 < np, kok = kq + kk_l < np;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (4 * i >= nr) continue;                       // wave-uniform
                        const double v = zbuf[((kq + kk_l) >> 1) * (2 * BS_P) + (quarter + 4 * i) * 2 + (kk_l & 1)];
                        if (kfull && 4 * i + 3 < nr) bl[(long)(4 * i) * g.nt + kq] = v;      // whole piece: no lane mask
                        else if (kok && quarter + 4 * i < nr) bl[(long)(4 * i) * g.nt + kq] = v;
                    }
                }
            }
            __syncthreads();                                             // the block is free for the next pass
        }
    }
}

static int band_npad(int npmax) { return (npmax + BS_BATCH - 1) / BS_BATCH * BS_BATCH; }
static size_t band_solve_lds(int npmax) {
    const int npad = band_npad(npmax);
    return ((size_t)npad * BS_P + (size_t)(npad + 2 * BD_PAD) * BD_NC) * sizeof(double);
}
// trials per pass (64) if column blocks of up to npmax fit the solve kernel's LDS block, else 0
int k_band_solve_pass(int npmax, int R) {
    return (R >= 16 && npmax <= BD_KMAX && band_solve_lds(npmax) <= (size_t)150 * 1024) ? BS_P : 0;
}

static BandArgs band_args(const double *W, double *B, const double *es, const double *const bd[2], const int ld[2],
                          const double *const amax[2], const double *sig, int nx, int R, int nt, const int np[2], const int c0[2]) {
    BandArgs g{};
    g.W = W; g.B = B; g.es = es; g.sig = sig; g.nx = nx; g.R = R; g.nt = nt;
    for (int p = 0; p < 2; ++p) {
        g.bd[p] = bd[p]; g.ld[p] = ld[p]; g.amax[p] = amax[p]; g.np[p] = np[p]; g.c0[p] = c0[p];
    }
    return g;
}

// gram.hip
void ll_tridiag_reduce_launch(gpcsd_ctx *c, const double *partials, int nitems, double *out_sumlog, double *out_quad, double *host_slot,
                              const double *status_src, int status_at, int status_doubles, hipStream_t s, bool *wrote);

bool k_ll_band(gpcsd_ctx *c, const double *W, const double *es, const double *const bd[2], const int ld[2], const double *const amax[2],
               const double *sig, int nx, int R, int nt, const int np[2], const int c0[2], double *out_sumlog, double *out_quad,
               hipStream_t s, double *host_slot, const double *status_src, int status_at, int status_doubles) {
    GP_REQUIRE(std::max(np[0], np[1]) <= BD_KMAX, -3, "ll_band: temporal blocks of %d columns", std::max(np[0], np[1]));
    BandArgs g = band_args(W, nullptr, es, bd, ld, amax, sig, nx, R, nt, np, c0);
    const int nitems = 2 * nx;
    g.partials = c->buf<double>("ll_tridiag_partials", (size_t)2 * nitems);
    ProfScope ps(c, "ll_band", 0.0, s);
    hipLaunchKernelGGL(ll_band_kernel, dim3(ceil_div(nitems, BL_WAVES)), dim3(64 * BL_WAVES), 0, s, g);
    bool wrote = false;
    ll_tridiag_reduce_launch(c, g.partials, nitems, out_sumlog, out_quad, host_slot, status_src, status_at, status_doubles, s, &wrote);
    GP_HIP(hipGetLastError());
    return wrote;
}

void k_band_solve(gpcsd_ctx *c, const double *W, double *B, const double *es, const double *const bd[2], const int ld[2],
                  const double *const amax[2], const double *sig, int nx, int R, int nt, const int np[2], const int c0[2], hipStream_t s) {
    const int npmax = std::max(np[0], np[1]);
    GP_REQUIRE(k_band_solve_pass(npmax, R) > 0, -3, "band_solve: temporal blocks of %d columns do not fit the solve kernel", npmax);
    BandArgs g = band_args(W, B, es, bd, ld, amax, sig, nx, R, nt, np, c0);
    g.npad = band_npad(npmax);
    const int nitems = 2 * nx;
    g.coef = c->buf<double>("band_coef", (size_t)nitems * (g.npad + 2 * BD_PAD) * BD_NC);
    const size_t lds = band_solve_lds(npmax);
    static size_t attr = 0;
    if (lds > attr) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(band_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    {
        ProfScope ps(c, "band_factor", 0.0, s);
        hipLaunchKernelGGL(band_factor_kernel, dim3(ceil_div(nitems, BL_WAVES)), dim3(64 * BL_WAVES), 0, s, g);
    }
    ProfScope ps(c, "band_solve", 6.0 * nx * (double)R * nt, s);
    static const int grid_cap = getenv("GPCSD_TS_GRID") ? atoi(getenv("GPCSD_TS_GRID")) : 192;     // (as k_tridiag_solve: leave CUs to the chains)
    const int grid = grid_cap > 0 ? std::min(nitems, grid_cap) : nitems;
    hipLaunchKernelGGL(band_solve_kernel, dim3(grid), dim3(256), lds, s, g);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
