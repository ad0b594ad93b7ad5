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
// with the older values are summed first).  Coefficients of a column k, ten doubles: L[k+1..k+4][k] (the backward sweep's), 1 / D_k,
// D_k, L[k][k-1..k-4] (the forward sweep's: the same multipliers by rows, so that a sweep reads five values per step).
#include <type_traits>

#include "devutil.hpp"
#include "kernels.hpp"

namespace gpcsd {

// (half-bandwidth 4 throughout: sytrd_bandtail.hpp's BT_W)
constexpr int BD_NC = 10;                   // doubles per column of a factor: column k's multipliers, 1 / D, D, row k's multipliers
constexpr int BD_KMAX = 256;                // columns of a block at most
typedef double bd_d2 __attribute__((ext_vector_type(2)));
// Read-only global memory through the CONSTANT address space: a load with a wave-uniform address is then a scalar load (s_load_*
// into SGPRs).  Plain `const double *__restrict__` leaves uniform loads as vector loads -- 64 lanes fetching one address each.
typedef const double __attribute__((address_space(4))) *bd_cptr;
__device__ __forceinline__ bd_cptr bd_const(const double *p) { return (bd_cptr)(unsigned long long)p; }

// In: cf[k][0..4] = the band of A (diagonal, four sub-diagonals) of column k, k < np.  Out: cf[k] = (l1, l2, l3, l4, 1 / D_k, D_k,
// L[k][k-1], L[k][k-2], L[k][k-3], L[k][k-4]).
// Called by every lane of one wave with the same arguments (cf in LDS, wave-private); lane 0 stores.
__device__ __forceinline__ void band_factor(double (*cf)[BD_NC], int np, int lane) {
    // History of the last four columns, column c in slot c & 3 (the loop is unrolled by four: every slot index below is a
    // compile-time constant -- shifting four columns of history through registers cost 48 moves per column step):
    //   hl[s][i-1] = L[c+i][c],  hu[s][i-1] = L[c+i][c] D_c    (sub-diagonal i = 1 .. 4)
    double hl[4][4], hu[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) hl[s][i] = hu[s][i] = 0.0;
    // the band of the next four columns is read while the current four are factored (an LDS round trip per column step, ~120
    // cycles on a chain of ~60, was most of the first version's 0.2 us per column)
    bd_d2 a01[4], a23[4];
    double a4[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + u, np - 1);
            a01[u] = *reinterpret_cast<const bd_d2 *>(&cf[k][0]);
            a23[u] = *reinterpret_cast<const bd_d2 *>(&cf[k][2]);
            a4[u] = cf[k][4];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < np; k0 += 4) {
        bd_d2 c01[4], c23[4];
        double c4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { c01[u] = a01[u]; c23[u] = a23[u]; c4[u] = a4[u]; }
        if (k0 + 4 < np) fetch(k0 + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u < np) {                                      // wave-uniform
                // column k = k0 + u in slot u; column k - j in slot (u - j) & 3.  Row k of L: L[k][k-j] = l_{k-j}[j];  row k+i:
                // L[k+i][k-j] = l_{k-j}[i+j].  The terms of the older columns first, the newest column (whose multipliers come out
                // of the previous step's reciprocal) last.
                constexpr int S1 = 3, S2 = 2, S3 = 1;               // (u + S) & 3 = slot of column k-1, k-2, k-3; k-4 is slot u itself
                const int s1 = (u + S1) & 3, s2 = (u + S2) & 3, s3 = (u + S3) & 3, s4 = u;
                double dk = c01[u].x;
                dk = fma(-hl[s4][3], hu[s4][3], dk);
                dk = fma(-hl[s3][2], hu[s3][2], dk);
                dk = fma(-hl[s2][1], hu[s2][1], dk);
                dk = fma(-hl[s1][0], hu[s1][0], dk);
                double n1 = c01[u].y, n2 = c23[u].x, n3 = c23[u].y;
                const double n4 = c4[u];
                n1 = fma(-hl[s3][3], hu[s3][2], n1);
                n1 = fma(-hl[s2][2], hu[s2][1], n1);
                n1 = fma(-hl[s1][1], hu[s1][0], n1);
                n2 = fma(-hl[s2][3], hu[s2][1], n2);
                n2 = fma(-hl[s1][2], hu[s1][0], n2);
                n3 = fma(-hl[s1][3], hu[s1][0], n3);
                const double ri = fast_rcp(dk);
                const double m1 = n1 * ri, m2 = n2 * ri, m3 = n3 * ri, m4 = n4 * ri;
                if (lane == 0) {
                    *reinterpret_cast<bd_d2 *>(&cf[k0 + u][0]) = bd_d2{m1, m2};
                    *reinterpret_cast<bd_d2 *>(&cf[k0 + u][2]) = bd_d2{m3, m4};
                    *reinterpret_cast<bd_d2 *>(&cf[k0 + u][4]) = bd_d2{ri, dk};
                    *reinterpret_cast<bd_d2 *>(&cf[k0 + u][6]) = bd_d2{hl[s1][0], hl[s2][1]};      // row k: L[k][k-j] = l_{k-j}[j]
                    *reinterpret_cast<bd_d2 *>(&cf[k0 + u][8]) = bd_d2{hl[s3][2], hl[s4][3]};
                }
                hl[u][0] = m1; hl[u][1] = m2; hl[u][2] = m3; hl[u][3] = m4;       // (overwrites column k - 4: read above)
                hu[u][0] = n1; hu[u][1] = n2; hu[u][2] = n3; hu[u][3] = n4;
            }
        }
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
    double *coef;                // factor / solve: [column][item][BD_NC]
    int npad;
};

// ------------------------------------------------------------------------------------------------
// factors of all items by one launch: ONE LANE PER ITEM
// ------------------------------------------------------------------------------------------------
// The recurrence is serial over the columns and tiny per column (ten FMAs, a reciprocal): with a whole wave walking ONE item
// redundantly (the first version) a launch took 60-70 us -- 62 instructions per column x 64 lanes of the same numbers.  Here a
// lane owns an item (its own shift lam = es[x'] m_p; the band itself is shared, in LDS): 768 items are twelve waves, one per CU.
// Coefficients go out through an LDS transpose, eight columns at a time, so that an item's factor is contiguous in global memory
// (the sweeps read it with scalar loads: uniform addresses).  The sweeps read a factor's coefficients with SCALAR loads (s_load
// into SGPRs, the FMAs take them as scalar operands): as broadcast reads from LDS -- 4.5 reads per column step and wave against
// 6 FMAs, four items per CU -- they were bound by the CU's one LDS pipe (75 us for the log-likelihood's kernel, 29 us for the
// tridiagonal form's).
__global__ __launch_bounds__(64) void band_factor_kernel(BandArgs g) {
    __shared__ __attribute__((aligned(16))) double sb[2][BD_KMAX + 4][6];      // the two blocks' bands, [p][k][j]; zero beyond a block
    const int lane = threadIdx.x, nitems = 2 * g.nx;
    const int item = blockIdx.x * 64 + lane;
    for (int p = 0; p < 2; ++p)
        for (int idx = lane; idx < (BD_KMAX + 4) * 6; idx += 64) {
            const int k = idx / 6, j = idx - 6 * k;
            sb[p][k][j] = (j < 5 && k + j < g.np[p]) ? g.bd[p][(long)j * g.ld[p] + k] : 0.0;
        }
    __syncthreads();
    if (item >= nitems) return;
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    const double lam = g.es[xr] * g.amax[p][0], sig = g.sig[0];
    // Layout of the output: [column k][item][BD_NC] -- the lanes of a wave (64 items) write 64 adjacent 80-byte records per column.
    // Item-major ([item][k][..]: every lane its own cache lines, 64 lines per store instruction) made this kernel 76 us.
    const long cs = (long)nitems * BD_NC;                          // doubles between consecutive columns of an item
    double *__restrict__ out = g.coef + (long)item * BD_NC;
    // history of the last four columns, column c in slot c & 3:  hl[s][i-1] = L[c+i][c],  hu[s][i-1] = L[c+i][c] D_c
    double hl[4][4], hu[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) hl[s][i] = hu[s][i] = 0.0;
    // sum log D_k as the logarithm of a running product, renormalised by its exponent every four columns (a logarithm per column
    // would be most of the loop)
    double prod = 1.0;
    int pexp = 0;
    // one column step; CHECK: the column may lie behind the block (multipliers 0, reciprocal pivot 0, history cleared)
    auto step = [&](const int k, auto uc, auto checkc) {
        constexpr int u = decltype(uc)::value;
        constexpr bool CHECK = decltype(checkc)::value;
        constexpr int s1 = (u + 3) & 3, s2 = (u + 2) & 3, s3 = (u + 1) & 3, s4 = u;
        const bool in = !CHECK || k < np;
        const bd_d2 b01 = *reinterpret_cast<const bd_d2 *>(&sb[p][k][0]), b23 = *reinterpret_cast<const bd_d2 *>(&sb[p][k][2]);
        const double b4 = sb[p][k][4];
        double dk = fma(lam, b01.x, sig);
        double n1 = lam * b01.y, n2 = lam * b23.x, n3 = lam * b23.y;
        double n4 = lam * b4;
        dk = fma(-hl[s4][3], hu[s4][3], dk);
        dk = fma(-hl[s3][2], hu[s3][2], dk);
        dk = fma(-hl[s2][1], hu[s2][1], dk);
        dk = fma(-hl[s1][0], hu[s1][0], dk);
        n1 = fma(-hl[s3][3], hu[s3][2], n1);
        n1 = fma(-hl[s2][2], hu[s2][1], n1);
        n1 = fma(-hl[s1][1], hu[s1][0], n1);
        n2 = fma(-hl[s2][3], hu[s2][1], n2);
        n2 = fma(-hl[s1][2], hu[s1][0], n2);
        n3 = fma(-hl[s1][3], hu[s1][0], n3);
        double ri = fast_rcp(dk);
        bd_d2 r01 = bd_d2{hl[s1][0], hl[s2][1]}, r23 = bd_d2{hl[s3][2], hl[s4][3]};        // row k: L[k][k-j] = l_{k-j}[j]
        if (CHECK && !in) {
            ri = 0.0; n1 = n2 = n3 = n4 = 0.0; dk = 1.0;
            r01 = bd_d2{0.0, 0.0}; r23 = bd_d2{0.0, 0.0};
        }
        const double m1 = n1 * ri, m2 = n2 * ri, m3 = n3 * ri, m4 = n4 * ri;
        double *__restrict__ o = out + (long)k * cs;
        *reinterpret_cast<bd_d2 *>(o + 0) = bd_d2{m1, m2};
        *reinterpret_cast<bd_d2 *>(o + 2) = bd_d2{m3, m4};
        *reinterpret_cast<bd_d2 *>(o + 4) = bd_d2{ri, dk};
        *reinterpret_cast<bd_d2 *>(o + 6) = r01;
        *reinterpret_cast<bd_d2 *>(o + 8) = r23;
        prod *= dk;
        hl[u][0] = m1; hl[u][1] = m2; hl[u][2] = m3; hl[u][3] = m4;
        hu[u][0] = n1; hu[u][1] = n2; hu[u][2] = n3; hu[u][3] = n4;
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>;
    const int nfull = np & ~3;                                      // (np differs between the parities: the trip count is per lane)
    int k0 = 0;
    for (; k0 < nfull; k0 += 4) {
        step(k0, T0{}, std::false_type{});
        step(k0 + 1, T1{}, std::false_type{});
        step(k0 + 2, T2{}, std::false_type{});
        step(k0 + 3, T3{}, std::false_type{});
        int ex;
        prod = frexp(prod, &ex);
        pexp += ex;
    }
    for (; k0 < g.npad; k0 += 4) {                                  // the last columns of the block and the zero columns behind it
        step(k0, T0{}, std::true_type{});
        step(k0 + 1, T1{}, std::true_type{});
        step(k0 + 2, T2{}, std::true_type{});
        step(k0 + 3, T3{}, std::true_type{});
        int ex;
        prod = frexp(prod, &ex);
        pexp += ex;
    }
    if (g.partials) g.partials[nitems + item] = np > 0 ? log(prod) + 0.6931471805599453 * (double)pexp : 0.0;
}

// ------------------------------------------------------------------------------------------------
// log-likelihood: one wave per item, forward sweep with the quadratic form
// ------------------------------------------------------------------------------------------------
constexpr int BL_CK = 16, BL_WAVES = 4;     // columns per chunk; waves (items) per workgroup
__global__ __launch_bounds__(64 * BL_WAVES) void ll_band_kernel(BandArgs g) {
    __shared__ double tile[BL_WAVES][2][64][BL_CK + 1];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * BL_WAVES + wid), nitems = 2 * g.nx;
    if (item >= nitems) return;                                   // (whole waves leave: no workgroup barrier below)
    const int xr = item >> 1, p = item & 1;
    const int np = g.np[p];
    if (np <= 0) {
        if (lane == 0) g.partials[item] = 0.0;
        return;
    }
    const double *__restrict__ Wx = g.W + (long)xr * g.R * g.nt + g.c0[p];
    // column k of the item's factor at cg + k * cs (layout [k][item][BD_NC]: band_factor_kernel)
    const double *__restrict__ cg = g.coef + (long)item * BD_NC;
    const long cs = (long)nitems * BD_NC;
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
    // ---- the sweeps: the item's rows in passes of 64, chunks of BL_CK columns
    double quad = 0.0;
    for (int r0 = 0; r0 < g.R; r0 += 64) {
        const int nr = min(64, g.R - r0);
        load_chunk(r0, nr, 0);
        __builtin_amdgcn_wave_barrier();
        store_chunk(0);
        double z1 = 0.0, z2 = 0.0, z3 = 0.0, z4 = 0.0, q[2] = {0.0, 0.0};
        int buf = 0;
        for (int k0 = 0; k0 < np; k0 += BL_CK, buf ^= 1) {
            if (k0 + BL_CK < np) load_chunk(r0, nr, k0 + BL_CK);  // the next chunk's loads fly during this chunk's recurrence
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int kb = 0; kb < BL_CK; kb += 4) {               // four rows at a time: 20 coefficients = 40 SGPRs
                const bd_cptr ck = bd_const(cg + (long)(k0 + kb) * cs);          // uniform addresses: scalar loads
                double wv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wv[kk] = tile[wid][buf][lane][kb + kk];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {                  // row k: z_k = w_k - sum_j L[k][k-j] z_{k-j}
                    double t = wv[kk];
                    t = fma(-ck[kk * cs + 9], z4, t);
                    t = fma(-ck[kk * cs + 8], z3, t);
                    t = fma(-ck[kk * cs + 7], z2, t);
                    const double z = fma(-ck[kk * cs + 6], z1, t);      // the one dependent operation of the step
                    q[kk & 1] = fma(z * z, ck[kk * cs + 4], q[kk & 1]);
                    z4 = z3; z3 = z2; z2 = z1; z1 = z;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (k0 + BL_CK < np) store_chunk(buf ^ 1);
        }
        quad += wave_sum(lane < nr ? q[0] + q[1] : 0.0);
    }
    if (lane == 0) g.partials[item] = quad;
}

// The solve: the layout and the load / store phases of tridiag_solve_kernel (gram.hip) -- an item's rows come into LDS in one burst,
// four waves a batch of 64 columns each, the sweeps run in place on wave 0 -- with the factor read from band_factor_kernel's
// output (scalar loads) instead of being formed here, and four multipliers per column.
constexpr int BS_BATCH = 64, BS_P = 64, BS_HB = 8;
__global__ __launch_bounds__(256) void band_solve_kernel(BandArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zbuf = smem;                               // [npad / 2][64][2]: w, then z, then x, in place
    double (*cf)[BD_NC] = reinterpret_cast<double (*)[BD_NC]>(zbuf + (long)g.npad * BS_P);     // [npad][BD_NC]: the item's factor
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nitems = 2 * g.nx;
    const long cs = (long)nitems * BD_NC;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int xr = item >> 1, p = item & 1;
        const int np = g.np[p];
        if (np <= 0) continue;                             // (the whole workgroup: no barrier is skipped by part of it)
        const int nbatch = (np + BS_BATCH - 1) / BS_BATCH, npad = nbatch * BS_BATCH;
        const long rowbase = (long)xr * g.R * g.nt + g.c0[p];
        const int quarter = lane >> 4, kk_l = lane & 15;   // a global access: four rows of 16 columns
        const double *__restrict__ cin = g.coef + (long)item * BD_NC;         // column k of the factor at cin + k * cs
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
            if (r0 == 0)                                                 // the factor: pairs of doubles, a column's ten contiguous
                for (int i = tid; i < npad * (BD_NC / 2); i += 256) {
                    const int k = i / (BD_NC / 2), j2 = i - k * (BD_NC / 2);
                    *reinterpret_cast<bd_d2 *>(&cf[k][2 * j2]) = *reinterpret_cast<const bd_d2 *>(cin + (long)k * cs + 2 * j2);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (b0 < npad) zbuf[((b0 + 16 * q + kk_l) >> 1) * (2 * BS_P) + (quarter + 4 * i) * 2 + (kk_l & 1)] = stg[q][i];
            __syncthreads();
            if (wid == 0) {
                bd_d2 *const zl = reinterpret_cast<bd_d2 *>(zbuf) + lane;     // this lane's row: columns 2j, 2j + 1 at zl[j * 64]
                // ---- forward: z_k = w_k - sum_j L[k][k-j] z_{k-j}   (padding columns: multipliers 0 -- z passes through)
                double z1 = 0.0, z2 = 0.0, z3 = 0.0, z4 = 0.0;
                for (int h0 = 0; h0 < npad; h0 += BS_HB) {
                    bd_d2 *const zb = zl + (h0 >> 1) * BS_P;
                    bd_d2 v[BS_HB / 2], c67[BS_HB], c89[BS_HB];
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) v[j] = zb[j * BS_P];
#pragma unroll
                    for (int kk = 0; kk < BS_HB; ++kk) {
                        c67[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][6]);
                        c89[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][8]);
                    }
#pragma unroll
                    for (int kk = 0; kk < BS_HB; ++kk) {
                        double t = (kk & 1) ? v[kk >> 1].y : v[kk >> 1].x;
                        t = fma(-c89[kk].y, z4, t);
                        t = fma(-c89[kk].x, z3, t);
                        t = fma(-c67[kk].y, z2, t);
                        const double z = fma(-c67[kk].x, z1, t);
                        if (kk & 1) v[kk >> 1].y = z; else v[kk >> 1].x = z;
                        z4 = z3; z3 = z2; z2 = z1; z1 = z;
                    }
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) zb[j * BS_P] = v[j];
                }
                // ---- backward: x_k = z_k / D_k - sum_j L[k+j][k] x_{k+j}
                double x1 = 0.0, x2 = 0.0, x3 = 0.0, x4 = 0.0;
                for (int h0 = npad - BS_HB; h0 >= 0; h0 -= BS_HB) {
                    bd_d2 *const zb = zl + (h0 >> 1) * BS_P;
                    bd_d2 v[BS_HB / 2], c01[BS_HB], c23[BS_HB], c45[BS_HB];
#pragma unroll
                    for (int j = 0; j < BS_HB / 2; ++j) v[j] = zb[j * BS_P];
#pragma unroll
                    for (int kk = 0; kk < BS_HB; ++kk) {
                        c01[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][0]);
                        c23[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][2]);
                        c45[kk] = *reinterpret_cast<const bd_d2 *>(&cf[h0 + kk][4]);
                    }
#pragma unroll
                    for (int kk = BS_HB - 1; kk >= 0; --kk) {
                        double t = ((kk & 1) ? v[kk >> 1].y : v[kk >> 1].x) * c45[kk].x;
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
    return ((size_t)npad * BS_P + (size_t)npad * BD_NC) * sizeof(double);
}
// trials per pass (64) if column blocks of up to npmax fit the solve kernel's LDS block, else 0
int k_band_solve_pass(int npmax, int R) {
    return (R >= 16 && npmax <= BD_KMAX && band_solve_lds(npmax) <= (size_t)156 * 1024) ? BS_P : 0;
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

// The log-likelihood's factors (band_factor_kernel with the log-determinants) on stream `sf`: the caller orders `sf` behind the
// spatial spectrum and the band, and the sweep (k_ll_band, on its own stream) behind this launch.
void k_ll_band_factor(gpcsd_ctx *c, const double *es, const double *const bd[2], const int ld[2], const double *const amax[2],
                      const double *sig, int nx, const int np[2], hipStream_t sf) {
    const int npmax = std::max(np[0], np[1]);
    GP_REQUIRE(npmax <= BD_KMAX, -3, "ll_band: temporal blocks of %d columns", npmax);
    const int c0[2] = {0, 0};
    BandArgs g = band_args(nullptr, nullptr, es, bd, ld, amax, sig, nx, 0, 0, np, c0);
    const int nitems = 2 * nx;
    g.npad = band_npad(npmax);
    g.partials = c->buf<double>("ll_tridiag_partials", (size_t)2 * nitems);
    g.coef = c->buf<double>("band_coef_ll", (size_t)nitems * g.npad * BD_NC);
    ProfScope ps(c, "band_factor_ll", 0.0, sf);
    hipLaunchKernelGGL(band_factor_kernel, dim3(ceil_div(nitems, 64)), dim3(64), 0, sf, g);
    GP_HIP(hipGetLastError());
}

bool k_ll_band(gpcsd_ctx *c, const double *W, const double *es, const double *const bd[2], const int ld[2], const double *const amax[2],
               const double *sig, int nx, int R, int nt, const int np[2], const int c0[2], double *out_sumlog, double *out_quad,
               hipStream_t s, double *host_slot, const double *status_src, int status_at, int status_doubles) {
    const int npmax = std::max(np[0], np[1]);
    GP_REQUIRE(npmax <= BD_KMAX, -3, "ll_band: temporal blocks of %d columns", npmax);
    BandArgs g = band_args(W, nullptr, es, bd, ld, amax, sig, nx, R, nt, np, c0);
    const int nitems = 2 * nx;
    g.npad = band_npad(npmax);
    g.partials = c->buf<double>("ll_tridiag_partials", (size_t)2 * nitems);
    g.coef = c->buf<double>("band_coef_ll", (size_t)nitems * g.npad * BD_NC);
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
    g.coef = c->buf<double>("band_coef", (size_t)nitems * g.npad * BD_NC);
    g.partials = nullptr;
    const size_t lds = band_solve_lds(npmax);
    static size_t attr = 0;
    if (lds > attr) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(band_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    {
        ProfScope ps(c, "band_factor", 0.0, s);
        hipLaunchKernelGGL(band_factor_kernel, dim3(ceil_div(nitems, 64)), dim3(64), 0, s, g);
    }
    ProfScope ps(c, "band_solve", 6.0 * nx * (double)R * nt, s);
    static const int grid_cap = getenv("GPCSD_TS_GRID") ? atoi(getenv("GPCSD_TS_GRID")) : 192;     // (as k_tridiag_solve: leave CUs to the chains)
    const int grid = grid_cap > 0 ? std::min(nitems, grid_cap) : nitems;
    hipLaunchKernelGGL(band_solve_kernel, dim3(grid), dim3(256), lds, s, g);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
