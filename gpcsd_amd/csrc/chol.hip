// Blocked right-looking Cholesky: numpy.linalg.cholesky at gpcsd1d.py:303-304 and
// gpcsd2d.py:343-350 (sample_prior), plus the dense-K cross-check path (potrf + log-det + trsm) at the N ~ 12 000 of a
// 24 x 500 problem.
//
// Two block levels.  Outer blocks of NBO = 256 columns carry the flops: per outer step
//   D  the 256 x 256 diagonal block is factored AND inverted (X = L11^-1): diag128_kernel on its two NB = 128 sub-blocks -- one
//      workgroup of 512 threads each: 16-column panel steps by one wave (v_readlane multipliers), the rank-16 update of the rest of
//      the sub-block on MFMA by all eight waves, then the inverse of the triangular factor by doubling, 16 -> 32 -> 64 -> 128 with
//      MFMA products -- a 128 x 128 EPI_SUB product between them, and the block inverse assembled from the two sub-block inverses
//      (X_lowerleft = -X_22 (L21 X_11): two small products);
//   P  the panel solve L21 = A21 X^T is ONE MFMA GEMM with K = 256 (no triangular solve kernel: trsm by inverted diagonal blocks
//      is a product);
//   T  the trailing update A22 -= L21 L21^T is a rank-256 MFMA GEMM over the tiles on or below the diagonal only, accumulators
//      preloaded with -C (EPI_SUB: no second pass over A22); rank 256 is 32 flop per byte of A22, above the fp64 MFMA ridge of ~10.
// Look-ahead with gates: the next step's D runs on a side stream beside T, which is split by rows into T_b1 / T_b2 behind one-wave
// gate launches that hold each part back until the diagonal workgroup it has to leave a CU to is resident (under the update's
// thousands of tiles no CU ever drains by itself).  n = 12 000: 18 ms = 0.40 of the fp64 MFMA peak.
// Triangular solves with many right-hand sides use the same inverted-diagonal-block + GEMM scheme.
#include "kernels.hpp"

namespace gpcsd {

constexpr int NB = 128;       // sub-block: one workgroup factors and inverts it
constexpr int NBO = 256;      // outer block: rank of the trailing updates
// leading dimensions of the panel buffers: NOT a power of two -- rows 2 KB apart fall on the same few memory channels, and
// the rank-256 update reads its operand panel tile by tile at that stride (31 TF/s with ld = 256)
constexpr int LDW = NBO + 16, LDWS = NB + 16;

// sqrt(d) and 1 / sqrt(d) together from v_rsq_f64 by two coupled Newton (Goldschmidt) steps and one correction of the root: ten
// dependent operations instead of an IEEE square root followed by an IEEE division (~35), on the serial path of every column
// step.  Within 1-2 ulp; d <= 0 or non-finite gives NaN / inf as the plain sqrt would (the caller flags d <= 0).
__device__ __forceinline__ void sqrt_rsqrt(double d, double &root, double &rinv) {
    const double y = __builtin_amdgcn_rsq(d);
    double gq = d * y, h = 0.5 * y;
    double r = fma(-gq, h, 0.5);
    gq = fma(gq, r, gq);
    h = fma(h, r, h);
    r = fma(-gq, h, 0.5);
    gq = fma(gq, r, gq);
    h = fma(h, r, h);
    const double e = fma(-gq, gq, d);
    root = fma(e, h, gq);
    rinv = h + h;
}

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int lane) {       // lane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------------------------------
// One workgroup (16 waves) factors AND inverts a 128 x 128 diagonal block; the block lives in LDS.
//
// Factor: eight panels of 16 columns.  A panel is factored by ONE wave with no barrier and no LDS traffic inside it: lane l holds
// the panel's 16 entries of rows k0 + l and k0 + 64 + l in registers, the pivot and the 15 multipliers of a column step reach the
// other lanes by v_readlane (the diagonal 16 x 16 block is lanes 0..15), root and reciprocal root come from v_rsq_f64 + two
// Goldschmidt steps.  The rank-16 update of the rest of the block is fp64 MFMA over its 16 x 16 fragments on or below the diagonal,
// dealt to the 16 waves -- a column step costs the serial wave ~300 cycles and the block two barriers per panel (a rank-1 sweep
// with every thread reading its rows' and columns' multipliers from LDS is LDS-pipe bound at ~2000 cycles per column step: that
// version measured 326 us per block).
// Inverse: X = L^-1 -- the eight diagonal 16 x 16 blocks by forward substitution in registers (one wave each, side by side), then
// three doubling levels X_BA = -X_C (B X_A) (h = 16, 32, 64) as MFMA products.  X is kept transposed in the strictly upper part of
// the block's LDS storage (X[i][j] at [j][i + 1]; L occupies [i][j], j <= i), the intermediate B X_A in the slot X_BA will occupy.
// Fragment maps (cdna_hip_programming.md section 3): A lane l holds A[l & 15][l >> 4], B lane l holds B[l >> 4][l & 15], C lane l
// register r holds C[(l >> 4) + 4 r][l & 15].
constexpr int DN = 128, DLD = 130, LPLD = 17;
constexpr int DNT = 512, DNW = DNT / 64;      // threads / waves of the workgroup: 8 waves keep 256 VGPRs each (with 16 waves and 128
                                              // registers the panel arrays and the inverse's operand strips spilled to scratch)
constexpr size_t DIAG_LDS_BYTES = ((size_t)DN * DLD + (size_t)DN * LPLD + DN) * sizeof(double);

template <int H>
__device__ __forceinline__ void inverse_level(double *As, int wid, int lane) {
    constexpr int NF = H / 16, NPAIR = 64 / H;
    const int fr = lane & 15, fq = lane >> 4;
    // product 1: T = B X_A, fragment (fi, fj) of pair q; count = NPAIR * NF * NF = H / 4
    for (int t = wid; t < NPAIR * NF * NF; t += DNW) {
        const int q = t / (NF * NF), rem = t % (NF * NF), fi = rem / NF, fj = rem % NF;
        const int o = 2 * H * q;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int kf = fj; kf < NF; ++kf) {                       // X_A is lower triangular: rows k >= columns j
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int kl = 16 * kf + 4 * kk + fq;            // k inside the pair's first block
                const double a = As[(o + H + 16 * fi + fr) * DLD + (o + kl)];                 // B[i][k] = L[o + H + i][o + k]
                const int jl = 16 * fj + fr;
                double b = As[(o + jl) * DLD + (o + kl) + 1];                                  // X_A[k][j], stored at [j][k + 1]
                if (kl < jl) b = 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)                               // T[i][j] -> the slot of X[o + H + i][o + j]: [o + j][o + H + i + 1]
            As[(o + 16 * fj + fr) * DLD + (o + H + 16 * fi + fq + 4 * r) + 1] = acc[r];
    }
    __syncthreads();
    // product 2: X_BA = -X_C T, one wave per column strip fj of a pair (it reads the whole strip of T before it writes any of it)
    if (wid < NPAIR * NF) {
        const int q = wid / NF, fj = wid % NF;
        const int o = 2 * H * q;
        double tb[NF][4];
#pragma unroll
        for (int kf = 0; kf < NF; ++kf)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                tb[kf][kk] = As[(o + 16 * fj + fr) * DLD + (o + H + 16 * kf + 4 * kk + fq) + 1];   // T[k][j], B-operand layout
        d4 out[NF];
#pragma unroll
        for (int fi = 0; fi < NF; ++fi) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kf = 0; kf < NF; ++kf) {
                if (kf <= fi) {                                   // X_C is lower triangular: columns k <= rows i
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int il = 16 * fi + fr, kl = 16 * kf + 4 * kk + fq;
                        double a = -As[(o + H + kl) * DLD + (o + H + il) + 1];                 // X_C[i][k], stored at [k][i + 1]
                        if (kl > il) a = 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tb[kf][kk], acc, 0, 0, 0);
                    }
                }
            }
            out[fi] = acc;
        }
#pragma unroll
        for (int fi = 0; fi < NF; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                As[(o + 16 * fj + fr) * DLD + (o + H + 16 * fi + fq + 4 * r) + 1] = out[fi][r];
    }
    __syncthreads();
}

// FACTOR: Cholesky of the nb x nb (<= 128) block at Ablk (in: its lower part; out: L in place, upper part zeroed), then inv(L) to
// Xout (lower; leading dimension ldx).  !FACTOR: the block already is a Cholesky factor: inverse only.
// status: first failing pivot + 1 (global index pivot_base + j).  Blocks smaller than 128 are padded with the identity.
// started (device word, agent-scope store: visible to the other XCDs at once; may be null): stamped with `token` as soon as the workgroup runs -- it then HAS its CU; potrf's gate
// launches poll the word before they let the trailing update's tiles take every other one (see potrf_device).
template <bool FACTOR>
__global__ __launch_bounds__(DNT) void diag128_kernel(double *Ablk, long lda, int nb, double *Xout, long ldx, int *status,
                                                       int pivot_base, unsigned long long *clk, unsigned int *started,
                                                       unsigned int token) {
    __builtin_amdgcn_s_setprio(3);
    if (started && threadIdx.x == 0) __hip_atomic_store(started, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // clk (measurement aid, normally null): wall-clock stamps (100 MHz) at the phase boundaries -- [0] start, [1] block loaded,
    // [2] / [3] ticks spent in the serial panels / the rank-16 updates, [4] L stored, [5] diagonal inverses, [6..8] doubling levels,
    // [9] X stored
    auto stamp = [&](int k) { if (clk && threadIdx.x == 0) clk[k] = wall_clock64(); };
    unsigned long long t_f1 = 0, t_f2 = 0, t_last = 0;
    stamp(0);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *As = smem;                        // [DN][DLD]
    double *Lp = As + DN * DLD;               // [DN][LPLD]: the current panel of L
    double *dinv = Lp + DN * LPLD;            // [DN]: reciprocals of L's diagonal
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    {   // all sixteen loads of a thread in flight before the first LDS store (interleaved, each store waited for its load: 10 us)
        constexpr int NL = DN * DN / DNT;
        double v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + DNT * k, i = e >> 7, c = e & 127;
            v[k] = (i < nb && c < nb) ? (c <= i ? Ablk[(long)i * lda + c] : 0.0) : (i == c ? 1.0 : 0.0);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + DNT * k;
            As[(e >> 7) * DLD + (e & 127)] = v[k];
        }
    }
    __syncthreads();
    stamp(1);
    if (clk) t_last = wall_clock64();
    if (FACTOR) {
#pragma unroll 1
        for (int pb = 0; pb < 8; ++pb) {
            const int k0 = 16 * pb;
            if (wid == 0) {
                const int row0 = k0 + lane, row1 = k0 + 64 + lane;          // row0 < 128 always (k0 <= 112, lane < 16 ... 63: checked)
                double p0[16], p1[16];
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    p0[cc] = row0 < DN ? As[row0 * DLD + k0 + cc] : 0.0;
                    p1[cc] = row1 < DN ? As[row1 * DLD + k0 + cc] : 0.0;
                }
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const double piv = readlane_d(p0[jj], jj);
                    if (lane == 0 && !(piv > 0.0) && k0 + jj < nb) atomicCAS(status, 0, pivot_base + k0 + jj + 1);
                    double root, rinv;
                    sqrt_rsqrt(piv, root, rinv);
                    if (lane == 0) dinv[k0 + jj] = rinv;
                    p0[jj] = lane == jj ? root : p0[jj] * rinv;
                    p1[jj] *= rinv;
#pragma unroll
                    for (int cc = jj + 1; cc < 16; ++cc) {
                        // (the multipliers through LDS instead -- written by the diagonal block's lanes, read back as broadcasts -- or
                        // through ds_bpermute: no faster; a column step is its ~11 levels of dependent fp64 operations, ~25 cycles each)
                        const double lc = readlane_d(p0[jj], cc);
                        p0[cc] = fma(-p0[jj], lc, p0[cc]);
                        p1[cc] = fma(-p1[jj], lc, p1[cc]);
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    if (row0 < DN) {
                        const double v = cc <= lane ? p0[cc] : 0.0;        // (above the diagonal of the 16 x 16 block: zero)
                        As[row0 * DLD + k0 + cc] = v;
                        Lp[row0 * LPLD + cc] = v;
                    }
                    if (row1 < DN) {
                        As[row1 * DLD + k0 + cc] = p1[cc];
                        Lp[row1 * LPLD + cc] = p1[cc];
                    }
                }
            }
            __syncthreads();
            if (clk) { const unsigned long long t = wall_clock64(); t_f1 += t - t_last; t_last = t; }
            // rank-16 update of the fragments (fi, fc), pb < fc <= fi <= 7
            const int m = 7 - pb, count = m * (m + 1) / 2;
            for (int t = wid; t < count; t += DNW) {
                int ri = 0;
                while ((ri + 1) * (ri + 2) / 2 <= t) ++ri;
                const int rc = t - ri * (ri + 1) / 2;
                const int fi = pb + 1 + ri, fc = pb + 1 + rc;
                d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = As[(16 * fi + fq + 4 * r) * DLD + 16 * fc + fr];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double a = -Lp[(16 * fi + fr) * LPLD + 4 * kk + fq];
                    const double b = Lp[(16 * fc + fr) * LPLD + 4 * kk + fq];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) As[(16 * fi + fq + 4 * r) * DLD + 16 * fc + fr] = acc[r];
            }
            __syncthreads();
            if (clk) { const unsigned long long t = wall_clock64(); t_f2 += t - t_last; t_last = t; }
        }
        if (clk && tid == 0) { clk[2] = t_f1; clk[3] = t_f2; }
        for (int e = tid; e < DN * DN; e += DNT) {
            const int i = e >> 7, c = e & 127;
            if (i < nb && c < nb) Ablk[(long)i * lda + c] = c <= i ? As[i * DLD + c] : 0.0;           // L in place, upper part zeroed
        }
        stamp(4);
    } else {
        if (tid < DN) dinv[tid] = 1.0 / As[tid * DLD + tid];
        __syncthreads();
    }
    // ---- inverse: diagonal 16 x 16 blocks (one wave each), then the doubling levels
    if (wid < 8) {
        const int b0 = 16 * wid, j = fr;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) sacc = fma(-As[(b0 + i) * DLD + b0 + k], x[k], sacc);
            x[i] = sacc * dinv[b0 + i];
            __builtin_amdgcn_sched_barrier(0);             // (keeps the 120 LDS reads of the block from being hoisted into 240 registers)
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i >= j) As[(b0 + j) * DLD + (b0 + i) + 1] = x[i];                                   // X[i][j] at [j][i + 1]
        }
    }
    __syncthreads();
    stamp(5);
    inverse_level<16>(As, wid, lane);
    stamp(6);
    inverse_level<32>(As, wid, lane);
    stamp(7);
    inverse_level<64>(As, wid, lane);
    stamp(8);
    for (int e = tid; e < DN * DN; e += DNT) {
        const int i = e >> 7, c = e & 127;
        if (i < nb && c < nb) Xout[(long)i * ldx + c] = c <= i ? As[c * DLD + i + 1] : 0.0;
    }
    stamp(9);
}

static void launch_diag(double *Ablk, long lda, int nb, double *Xout, long ldx, int *status, int pivot_base, bool factor,
                        hipStream_t s, unsigned long long *clk = nullptr, unsigned int *started = nullptr, unsigned int token = 0) {
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diag128_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)DIAG_LDS_BYTES));
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diag128_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)DIAG_LDS_BYTES));
    }
    if (factor)
        hipLaunchKernelGGL(diag128_kernel<true>, dim3(1), dim3(DNT), DIAG_LDS_BYTES, s, Ablk, lda, nb, Xout, ldx, status, pivot_base, clk,
                           started, token);
    else
        hipLaunchKernelGGL(diag128_kernel<false>, dim3(1), dim3(DNT), DIAG_LDS_BYTES, s, Ablk, lda, nb, Xout, ldx, status, pivot_base, clk,
                           started, token);
}

__global__ void copy2d_kernel(const double *__restrict__ src, long lds_, double *__restrict__ dst, long ldd, int rows, int cols) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < (long)rows * cols) {
        const int i = (int)(e / cols), j = (int)(e % cols);
        dst[(long)i * ldd + j] = src[(long)i * lds_ + j];
    }
}

__global__ void zero_upper_kernel(double *A, int n) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < (long)n * n) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (j > i) A[e] = 0.0;
    }
}

// Tile configuration of the diagonal chain's small products.  On their own the 32 x 32 tiles with deep K slabs are fastest; beside
// the trailing update a workgroup of that configuration (64 KB of LDS, more registers than an update tile) only finds room on a CU
// when TWO update tiles have retired there -- the products then took 58 us on average, up to 500.  The update's own 64 x 64
// configuration fits wherever one of its tiles retires.
static thread_local bool g_chain_under_update = false;      // set by potrf_device around a chain that runs beside a long update
static int small_cfg() { return g_chain_under_update ? 2 : 0; }

static void small_gemm(gpcsd_ctx *c, int M, int N, int K, const double *A, long lda, const double *B, long ldb, bool tb, double *C,
                       long ldc, double alpha, int epi, const char *name, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    GemmDesc g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.transB = tb; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.epi = epi;
    g.prio = 1;
    g.cfg = small_cfg();
    g.prof_name = name;
    gemm_f64(c, g, s);
}

// X (nb x nb, leading dimension NBO, lower) = inverse of the lower-triangular block L at Lblk (leading dimension ld) whose
// 64 x 64 diagonal sub-blocks' inverses already sit on X's diagonal: adjacent pairs of inverted blocks of size h combine as
//   inv [[A, 0], [B, C]] = [[X_A, 0], [-X_C (B X_A), X_C]],   h = 64, 128, ...
static void assemble_block_inverse(gpcsd_ctx *c, const double *Lblk, long ld, int nb, double *X, double *tmp, hipStream_t s,
                                   const double *B0 = nullptr, long ldb0 = 0) {   // B0: the first pair's B block, if it also sits elsewhere
    for (int h = NB; h < nb; h *= 2) {
        for (int o = 0; o + h < nb; o += 2 * h) {
            const int h2 = std::min(h, nb - (o + h));                 // rows of the lower block (the last one may be ragged)
            // tmp (h2 x h) = B X_A,  B = L[o + h : o + h + h2, o : o + h]
            const bool alt = B0 && h == NB && o == 0;
            small_gemm(c, h2, h, h, alt ? B0 : Lblk + (long)(o + h) * ld + o, alt ? ldb0 : ld, X + (long)o * NBO + o, NBO, false, tmp, NBO,
                       1.0, EPI_STORE, "potrf_inv", s);
            // X[o + h.., o..] = -X_C tmp
            small_gemm(c, h2, h, h2, X + (long)(o + h) * NBO + (o + h), NBO, tmp, NBO, false, X + (long)(o + h) * NBO + o, NBO, -1.0,
                       EPI_STORE, "potrf_inv", s);
        }
    }
}

// D of the header: factor the nb x nb (<= NBO) diagonal block at Ablk in place and leave its inverse in X (NBO x NBO buffer).
// Dependent launches at nb = 256: factor + invert the upper 128 block, panel product, rank-128 update, factor + invert the lower
// block, copy, two products for the lower-left block of the inverse.
// One wave that waits (bounded) until *flag has reached token: placed in front of a machine-filling launch, it holds that launch
// back until a workgroup of another stream that needs a whole CU is resident.  Steers the order of execution only: every data
// dependence is an event; when the time is up the stream simply goes on.
// The word lives in device memory (flag[0]; flag[1] counts the gates whose time ran out -- gpcsd_potrf_bench reports them, so a
// regression shows as a count and not as unexplained milliseconds); loads and stores are agent-scope atomics.
__global__ __launch_bounds__(64) void gate_kernel(unsigned int *flag, unsigned int token, unsigned int max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - token) < 0) {
        if (wall_clock64() - t0 >= max_ticks) {
            atomicAdd(flag + 1, 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// first_token != 0: the diagonal launches stamp c->h_chol_flag (a device word) with first_token, first_token + 1, ..
static void factor_diag_block(gpcsd_ctx *c, double *Ablk, long ld, int nb, double *X, double *tmp, double *Wsub, int *d_status,
                              int pivot_base, hipStream_t s, unsigned int first_token = 0) {
    ProfScope ps(c, "potrf_diag_block", (double)nb * nb * nb * (1.0 / 3.0 + 1.0 / 3.0), s);
    for (int p0 = 0; p0 < nb; p0 += NB) {
        const int b = std::min(NB, nb - p0), rows = nb - (p0 + b);
        double *App = Ablk + (long)p0 * ld + p0;
        launch_diag(App, ld, b, X + (long)p0 * NBO + p0, (long)NBO, d_status, pivot_base + p0, true, s, nullptr,
                    first_token ? c->h_chol_flag : nullptr, first_token + (unsigned)(p0 / NB));
        if (rows > 0) {
            double *A21 = Ablk + (long)(p0 + b) * ld + p0;
            // L21 = A21 X_pp^T (into Wsub; copied into place below), A22 -= L21 L21^T -- all inside the diagonal block
            small_gemm(c, rows, b, b, A21, ld, X + (long)p0 * NBO + p0, NBO, true, Wsub, LDWS, 1.0, EPI_STORE, "potrf_blk_panel", s);
            GemmDesc g;
            g.M = rows; g.N = rows; g.K = b;
            g.A = Wsub; g.lda = LDWS; g.B = Wsub; g.ldb = LDWS; g.transB = true;
            g.C = Ablk + (long)(p0 + b) * ld + (p0 + b); g.ldc = ld;
            g.epi = EPI_SUB; g.lower = true; g.prio = 1; g.cfg = small_cfg();
            g.prof_name = "potrf_blk_syrk";
            gemm_f64(c, g, s);
            if (nb > 2 * NB)             // (more than two sub-blocks: the next sub-step reuses Wsub -- L21 into its place now)
                hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)rows * b, 256)), dim3(256), 0, s, (const double *)Wsub, (long)LDWS,
                                   A21, ld, rows, b);
        }
    }
    if (nb > NB && nb <= 2 * NB) {
        // two sub-blocks (the usual 256): the inverse's lower-left block reads L21 from the panel buffer, and the copy of L21
        // into its place comes last -- one dependent launch less in front of the panel solve that waits for this chain
        assemble_block_inverse(c, Ablk, ld, nb, X, tmp, s, Wsub, LDWS);
        hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)(nb - NB) * NB, 256)), dim3(256), 0, s, (const double *)Wsub, (long)LDWS,
                           Ablk + (long)NB * ld, ld, nb - NB, NB);
    } else {
        assemble_block_inverse(c, Ablk, ld, nb, X, tmp, s);
    }
    GP_HIP(hipGetLastError());
}

static int potrf_cfg_env(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

void potrf_device(gpcsd_ctx *c, double *A, int n, int *d_status, hipStream_t s_in) {
    ProfScope ps(c, "potrf", (double)n * n * n / 3.0, s_in);
    double *X = c->buf<double>("chol_Linv", (size_t)NBO * NBO);
    double *tmp = c->buf<double>("chol_inv_tmp", (size_t)NBO * NBO);
    double *Wsub = c->buf<double>("chol_blk_panel", (size_t)NBO * LDWS);
    double *W = c->buf<double>("chol_panel", (size_t)n * LDW);
    // look-ahead: the next diagonal block on a side stream (a chain stream of the context: high priority -- its launches are
    // tiny and must not queue behind the update's tiles), beside the bulk of the trailing update
    static const bool lookahead = potrf_cfg_env("GPCSD_POTRF_LOOKAHEAD", 1) != 0;
    static const int t_cfg = potrf_cfg_env("GPCSD_POTRF_TCFG", 0);      // tile configuration of the trailing update (0: automatic)
    const bool la = lookahead && n > 2 * NBO;
    static const bool gate_on = potrf_cfg_env("GPCSD_POTRF_GATES", 1) != 0;
    // (tried: two streams with disjoint CU masks, 8 CUs for the side chain -- every GEMM on the CU-restricted main stream ran at
    // a fraction of its rate: 30.3 against 23.4 ms at n = 12 000)
    hipStream_t s = s_in;
    hipStream_t sd = (s_in == c->stream3) ? c->stream2 : c->stream3;
    if (la) {                                                          // the side stream starts behind whatever produced A
        GP_HIP(hipEventRecord(c->ev_chol_a, s_in));
        GP_HIP(hipStreamWaitEvent(sd, c->ev_chol_a, 0));
    }
    // the panel solve multiplies by the whole NBO x NBO inverse: the blocks above its diagonal are never written and must be zero
    GP_HIP(hipMemsetAsync(X, 0, (size_t)NBO * NBO * sizeof(double), la ? sd : s));
    // first diagonal block
    factor_diag_block(c, A, n, std::min(NBO, n), X, tmp, Wsub, d_status, 0, la ? sd : s);
    if (la) {
        GP_HIP(hipEventRecord(c->ev_chol_d, sd));
        GP_HIP(hipStreamWaitEvent(s, c->ev_chol_d, 0));
    }
    for (int k0 = 0; k0 < n; k0 += NBO) {
        const int nb = std::min(NBO, n - k0), m = n - (k0 + nb);
        if (m <= 0) break;
        const int k1 = k0 + nb;
        double *A21 = A + (long)k1 * n + k0;
        {   // P: W = A21 X^T (m x nb, K = nb)
            GemmDesc p;
            p.M = m; p.N = nb; p.K = nb;
            p.A = A21; p.lda = n; p.B = X; p.ldb = NBO; p.transB = true; p.C = W; p.ldc = LDW;
            p.prof_name = "potrf_panel";
            gemm_f64(c, p, s);
        }
        const int na = std::min(NBO, m);                               // T_a: the next panel's columns (all rows)
        GemmDesc ta;
        ta.M = m; ta.N = na; ta.K = nb;
        ta.A = W; ta.lda = LDW; ta.B = W; ta.ldb = LDW; ta.transB = true;
        ta.C = A + (long)k1 * n + k1; ta.ldc = n;
        ta.epi = EPI_SUB; ta.lower = true; ta.cfg = t_cfg;
        ta.prof_name = "potrf_syrk_next_panel";
        gemm_f64(c, ta, s);
        // D of the next step beside T_b.  Its two factor + invert workgroups need a whole CU each, and under the update's tiles no
        // CU ever drains (a retiring tile is replaced at once: the chain then only ran when the update was over).  So the update
        // is held back by gates: T_b1 starts when the first diagonal workgroup is resident, T_b2 -- the rest -- when the second
        // one is (T_b1 is sized to end about then).  The small products of the chain fit between the tiles anyway.
        const int mm = m - na;
        unsigned int tok = 0;
        const int ndiag = (na + NB - 1) / NB;
        const double t_update_us = (double)mm * mm * nb / 48e6;            // ~48 TF/s on the tiles that run (mm^2 nb flops)
        // (measured at n = 12 000: gating every update, however short, 18.11 ms; only those above 30 / 60 / 120 us: 18.25 / 18.5 /
        // 18.8; a first part of 60 / 80 / 100 / 130 / 170 us: 18.55 / 18.27 / 18.36 / 18.19 / 18.32)
        const double tb1_us = 130.0;
        const bool gated = la && gate_on && mm > 0;
        if (la) {
            GP_HIP(hipEventRecord(c->ev_chol_a, s));
            GP_HIP(hipStreamWaitEvent(sd, c->ev_chol_a, 0));
            if (gated) {
                tok = c->chol_token + 1;
                c->chol_token += (unsigned)ndiag;
            }
            g_chain_under_update = gated && t_update_us > 250.0;       // (else the chain is what the main stream waits for: fastest tiles)
            factor_diag_block(c, A + (long)k1 * n + k1, n, na, X, tmp, Wsub, d_status, k1, sd, tok);
            g_chain_under_update = false;
            GP_HIP(hipEventRecord(c->ev_chol_d, sd));
        }
        // L21 into its place (W is reused by the next step): queued here, where the main stream would otherwise only wait for the
        // next diagonal workgroup to become resident
        hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)m * nb, 256)), dim3(256), 0, s, (const double *)W, (long)LDW, A21,
                           (long)n, m, nb);
        if (mm > 0) {                                                  // T_b: everything to the right of the next panel
            auto update_rows = [&](int r0, int r1) {                   // rows [r0, r1) of the lower-triangular update
                GemmDesc tb;
                tb.M = r1 - r0; tb.N = r1; tb.K = nb;
                tb.A = W + (long)(na + r0) * LDW; tb.lda = LDW; tb.B = W + (long)na * LDW; tb.ldb = LDW; tb.transB = true;
                tb.C = A + (long)(k1 + na + r0) * n + (k1 + na); tb.ldc = n;
                tb.epi = EPI_SUB; tb.lower = true; tb.lower_shift = r0; tb.cfg = t_cfg;
                tb.prof_name = "potrf_syrk";
                gemm_f64(c, tb, s);
            };
            if (!gated) {
                update_rows(0, mm);
            } else {
                const unsigned int max_ticks = 30000;                  // 300 us
                hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, c->h_chol_flag, tok, max_ticks);
                int r1 = mm;
                if (ndiag > 1) {                                       // T_b1: ~100 us of tiles (first diagonal launch + the two small products)
                    const double frac = std::min(0.5, tb1_us / t_update_us);
                    r1 = std::min(mm, std::max(256, (int)(mm * std::sqrt(frac)) / 64 * 64));
                }
                update_rows(0, r1);
                if (r1 < mm) {
                    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, c->h_chol_flag, tok + 1, max_ticks);
                    update_rows(r1, mm);
                }
            }
        }
        if (la) GP_HIP(hipStreamWaitEvent(s, c->ev_chol_d, 0));
        else factor_diag_block(c, A + (long)k1 * n + k1, n, na, X, tmp, Wsub, d_status, k1, s);
    }
    hipLaunchKernelGGL(zero_upper_kernel, dim3(ceil_div((long)n * n, 256)), dim3(256), 0, s, A, n);
    GP_HIP(hipGetLastError());
}

// measurement aid (tools/potrf_probe.py): one factor + invert launch on the leading 128 x 128 block of A with phase stamps
void potrf_diag128_probe(gpcsd_ctx *c, double *A, int n, double *X, int *d_status, unsigned long long *clk_dev, hipStream_t s) {
    launch_diag(A, n, std::min(n, 128), X, 128, d_status, 0, true, s, clk_dev);
    GP_HIP(hipGetLastError());
}

void trsm_lower_device(gpcsd_ctx *c, const double *L, int n, double *B, int nrhs, hipStream_t s) {
    ProfScope ps(c, "trsm", (double)n * n * nrhs, s);
    double *X = c->buf<double>("chol_Linv", (size_t)NBO * NBO);
    double *tmp = c->buf<double>("chol_inv_tmp", (size_t)NBO * NBO);
    double *Xb = c->buf<double>("trsm_xblk", (size_t)NBO * nrhs);
    for (int k0 = 0; k0 < n; k0 += NBO) {
        const int nb = std::min(NBO, n - k0);
        const double *Lkk = L + (long)k0 * n + k0;
        for (int p0 = 0; p0 < nb; p0 += NB)
            launch_diag(const_cast<double *>(Lkk) + (long)p0 * n + p0, (long)n, std::min(NB, nb - p0), X + (long)p0 * NBO + p0, (long)NBO,
                        nullptr, 0, false, s);
        assemble_block_inverse(c, Lkk, n, nb, X, tmp, s);
        // X_blk = inv(L11) B_blk (the strictly upper part of the inverse buffer is stale: zero it through K-limited products)
        // -- row block i of the inverse has its non-zeros in columns [0, NB (i + 1)): one product per NB-row block
        for (int p0 = 0; p0 < nb; p0 += NB) {
            const int b = std::min(NB, nb - p0);
            small_gemm(c, b, nrhs, p0 + b, X + (long)p0 * NBO, NBO, B + (long)k0 * nrhs, nrhs, false, Xb + (long)p0 * nrhs, nrhs, 1.0,
                       EPI_STORE, "trsm_gemm", s);
        }
        hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)nb * nrhs, 256)), dim3(256), 0, s, (const double *)Xb, (long)nrhs,
                           B + (long)k0 * nrhs, (long)nrhs, nb, nrhs);
        const int rows = n - (k0 + nb);
        if (rows > 0)                                     // B[below] -= L21 X_blk
            small_gemm(c, rows, nrhs, nb, L + (long)(k0 + nb) * n + k0, n, Xb, nrhs, false, B + (long)(k0 + nb) * nrhs, nrhs, -1.0,
                       EPI_ACCUM, "trsm_gemm", s);
    }
    GP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void logdet_chol_kernel(const double *__restrict__ L, int n, double *out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += log(L[(long)i * n + i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = 2.0 * sh[0];
}

void logdet_chol_device(gpcsd_ctx *c, const double *L, int n, double *out, hipStream_t s) {
    hipLaunchKernelGGL(logdet_chol_kernel, dim3(1), dim3(256), 0, s, L, n, out);
    GP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const double *__restrict__ x, long n, double *partials) {
    __shared__ double sh[256];
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ p, int n, double *out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

void sumsq_device(gpcsd_ctx *c, const double *x, long n, double *out, hipStream_t s) {
    int blocks = (int)((n + 2047) / 2048);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    double *part = c->buf<double>("sumsq_partials", 256);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, x, n, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double *)part, blocks, out);
    GP_HIP(hipGetLastError());
}

}  // namespace gpcsd
