// Large-n symmetric eigensolver for gfx950 (replaces LAPACK dsyevd behind numpy.linalg.eigh at
// src/gpcsd/utility_functions.py:58-59).  Three stages, all on the device:
//
//  1. sytrd  -- Householder tridiagonalisation, ONE launch per column, batched over independent problems.
//               Every workgroup owns a slab of trailing rows (full symmetric storage, ping-pong buffers); it redundantly
//               rebuilds w from the previous step's y = A v, applies the rank-2 update to its slab, derives the next
//               reflector from the updated pivot row and accumulates its part of the next y in the same pass.  No
//               atomics, no grid barriers, deterministic; the only cross-workgroup hand-off is the kernel boundary.
//  2. stedc  -- Cuppen divide & conquer on the tridiagonal: Jacobi leaves in LDS, then per level: deflation scan,
//               secular equation (one wave per root, origin-shifted), Gu/Eisenstat z-hat (Loewner) for orthogonality,
//               eigenvector block via the fp64 MFMA GEMM with the deflated size read on the device, rank-sort merge.
//  3. ormtr  -- back-transformation Z <- (H_0 ... H_{n-3}) Z in compact-WY panels of 64 reflectors: T factors for all
//               panels in one batched GEMM + one launch, then two MFMA GEMMs per panel.
//
// tools/dc_prototype.py is the NumPy model of stage 2 used to validate the algorithm before porting.
#include <algorithm>
#include <cstring>
#include <vector>

#include "devutil.hpp"
#include "kernels.hpp"

namespace gpcsd {

constexpr int SY_RPW = 8;            // trailing rows per workgroup in the sytrd step kernel
constexpr int WY_NB = 64;            // reflectors per compact-WY panel
constexpr int AMAX_PARTS = 64;

// stedc.hip
struct StedcProb {               // one class of tridiagonal problems (replicas: d / e s_in apart, outputs sw / sZ apart)
    const double *d, *e;
    int n;
    double *w, *Z;
    std::string tag;
    int count = 1;
    long s_in = 0, sw = 0, sZ = 0;
};
// wy != nullptr: the leaf launch also forms the T factors of these back-transformations (wy_batch_device: prep_done)
void stedc_batch_device(gpcsd_ctx *c, StedcProb *probs, int nclass, int *d_status, int status_stride, hipStream_t s,
                        const WyBatch *wy = nullptr);

// ------------------------------------------------------------------------------------------------------------------
// stage 0: scaling  A0 = A / max|A|   (two launches: per-workgroup partial maxima, then scale + final maximum)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_partial_kernel(const double *__restrict__ A, long n2, double *part) {
    __shared__ double red[4];
    double m = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) m = fmax(m, fabs(A[i]));
    m = block_max256(m, red);
    if (threadIdx.x == 0) part[2 + blockIdx.x] = m;
}
// amax[0] = final maximum (written by workgroup 0 for the eigenvalue rescale), amax[2..] = partials
__global__ __launch_bounds__(256) void scale_copy_kernel(const double *__restrict__ A, long n2, double *amax,
                                                         double *__restrict__ out) {
    double m = 0.0;
    for (int i = 0; i < AMAX_PARTS; ++i) m = fmax(m, amax[2 + i]);
    m = (m > 0.0 && m <= 1.7e308) ? m : 1.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) amax[0] = m;
    const double inv = 1.0 / m;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) out[i] = A[i] * inv;
}
// eigenvalues of the scaled matrices back to the scale of the inputs, all classes and replicas in one launch (stage 2 of a
// staged solve; elsewhere the rescale rides in the back-transformation's apply launch)
__global__ void scale_w_batch_kernel(WyBatch b) {
    const WyProb P = wy_resolve(b, blockIdx.y);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (P.w_scale && i < P.n) P.w_scale[i] *= P.amax[0];
}
__global__ void scale_vec_kernel(double *w, int n, const double *__restrict__ amax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] *= amax[0];
}

// ------------------------------------------------------------------------------------------------------------------
// stage 1: tridiagonalisation
// ------------------------------------------------------------------------------------------------------------------
// One class of problems (see EigReq): the pointers are those of replica 0, every buffer of replica r sits r * blk doubles
// further (all of them are slices of one arena per class).
struct SytrdProb {
    double *A0, *A1;      // ping-pong full symmetric storage (n x n)
    double *V;            // (n + WY_NB) x n, row k = reflector k (zero for j <= k, V[k][k+1] = 1)
    double *tau, *d, *e;  // n each
    double *y0, *y1;      // ping-pong A v products
    int n;
    int k_tail;           // first column handled by the in-LDS tail kernel (n - 1: no tail)
    int psd;              // the matrix is a Gram matrix (positive semi-definite up to rounding): the tail may stop early (sytrd_regtail.hpp)
    long blk;             // replica stride of the arena, in doubles
    int pipe;             // the register tail publishes its progress panel by panel (sy_progress_word; whole problems only)
};
// the progress word of a problem: behind tau's n + WY_NB entries (the slice has two more; cleared with tau by whoever fills the arena)
__host__ __device__ inline unsigned *sy_progress_word(const SytrdProb &P) { return reinterpret_cast<unsigned *>(P.tau + P.n + WY_NB); }
struct SytrdBatch {
    SytrdProb p[MAX_BATCH];
    int start[MAX_BATCH + 1];
    // measurement (gpcsd_prof_enable modes 2 / 3): workgroup g of the single-workgroup tail stamps the device's wall clock at
    // its start and end into clk[2 g], clk[2 g + 1] (host-mapped memory) -- the one way to time this kernel inside a replayed
    // hipGraph, where event scopes cannot record.  nullptr: off.
    unsigned long long *clk;
};
__device__ __forceinline__ SytrdProb sy_resolve(const SytrdBatch &b, int g) {
    int cls, rep;
    class_of(b.start, g, cls, rep);
    SytrdProb P = b.p[cls];
    const long o = rep * P.blk;
    P.A0 += o; P.A1 += o; P.V += o; P.tau += o; P.d += o; P.e += o; P.y0 += o; P.y1 += o;
    return P;
}

}  // namespace gpcsd
#include "sytrd_regtail.hpp"
namespace gpcsd {

// JQ = ceil(trailing columns / 64) this launch may need (2, 4, 8, 16, 32 or 64).  Every global load of the step -- the slab
// rows, the pivot row, the previous reflector and its y -- is issued before the first barrier, so a step costs one
// L2/Infinity-Cache round trip plus LDS reductions instead of a chain of dependent loads.
template <int JQ>
__global__ __launch_bounds__(256) void sytrd_step_kernel(SytrdBatch b, int k) {
    const SytrdProb P = sy_resolve(b, blockIdx.y);
    const int n = P.n;
    if (k >= P.k_tail) return;
    const int row0 = k + 1 + blockIdx.x * SY_RPW;
    if (row0 >= n) return;
    constexpr int PQ = JQ / 4 + 1;
    const double *__restrict__ Ain = (k & 1) ? P.A1 : P.A0;
    double *__restrict__ Aout = (k & 1) ? P.A0 : P.A1;
    const double *__restrict__ yin = (k & 1) ? P.y1 : P.y0;
    double *__restrict__ yout = (k & 1) ? P.y0 : P.y1;
    __shared__ double sv[EIG_MAXN], sw[EIG_MAXN], svn[EIG_MAXN];
    __shared__ double red[4];
    __shared__ double s_dk, s_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rend = min(row0 + SY_RPW, n);

    // ---- all global loads up front ----
    double av[2][JQ];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = row0 + wid + 4 * r;
        const double *__restrict__ ai = Ain + (long)(i < rend ? i : row0) * n;
#pragma unroll
        for (int q = 0; q < JQ; ++q) {
            const int j = k + 1 + lane + 64 * q;
            av[r][q] = (i < rend && j < n) ? ai[j] : 0.0;
        }
    }
    double pr[PQ], pv[PQ], py[PQ];
    const double *__restrict__ arow = Ain + (long)k * n;
    const double *__restrict__ vp = P.V + (long)(k > 0 ? k - 1 : 0) * n;
    const double taup = (k > 0) ? P.tau[k - 1] : 0.0;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int j = k + tid + 256 * q;
        const bool ok = j < n;
        pr[q] = ok ? arow[j] : 0.0;
        pv[q] = (ok && k > 0) ? vp[j] : 0.0;
        py[q] = (ok && k > 0) ? yin[j] : 0.0;
    }

    // 1. w of the previous reflector: w = tau*y - (tau^2 (y.v)/2) v, on indices [k, n)
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < PQ; ++q) part += pv[q] * py[q];
    const double dot = block_sum256(part, red);
    const double cc = 0.5 * taup * taup * dot;
    double pw[PQ];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int j = k + tid + 256 * q;
        pw[q] = taup * py[q] - cc * pv[q];
        if (j < n) {
            sv[j] = pv[q];
            sw[j] = pw[q];
        }
    }
    __syncthreads();
    // 2. updated pivot row k -> d_k and the new reflector from x = a'[k, k+1:]
    const double vk = sv[k], wk = sw[k];
    double pa[PQ];
    part = 0.0;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int j = k + tid + 256 * q;
        pa[q] = pr[q] - vk * pw[q] - wk * pv[q];
        if (j < n && j >= k + 2) part += pa[q] * pa[q];
    }
    if (tid == 0) s_dk = pa[0];
    if (tid == 1) s_alpha = pa[0];
    const double xnorm2 = block_sum256(part, red);      // its barriers also publish s_dk / s_alpha
    const double dk = s_dk;
    const double alpha = s_alpha;
    double tau = 0.0, beta = alpha, scal = 0.0;
    // The matrix is scaled to max|a| = 1: a column with |x|^2 + alpha^2 < 1e-100 is rounding noise of rounding noise (exactly
    // low-rank inputs shrink by 1e-15 per column); its squares lose their bits as subnormals and the reflector built from them
    // is not orthogonal, so it counts as a zero column like in the single-workgroup tail (sytrd_regtail.hpp).
    if (k <= n - 3 && xnorm2 > 0.0 && alpha * alpha + xnorm2 > 1e-100) {
        beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
    }
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int j = k + tid + 256 * q;
        if (j < n && j >= k + 1) svn[j] = (j == k + 1) ? 1.0 : pa[q] * scal;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        double *vrow = P.V + (long)k * n;
        for (int j = tid; j < n; j += 256) vrow[j] = (j >= k + 1) ? svn[j] : 0.0;
        if (tid == 0) {
            P.d[k] = dk;
            P.e[k] = beta;
            P.tau[k] = tau;
        }
    }
    // 3. slab rows from registers: rank-2 update, store, and this slab's entries of y = A' v_new
    double acc0 = 0.0, acc1 = 0.0;
    const int i0 = row0 + wid, i1 = row0 + wid + 4;
    const double vi0 = (i0 < rend) ? sv[i0] : 0.0, wi0 = (i0 < rend) ? sw[i0] : 0.0;
    const double vi1 = (i1 < rend) ? sv[i1] : 0.0, wi1 = (i1 < rend) ? sw[i1] : 0.0;
#pragma unroll
    for (int q = 0; q < JQ; ++q) {
        const int j = k + 1 + lane + 64 * q;
        if (j < n) {
            const double swj = sw[j], svj = sv[j], vnj = svn[j];
            if (i0 < rend) {
                const double a = av[0][q] - vi0 * swj - wi0 * svj;
                Aout[(long)i0 * n + j] = a;
                acc0 += a * vnj;
            }
            if (i1 < rend) {
                const double a = av[1][q] - vi1 * swj - wi1 * svj;
                Aout[(long)i1 * n + j] = a;
                acc1 += a * vnj;
            }
        }
    }
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    if (lane == 0) {
        if (i0 < rend) yout[i0] = acc0;
        if (i1 < rend) yout[i1] = acc1;
    }
}

// d[n-1] after the final step: the single trailing element lives in the buffer written by step n-2
__global__ void sytrd_last_diag_kernel(SytrdBatch b) {
    const SytrdProb P = sy_resolve(b, blockIdx.x);
    if (threadIdx.x != 0 || P.n < 1 || P.k_tail < P.n - 1) return;
    const int n = P.n;
    if (n == 1) {
        P.d[0] = P.A0[0];
        return;
    }
    const int k = n - 2;                                  // last step wrote Aout of parity k
    const double *Aout = (k & 1) ? P.A0 : P.A1;
    P.d[n - 1] = Aout[(long)(n - 1) * n + (n - 1)];
    P.e[n - 1] = 0.0;
    P.tau[n - 1] = 0.0;
    if (n >= 2) P.tau[n - 2] = 0.0;
}

// rows the single-workgroup tail (sytrd_regtail.hpp) can hold: 192 in registers + up to 64 strip rows in LDS
static int sy_regtail_rows() { return RT_TMAX; }
int eigh_regtail_rows() { return RT_TMAX; }

static void sytrd_batch_launch(gpcsd_ctx *c, const SytrdBatch &b, int nclass, int nmax, hipStream_t s) {
    int klast = -1;                                        // last column handled by per-column launches
    bool any_tail = false, all_tail = true;
    for (int i = 0; i < nclass; ++i) {
        klast = std::max(klast, b.p[i].k_tail - 1);
        if (b.p[i].k_tail < b.p[i].n - 1) any_tail = true; else all_tail = false;
    }
    const int count = b.start[MAX_BATCH];                  // workgroup rows: every replica of every class
    for (int k = 0; k <= klast; ++k) {
        const int m = nmax - k - 1;
        dim3 grid(ceil_div(m, SY_RPW), count);
        if (m <= 128) hipLaunchKernelGGL(sytrd_step_kernel<2>, grid, dim3(256), 0, s, b, k);
        else if (m <= 256) hipLaunchKernelGGL(sytrd_step_kernel<4>, grid, dim3(256), 0, s, b, k);
        else if (m <= 512) hipLaunchKernelGGL(sytrd_step_kernel<8>, grid, dim3(256), 0, s, b, k);
        else if (m <= 1024) hipLaunchKernelGGL(sytrd_step_kernel<16>, grid, dim3(256), 0, s, b, k);
        else if (m <= 2048) hipLaunchKernelGGL(sytrd_step_kernel<32>, grid, dim3(256), 0, s, b, k);
        else hipLaunchKernelGGL(sytrd_step_kernel<64>, grid, dim3(256), 0, s, b, k);
    }
    static_assert(EIG_MAXN <= 64 * 64, "the widest step kernel covers 64 column chunks of 64");
    if (any_tail) {
        size_t sh = 0;
        for (int i = 0; i < nclass; ++i)
            if (b.p[i].k_tail < b.p[i].n - 1) sh = std::max(sh, rt_strip_bytes(b.p[i].n - b.p[i].k_tail));
        static PerDeviceOnce rt_attr;
        if (rt_attr.first())
            GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sytrd_rtail_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)rt_strip_bytes(RT_TMAX)));
        // profiled on its own: this single launch (one workgroup per problem) is the largest share of the GPU time of an
        // evaluation; flops = (4/3) T^3 per problem, the nominal count of a Householder tridiagonalisation
        double fl = 0.0;
        for (int i = 0; i < nclass; ++i) {
            const double T = b.p[i].n - b.p[i].k_tail;
            if (b.p[i].k_tail < b.p[i].n - 1) fl += 4.0 / 3.0 * T * T * T * (b.start[i + 1] - b.start[i]);
        }
        ProfScope ps(c, "sytrd_rtail", fl, s);
        SytrdBatch bt = b;
        bt.clk = nullptr;
        if (c->prof_mode >= 2 && c->tail_clk_dev && count <= gpcsd_ctx::TAIL_CLK_WGS) {
            const int region = (s == c->stream2) ? 0 : (s == c->stream3) ? 1 : 2;      // temporal chain, spatial chain, other
            bt.clk = c->tail_clk_dev + (size_t)region * 2 * gpcsd_ctx::TAIL_CLK_WGS;
            c->tail_clk_count[region] = count;
            c->tail_clk_flops[region] = fl;
        }
        hipLaunchKernelGGL(sytrd_rtail_kernel, dim3(count), dim3(RT_NTH), sh, s, bt);
    }
    if (!all_tail) hipLaunchKernelGGL(sytrd_last_diag_kernel, dim3(count), dim3(64), 0, s, b);
    GP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// stage 3: back-transformation with compact-WY panels
// ------------------------------------------------------------------------------------------------------------------
// For H_a ... H_b = I - V T V^T (forward, columnwise) the inverse of the upper-triangular T is known in closed form:
// T^{-1} = diag(1/tau) + striu(V^T V).  Column c of T solves U x = e_c by back substitution in axpy form; one WAVE per
// column with lane l holding the running right-hand side b_l: x_j = tau_j b_j is broadcast by a lane read, lanes l < j
// subtract G[l][j] x_j.  No barriers, 64 steps of one shuffle + one LDS read.  tau_j = 0 (H_j = I, and the zero padding
// rows of the last panel) gives x_j = 0, i.e. a zero row/column of T.  grid = (16, panels), 4 waves per workgroup.
__global__ __launch_bounds__(256) void wy_T_kernel(const double *__restrict__ G, const double *__restrict__ tau, int nrefl,
                                                   double *__restrict__ Tout) {
    const int p = blockIdx.y;
    __shared__ double g[WY_NB][WY_NB + 1];
    __shared__ double st[WY_NB];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int idx = tid; idx < WY_NB * WY_NB; idx += 256) g[idx / WY_NB][idx % WY_NB] = G[(long)p * WY_NB * WY_NB + idx];
    if (tid < WY_NB) {
        const int kk = p * WY_NB + tid;
        st[tid] = (kk < nrefl) ? tau[kk] : 0.0;
    }
    __syncthreads();
    const int c = blockIdx.x * 4 + (tid >> 6);           // column of T owned by this wave
    double b = (lane == c) ? 1.0 : 0.0;                  // right-hand side e_c
    double x = 0.0;
    for (int j = c; j >= 0; --j) {
        const double xj = st[j] * __shfl(b, j, 64);
        if (lane == j) x = xj;
        if (lane < j) b -= g[lane][j] * xj;
    }
    Tout[(long)p * WY_NB * WY_NB + (long)lane * WY_NB + c] = (lane <= c) ? x : 0.0;
}

static void ormtr_device(gpcsd_ctx *c, const double *V, const double *tau, int n, double *Z, hipStream_t s,
                         const std::string &T) {
    const int nrefl = n - 2;
    if (nrefl <= 0) return;
    const int P = ceil_div(nrefl, WY_NB);
    double *G = c->buf<double>(T + "wyG", (size_t)P * WY_NB * WY_NB);
    double *Tm = c->buf<double>(T + "wyT", (size_t)P * WY_NB * WY_NB);
    double *VT = c->buf<double>(T + "wyVT", (size_t)P * n * WY_NB);
    double *W1 = c->buf<double>(T + "wyW1", (size_t)WY_NB * n);
    GemmDesc gg;                                  // G_p = V_p V_p^T
    gg.M = WY_NB; gg.N = WY_NB; gg.K = n;
    gg.A = V; gg.lda = n; gg.B = V; gg.ldb = n; gg.transB = true;
    gg.C = G; gg.ldc = WY_NB;
    gg.batch = P; gg.sA = (long)WY_NB * n; gg.sB = (long)WY_NB * n; gg.sC = (long)WY_NB * WY_NB;
    gg.prof_name = "gemm_wy_gram";
    gemm_f64(c, gg, s);
    hipLaunchKernelGGL(wy_T_kernel, dim3(WY_NB / 4, P), dim3(256), 0, s, (const double *)G, tau, nrefl, Tm);
    GP_HIP(hipGetLastError());
    GemmDesc gv;                                  // VT_p = V_p^T T_p   (n x nb)
    gv.M = n; gv.N = WY_NB; gv.K = WY_NB;
    gv.A = V; gv.lda = n; gv.transA = true; gv.B = Tm; gv.ldb = WY_NB;
    gv.C = VT; gv.ldc = WY_NB;
    gv.batch = P; gv.sA = (long)WY_NB * n; gv.sB = (long)WY_NB * WY_NB; gv.sC = (long)n * WY_NB;
    gv.prof_name = "gemm_wy_vt";
    gemm_f64(c, gv, s);
    for (int p = P - 1; p >= 0; --p) {
        GemmDesc g1;                              // W1 = V_p Z  (nb x n)
        g1.M = WY_NB; g1.N = n; g1.K = n;
        g1.A = V + (long)p * WY_NB * n; g1.lda = n; g1.B = Z; g1.ldb = n; g1.C = W1; g1.ldc = n;
        g1.prof_name = "gemm_wy_vz";
        gemm_f64(c, g1, s);
        GemmDesc g2;                              // Z -= VT_p W1
        g2.M = n; g2.N = n; g2.K = WY_NB;
        g2.A = VT + (long)p * n * WY_NB; g2.lda = WY_NB; g2.B = W1; g2.ldb = n; g2.C = Z; g2.ldc = n;
        g2.alpha = -1.0; g2.epi = EPI_ACCUM;
        g2.prof_name = "gemm_wy_update";
        gemm_f64(c, g2, s);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------------------------
struct EigProb {                 // one class: `count` replicas, inputs / outputs sA / sw / sZ apart
    double *A, *w, *Z;
    int n;
    std::string tag;
    int count = 1;
    long sA = 0, sw = 0, sZ = 0;
    bool prefilled = false;      // EigReq::prefilled
    double *amax;                // 2 + AMAX_PARTS doubles per replica (in the arena)
    double *wyT;                 // T factors of the compact-WY panels (in the arena)
    SytrdProb sp;
};

// Workspace of a class: ONE allocation holding `count` identical blocks, so that replica r of every buffer is the
// replica-0 pointer + r * blk (what sy_resolve / wy_resolve / the prep kernels add).  Slices are 16-byte aligned.
struct ArenaLayout {
    size_t off = 0;
    size_t take(size_t ndoubles) {
        const size_t o = off;
        off += (ndoubles + 1) & ~(size_t)1;
        return o;
    }
};

// the class arena (grow-only: the same (tag, n, count) always maps to the same slices)
static void layout_arena(gpcsd_ctx *c, const std::string &tag, int n, int count, SytrdProb &sp, double *&amax, double *&wyT) {
    const size_t nn = (size_t)n * n;
    const std::string T = "eig_" + tag + "_";
    const int npanels = std::max(1, ceil_div(std::max(n - 2, 1), WY_NB));
    ArenaLayout L;
    const size_t oA0 = L.take(nn), oA1 = L.take(nn), oV = L.take((size_t)(n + WY_NB) * n), otau = L.take(n + WY_NB + 2);
    const size_t od = L.take(n), oe = L.take(n), oy0 = L.take(n), oy1 = L.take(n), oamax = L.take(2 + AMAX_PARTS);
    const size_t oT = L.take((size_t)npanels * WY_NB * WY_NB);
    double *base = c->buf<double>(T + "arena", L.off * (size_t)std::max(count, 1));
    sp.blk = (long)L.off;
    sp.A0 = base + oA0; sp.A1 = base + oA1; sp.V = base + oV; sp.tau = base + otau;
    sp.d = base + od; sp.e = base + oe; sp.y0 = base + oy0; sp.y1 = base + oy1;
    amax = base + oamax;
    wyT = base + oT;
}

EigArenaView eigh_arena_view(gpcsd_ctx *c, const char *tag, int n, int count) {
    SytrdProb sp{};
    double *amax = nullptr, *wyT = nullptr;
    layout_arena(c, tag, n, count, sp, amax, wyT);
    return EigArenaView{sp.A0, sp.V, sp.tau, amax, sp.blk, sp.d, sp.e, &c->arena_psd[tag]};
}

double *eigh_Q_view(gpcsd_ctx *c, const char *tag, int n, int count) {
    return c->buf<double>(std::string("eig_") + tag + "_Q", (size_t)n * n * std::max(count, 1));
}

static void prep_problem(gpcsd_ctx *c, EigProb &p, hipStream_t s) {
    const int n = p.n;
    p.sp.n = n;
    {
        // the trailing block finishes inside one workgroup (192 rows in registers + up to 64 strip rows in LDS)
        p.sp.k_tail = std::max(0, n - sy_regtail_rows());
    }
    layout_arena(c, p.tag, n, p.count, p.sp, p.amax, p.wyT);
    // positive semi-definite only on the word of whoever filled the arena (EigArenaView::psd); the library's own scaling pass
    // copies a caller's matrix, about which nothing is known
    bool &psd = c->arena_psd[p.tag];
    if (!p.prefilled) psd = c->claim_psd;         // (... unless the caller vouches for it: gpcsd_eigh_psd)
    p.sp.psd = (psd && c->tail_early_exit) ? 1 : 0;
    // progress words on the word of the caller (gpcsd_ctx::pipe_req: stage 5 of a staged chain follows on another stream)
    p.sp.pipe = (c->pipe_req && p.sp.k_tail == 0) ? 1 : 0;
    (void)s;
}

// scaling, copy into the ping-pong buffer and zeroing of the reflector storage for ALL problems in two launches
struct PrepBatch {
    const double *A[MAX_BATCH];
    long sA[MAX_BATCH];
    double *amax[MAX_BATCH];    // in the class arena: replica stride sp[].blk
    SytrdProb sp[MAX_BATCH];
    double *w[MAX_BATCH];       // eigenvalue outputs (for the final rescale)
    long sw[MAX_BATCH];
    int start[MAX_BATCH + 1];
    int skip[MAX_BATCH];        // the class is prefilled (EigReq::prefilled): its arena already holds the scaled matrix
    int *status;                // numerical-failure words of the call (non-finite input is reported there) ...
    int status_stride;          // ... one per replica index when != 0
};
struct PrepView {
    const double *A;
    double *amax, *w;
    SytrdProb P;
    int *status;
    bool skip;
};
__device__ __forceinline__ PrepView prep_resolve(const PrepBatch &b, int g) {
    int cls, rep;
    class_of(b.start, g, cls, rep);
    PrepView v;
    v.P = b.sp[cls];
    const long o = rep * v.P.blk;
    v.P.A0 += o; v.P.A1 += o; v.P.V += o; v.P.tau += o; v.P.d += o; v.P.e += o; v.P.y0 += o; v.P.y1 += o;
    v.A = b.A[cls] + rep * b.sA[cls];
    v.amax = b.amax[cls] + o;
    v.w = b.w[cls] ? b.w[cls] + rep * b.sw[cls] : nullptr;
    v.status = b.status ? b.status + (long)rep * b.status_stride : nullptr;
    v.skip = b.skip[cls] != 0;
    return v;
}
__global__ __launch_bounds__(256) void absmax_partial_batch_kernel(PrepBatch b) {
    __shared__ double red[4];
    const PrepView v = prep_resolve(b, blockIdx.y);
    if (v.skip) return;
    const long n2 = (long)v.P.n * v.P.n;
    const double *__restrict__ A = v.A;
    double m = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) m = fmax(m, fabs(A[i]));
    m = block_max256(m, red);
    if (threadIdx.x == 0) v.amax[2 + blockIdx.x] = m;
}
__global__ __launch_bounds__(256) void scale_copy_zero_batch_kernel(PrepBatch b) {
    const PrepView v = prep_resolve(b, blockIdx.y);
    if (v.skip) return;
    const SytrdProb &P = v.P;
    const long n = P.n, n2 = n * n;
    double *amax = v.amax;
    double m = 0.0;
    for (int i = 0; i < AMAX_PARTS; ++i) m = fmax(m, amax[2 + i]);
    m = (m > 0.0 && m <= 1.7e308) ? m : 1.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) amax[0] = m;
    const double inv = 1.0 / m;
    const double *__restrict__ A = v.A;
    const long stride = (long)gridDim.x * 256, i0 = blockIdx.x * 256L + threadIdx.x;
    // Non-finite input (a NaN hyper-parameter makes a whole Gram matrix NaN) must not reach the solver: its rank sorts and
    // merges assume a total order and would index out of range.  Such entries are zeroed and the call reports failure.
    bool bad = false;
    for (long i = i0; i < n2; i += stride) {
        double x = (m > 1e300) ? A[i] / m : A[i] * inv;         // 1 / m is subnormal beyond 1e300
        if (!(fabs(x) <= 2.0)) {
            x = 0.0;
            bad = true;
        }
        P.A0[i] = x;
    }
    if (bad && v.status) atomicMax(v.status, 4);
    const long nv = (n + WY_NB) * n;
    for (long i = i0; i < nv; i += stride) P.V[i] = 0.0;
    for (long i = i0; i < n + WY_NB + 2; i += stride) P.tau[i] = 0.0;     // (+2: the progress word)
}

static PrepBatch prep_batch_launch(gpcsd_ctx *c, EigProb *probs, int nclass, hipStream_t s, int *d_status = nullptr,
                                   int status_stride = 0, bool launch = true) {
    PrepBatch pb{};
    pb.status = d_status;
    pb.status_stride = status_stride;
    int total = 0;
    for (int i = 0; i < MAX_BATCH; ++i) {
        pb.start[i] = total;
        if (i >= nclass) continue;
        prep_problem(c, probs[i], s);
        pb.A[i] = probs[i].A;
        pb.sA[i] = probs[i].sA;
        pb.amax[i] = probs[i].amax;
        pb.sp[i] = probs[i].sp;
        pb.w[i] = probs[i].w;
        pb.sw[i] = probs[i].sw;
        pb.skip[i] = probs[i].prefilled ? 1 : 0;
        total += std::max(probs[i].count, 1);
    }
    pb.start[MAX_BATCH] = total;
    bool any = false;
    for (int i = 0; i < nclass; ++i) any = any || !probs[i].prefilled;
    if (any && launch) {                        // (every class prefilled: the chain starts at the tridiagonalisation)
        hipLaunchKernelGGL(absmax_partial_batch_kernel, dim3(AMAX_PARTS, total), dim3(256), 0, s, pb);
        hipLaunchKernelGGL(scale_copy_zero_batch_kernel, dim3(128, total), dim3(256), 0, s, pb);
        GP_HIP(hipGetLastError());
    }
    return pb;
}

static SytrdBatch sytrd_batch_of(const PrepBatch &pb) {
    SytrdBatch b{};
    b.clk = nullptr;
    for (int i = 0; i < MAX_BATCH; ++i) b.p[i] = pb.sp[i];
    for (int i = 0; i <= MAX_BATCH; ++i) b.start[i] = pb.start[i];
    return b;
}

void sytrd_device(gpcsd_ctx *c, double *A, int n, double *d, double *e, double *V, double *tau, hipStream_t s) {
    GP_REQUIRE(n >= 1 && n <= EIG_MAXN, -3, "sytrd: n=%d outside [1,%d]", n, EIG_MAXN);
    EigProb p;
    p.A = A; p.n = n; p.tag = "dbg"; p.w = nullptr; p.Z = nullptr;
    const PrepBatch pb = prep_batch_launch(c, &p, 1, s);
    sytrd_batch_launch(c, sytrd_batch_of(pb), 1, n, s);
    GP_HIP(hipMemcpyAsync(d, p.sp.d, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(e, p.sp.e, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(tau, p.sp.tau, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GP_HIP(hipMemcpyAsync(V, p.sp.V, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, s));
    // d, e are those of A / max|A|; rescale so the caller sees the tridiagonal of A itself
    hipLaunchKernelGGL(scale_vec_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, d, n, (const double *)p.amax);
    hipLaunchKernelGGL(scale_vec_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, e, n, (const double *)p.amax);
    GP_HIP(hipGetLastError());
}

// Test aid (gpcsd_debug_fault_stage2(ctx, 1): an explicit call on the context, never the environment): the divide & conquer stage
// of a STAGED solve reports failure 3 for every replica -- which call surfaces a failure of a stage that a tridiagonal-form
// log-likelihood does not wait for (tests/test_hip_fullsize.py::test_late_stage_failure_is_reported_by_the_call_that_joins_the_chain).
__global__ void fault_status_kernel(int *status, int stride, int count) {
    for (int r = threadIdx.x; r < count; r += blockDim.x) atomicMax(status + (long)r * stride, 3);
}

void eigh_large_batch(gpcsd_ctx *c, EigProb *probs, int nclass, int *d_status, int status_stride, hipStream_t s, int stage = 0) {
    GP_REQUIRE(nclass >= 1 && nclass <= MAX_BATCH, -3, "eigh: %d problem classes outside [1,%d]", nclass, MAX_BATCH);
    int nmax = 0;
    bool replicated = false;
    for (int i = 0; i < nclass; ++i) {
        GP_REQUIRE(probs[i].n > 2 && probs[i].n <= EIG_MAXN, -3, "eigh(large): n=%d outside (2,%d]", probs[i].n, EIG_MAXN);
        nmax = std::max(nmax, probs[i].n);
        replicated = replicated || probs[i].count > 1;
    }
    const PrepBatch pb = prep_batch_launch(c, probs, nclass, s, d_status, status_stride, /*launch=*/stage < 2);
    if (stage < 2) {
        ProfScope ps(c, "eigh_sytrd", 0.0, s);
        sytrd_batch_launch(c, sytrd_batch_of(pb), nclass, nmax, s);
    }
    const bool wy_fused = wy_fused_supported(nmax);
    GP_REQUIRE(stage == 0 || wy_fused, -3, "eigh: staged solves need the fused back-transformation (n=%d)", nmax);
    (void)replicated;
    WyBatch wb{};
    for (int i = 0; i <= MAX_BATCH; ++i) wb.start[i] = pb.start[i];
    if (wy_fused)
        for (int i = 0; i < nclass; ++i) {
            EigProb &p = probs[i];
            const int nrefl = p.n - 2, P = ceil_div(nrefl, WY_NB);
            wb.p[i].V = p.sp.V; wb.p[i].tau = p.sp.tau; wb.p[i].Z = p.Z;
            wb.p[i].T = p.wyT;
            wb.p[i].n = p.n; wb.p[i].npanels = P; wb.p[i].nrefl = nrefl;
            wb.p[i].w_scale = pb.w[i]; wb.p[i].amax = pb.amax[i];   // the final rescale of the eigenvalues rides in the apply launch
            wb.p[i].blk = p.sp.blk; wb.p[i].sZ = p.sZ; wb.p[i].sw = p.sw;
        }
    if (stage == 1) return;                    // the tridiagonalisation alone
    if (stage == 5) {
        // Stage 3 panel by panel beside a stage 1 that is still running (wy.hip: wy_q_pipeline; the tails were launched with
        // SytrdProb::pipe), each finished block of columns of Q followed by the caller's product on it (gpcsd_ctx::q_pipe_x).
        ProfScope ps(c, "eigh_stage5_TQX", 0.0, s);
        WyBatch wq = wb;
        wq.status = d_status;
        double *Qv[MAX_BATCH];
        for (int i = 0; i < nclass; ++i) {
            GP_REQUIRE(probs[i].sp.k_tail == 0, -3, "eigh: stage 5 needs whole problems in the register tail (n=%d)", probs[i].n);
            Qv[i] = eigh_Q_view(c, probs[i].tag.c_str(), probs[i].n, probs[i].count);
            wq.p[i].Z = Qv[i];
            wq.p[i].sZ = (long)probs[i].n * probs[i].n;
            wq.p[i].w_scale = nullptr;
        }
        const gpcsd_ctx::QPipeX x = c->q_pipe_x;
        GP_REQUIRE(!x.in || nclass <= 2, -3, "eigh: stage 5 with a product serves one folded problem (two classes)");
        int stage_k = 0, maxP = 0;
        for (int i = 0; i < nclass; ++i) maxP = std::max(maxP, wq.p[i].npanels);
        // which panels are followed by a launch of the product: a 64-column launch costs a tile's whole latency (~40 us for ~26 us
        // worth of the full product), so the first two panels share one (GPCSD_QPIPE_MASK, gpcsd_ctx::q_pipe_mask)
        const int mask = c->q_pipe_mask >= 0 ? c->q_pipe_mask : ~1;
        static const int chunk_cfg = getenv("GPCSD_QPIPE_CFG") ? atoi(getenv("GPCSD_QPIPE_CFG")) : 0;
        int pend0[MAX_BATCH], pend1[MAX_BATCH];            // columns of Q finished and not yet multiplied
        for (int i = 0; i < MAX_BATCH; ++i) pend0[i] = pend1[i] = -1;
        wy_q_pipeline(c, wq, nclass, s, [&](const int *col0_in, const int *col1_in) {
            if (!x.in) return;
            const int k = stage_k;
            for (int i = 0; i < nclass; ++i)
                if (col1_in[i] > col0_in[i]) {
                    if (pend0[i] < 0) pend0[i] = col0_in[i];
                    pend1[i] = col1_in[i];
                }
            if (!((mask >> k) & 1) && k + 1 < maxP) {      // no launch behind this panel: its columns wait for the next one
                ++stage_k;
                return;
            }
            int col0[MAX_BATCH], col1[MAX_BATCH];
            for (int i = 0; i < nclass; ++i) {
                col0[i] = pend0[i] >= 0 ? pend0[i] : 0;
                col1[i] = pend0[i] >= 0 ? pend1[i] : 0;
                pend0[i] = pend1[i] = -1;
            }
            // the product on the panel's columns runs on the MAIN stream behind the panel's event: beside the previous call's large
            // products it would only take their CUs (measured: 40 us of tiles took 190), behind them it fills the main stream's idle
            // time under the end of the tail
            hipEvent_t ev = c->ev_stage[stage_k++ & 7];
            c->tl("stage 5 panel end (s4)", s);
            GP_HIP(hipEventRecord(ev, s));
            GP_HIP(hipStreamWaitEvent(c->stream, ev, 0));
            hipStream_t sx = c->stream;
            GemmDesc g[2];
            int live = 0;
            for (int i = 0; i < nclass; ++i) {
                const int np = probs[i].n, w = col1[i] - col0[i];
                if (w <= 0) continue;
                GemmDesc &d = g[live++];
                d.M = x.M; d.N = w; d.K = np;
                d.A = x.in + x.c0[i]; d.lda = x.ld;
                d.B = Qv[i] + (size_t)x.rep * np * np + col0[i]; d.ldb = np;
                d.C = x.out + x.c0[i] + col0[i]; d.ldc = x.ld;
                d.prof_name = "gemm_ll_YQ";
                // the tile configuration of the WHOLE product (as tri_times_Q launches it): a block of columns then has the bits it
                // has there -- the order of the K loop does not depend on the configuration, but nothing else should either
                const bool both_equal = nclass == 2 && probs[0].n == probs[1].n;
                d.cfg = chunk_cfg ? chunk_cfg : gemm_auto_cfg(x.M, np, np, both_equal ? 2 : 1);
            }
            if (live == 2 && g[0].N == g[1].N && g[0].K == g[1].K) {     // the two parity blocks as one batched launch
                g[0].batch = 2;
                g[0].sA = g[1].A - g[0].A; g[0].sB = g[1].B - g[0].B; g[0].sC = g[1].C - g[0].C;
                gemm_f64(c, g[0], sx);
            } else {
                for (int i = 0; i < live; ++i) gemm_f64(c, g[i], sx);
            }
            c->tl("X block end (main)", sx);
        }, /*one_launch=*/x.in == nullptr);
        GP_HIP(hipGetLastError());
        return;
    }
    if (stage == 3) {
        // T factors of the reflector panels as a launch of their own (unstaged they ride in the divide & conquer's leaf launch),
        // then Q = the panels applied to the identity (the back-transformation's apply launch, its slab of Z starting as
        // columns of I).  Off the chain's own stream: stage 2 needs neither, stage 4 the T factors.
        ProfScope ps(c, "eigh_stage3_TQ", 0.0, s);
        wy_prep_device(c, wb, nclass, s);
        WyBatch wq = wb;
        for (int i = 0; i < nclass; ++i) {
            wq.p[i].Z = eigh_Q_view(c, probs[i].tag.c_str(), probs[i].n, probs[i].count);
            wq.p[i].sZ = (long)probs[i].n * probs[i].n;
            wq.p[i].w_scale = nullptr;
            wq.p[i].z_identity = 1;
        }
        wy_batch_device(c, wq, nclass, s, /*prep_done=*/true);
        GP_HIP(hipGetLastError());
        return;
    }
    bool prep_done = false;
    if (stage != 4) {
        ProfScope ps(c, "eigh_stedc", 0.0, s);
        StedcProb sp[MAX_BATCH];
        for (int i = 0; i < nclass; ++i) {
            sp[i].d = probs[i].sp.d; sp[i].e = probs[i].sp.e; sp[i].n = probs[i].n;
            sp[i].w = probs[i].w; sp[i].Z = probs[i].Z; sp[i].tag = probs[i].tag;
            sp[i].count = std::max(probs[i].count, 1);
            sp[i].s_in = probs[i].sp.blk; sp[i].sw = probs[i].sw; sp[i].sZ = probs[i].sZ;
        }
        // the T factors of the back-transformation need the reflectors only: they ride in the leaf launch of the D&C stage
        prep_done = wy_fused;
        // (stage 2: the T factors were formed by stage 1, the leaf launch carries leaves only)
        stedc_batch_device(c, sp, nclass, d_status, status_stride, s, (prep_done && stage != 2) ? &wb : nullptr);
    }
    if (stage == 2 && c->fault_stage2 && d_status) {
        int nrep = 1;
        for (int i = 0; i < nclass; ++i) nrep = std::max(nrep, probs[i].count);
        hipLaunchKernelGGL(fault_status_kernel, dim3(1), dim3(64), 0, s, d_status, std::max(status_stride, 1), nrep);
    }
    if (stage == 2) return;                    // the divide & conquer alone: stage 4 finishes behind stage 3's T factors
    if (stage == 4) prep_done = true;
    if (wy_fused) {
        ProfScope ps(c, "eigh_backtransform", 0.0, s);
        wy_batch_device(c, wb, nclass, s, prep_done);
    } else {                                   // n too large for the LDS-resident apply kernel: GEMM chain per panel
        ProfScope ps(c, "eigh_backtransform", 0.0, s);
        // (replicas one after the other: the panel workspaces of a class are shared)
        for (int i = 0; i < nclass; ++i) {
            EigProb &p = probs[i];
            for (int r = 0; r < std::max(p.count, 1); ++r) {
                const long o = (long)r * p.sp.blk;
                ormtr_device(c, p.sp.V + o, p.sp.tau + o, p.n, p.Z + r * p.sZ, s, "eig_" + p.tag + "_");
                hipLaunchKernelGGL(scale_vec_kernel, dim3(ceil_div(p.n, 256)), dim3(256), 0, s, p.w + r * p.sw, p.n,
                                   (const double *)(p.amax + o));
            }
        }
    }
    GP_HIP(hipGetLastError());
}

void eigh_large_multi(gpcsd_ctx *c, const EigReq *reqs, int nclass, int *d_status, int status_stride, hipStream_t s, int stage) {
    static_assert(MAX_EIG_BATCH <= MAX_BATCH, "batch limits");
    GP_REQUIRE(nclass >= 1 && nclass <= MAX_BATCH, -3, "eigh: %d problem classes outside [1,%d]", nclass, MAX_BATCH);
    EigProb probs[MAX_BATCH];
    for (int i = 0; i < nclass; ++i) {
        probs[i].A = reqs[i].A; probs[i].n = reqs[i].n; probs[i].w = reqs[i].w; probs[i].Z = reqs[i].Z;
        probs[i].tag = reqs[i].tag;
        probs[i].count = std::max(reqs[i].count, 1);
        probs[i].sA = reqs[i].sA; probs[i].sw = reqs[i].sw; probs[i].sZ = reqs[i].sZ;
        probs[i].prefilled = reqs[i].prefilled;
    }
    eigh_large_batch(c, probs, nclass, d_status, status_stride, s, stage);
}

}  // namespace gpcsd
