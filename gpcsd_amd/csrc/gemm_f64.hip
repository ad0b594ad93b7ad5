// fp64 MFMA GEMM core for gfx950 (v_mfma_f64_16x16x4_f64), row-major, optional transposes, fused epilogues.
//
// Every dense contraction of the GPCSD hot path runs through this kernel:
//   K7  Ks = A Kgl A^T                      (covariances.py:90,95 / :223,231)
//   K8  Kphig = A Kcross                    (covariances.py:66-72 / :198-202)
//   K14 alpha = Qs^T Y Qt                   (gpcsd1d.py:124-125 / gpcsd2d.py:147-148)
//   K15 sum alpha^2 / D                     fused epilogue EPI_QUAD (gpcsd1d.py:126-127)
//   K16/K17 predict contractions            (gpcsd1d.py:262-285)
//
// Design (MI355X): WM x WN waves per workgroup, each wave owns FM x FN MFMA 16x16 fragments.  The large flat GEMMs
// use a 128x128 block tile with 8 waves (4x2) of 32x64: 64 accumulator VGPRs per lane, so two workgroups fit a CU
// and every SIMD always has another wave's MFMAs to issue while one waits on LDS or the barrier
// (measured: v_mfma_f64_16x16x4_f64 issues every 64 cycles per wave; 78 TFLOP/s chip-wide, tools/mfma_f64_probe).  BK = 16 (four k=4 MFMA steps) for the large tile.  Operand tiles
// are register-staged global -> LDS, double-buffered, one barrier per K tile.  LDS row strides are chosen so the
// ds_read_b64 fragment reads are bank-conflict free on the 64-bank b64 path:
//   [k][o] tiles: stride BO+16 doubles (second k row lands on the other 32 banks),
//   [o][k] tiles: stride BK+2 = 18 doubles (18*i mod 32 distinct even slots for the 16 rows of a fragment).
// f64 MFMA fragment maps (cdna_hip_programming.md section 3): A lane l holds A[l&15][l>>4], B lane l holds
// B[l>>4][l&15], C/D lane l reg r holds C[(l>>4) + 4r][l&15].
#include "kernels.hpp"

namespace gpcsd {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int PAD_KO = 16;

struct GemmK {
    int M, N, K;
    const double *A;
    long lda;
    const double *B;
    long ldb;
    double *C;
    long ldc;
    double *C2;
    double *C3;
    const double *colscale, *rowscale;
    long sA, sB, sC;
    double alpha;
    const double *D;
    int rdiv;
    long ldd;
    double *partials;
    int tiles_n;
    const int *dyn;     // optional device scalar: effective N and K (= *dyn) of this launch (D&C merge GEMMs)
};

// One operand tile: BO "outer" rows/cols (M or N side) x BK, staged by NT threads.  KMAJOR: global storage is [K][O].
// Each thread owns PER_THREAD fixed (o, k) slots of the tile.  Their global pointers are formed once (outer index
// clamped into range: a duplicated edge row only feeds accumulators the epilogue never stores) and advance by a
// constant per K tile, so a full tile costs one global_load per slot and NO address arithmetic or select -- the
// loaded registers are first touched by the LDS store after the MFMA block, which is what lets the loads overlap the
// MFMAs.  Only the last, partial K tile takes the clamped + zero-masked path.
template <int BO, bool KMAJOR, int NT, int BK>
struct Tile {
    static constexpr int LD_OK = BK + 2;       // (BK+2) mod 32 == 2 for BK = 16, 64: rows land on distinct even slots
    static constexpr int LDS_ELEMS = KMAJOR ? BK * (BO + PAD_KO) : BO * LD_OK;
    static constexpr int PER_THREAD = BO * BK / NT;
    static_assert(BO * BK % NT == 0, "tile must divide evenly over the workgroup");

    __device__ static __forceinline__ int lds_index(int o, int k) {
        return KMAJOR ? k * (BO + PAD_KO) + o : o * LD_OK + k;
    }
    __device__ static __forceinline__ void slot(int tid, int i, int &o, int &k) {
        const int e = tid + NT * i;
        if (KMAJOR) {
            k = e / BO;
            o = e % BO;
        } else {
            o = e / BK;
            k = e % BK;
        }
    }
    // pointers of this thread's slots in K tile 0
    __device__ static __forceinline__ void setup(const double *(&p)[PER_THREAD], const double *__restrict__ base, long ld,
                                                 int o0, int Olim, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int o, k;
            slot(tid, i, o, k);
            const int go = o0 + o, goc = go < Olim ? go : Olim - 1;
            p[i] = base + (KMAJOR ? (long)k * ld + goc : (long)goc * ld + k);
        }
    }
    __device__ static __forceinline__ long step(long ld) { return KMAJOR ? (long)BK * ld : (long)BK; }
    // full tile number kt: plain loads
    __device__ static __forceinline__ void gload(double (&r)[PER_THREAD], const double *const (&p)[PER_THREAD], long off) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) r[i] = p[i][off];
    }
    // partial tile starting at k0: out-of-range k reads the slot's k = K-1 element instead (always valid)
    __device__ static __forceinline__ void gload_tail(double (&r)[PER_THREAD], const double *const (&p)[PER_THREAD], long ld,
                                                      int k0, int K, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int o, k;
            slot(tid, i, o, k);
            const int gk = k0 + k, gkc = gk < K ? gk : K - 1;
            r[i] = p[i][KMAJOR ? (long)(gkc - k) * ld : (long)(gkc - k)];
        }
    }
    template <bool MASK>
    __device__ static __forceinline__ void sstore(const double (&r)[PER_THREAD], double *lds, int k0, int K, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int o, k;
            slot(tid, i, o, k);
            lds[lds_index(o, k)] = (!MASK || k0 + k < K) ? r[i] : 0.0;
        }
    }
};

template <int WM, int WN, int FM, int FN, int BK, bool TA, bool TB, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f64_kernel(GemmK g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 16 * FM * WM, BN = 16 * FN * WN;
    using TileA = Tile<BM, TA, NT, BK>;    // transA: global [K][M]
    using TileB = Tile<BN, !TB, NT, BK>;   // !transB: global [K][N]
    __shared__ double lds[2 * (TileA::LDS_ELEMS + TileB::LDS_ELEMS)];
    // buffer `b` of operand A at lds + b*LDS_ELEMS; operand B follows the two A buffers
    auto ldsA = [&](int b) -> double * { return lds + b * TileA::LDS_ELEMS; };
    auto ldsB = [&](int b) -> double * { return lds + 2 * TileA::LDS_ELEMS + b * TileB::LDS_ELEMS; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (private L2 each), so without a remap the
    // tn tiles that share one A row panel land on 8 different L2s and the panel is fetched 8 times (measured with
    // FETCH_SIZE: 740 MB read per launch for 156 MB of operands).  Give XCD x the contiguous range of logical tiles
    // [x*q + min(x,r), ...): a bijection for any grid, and the n-tiles of a panel become L2 neighbours in time.
    int bx = blockIdx.x;
    long bz = blockIdx.z;
    {
        const int total = gridDim.x * gridDim.z;
        if (total >= 64) {
            const int L = blockIdx.z * gridDim.x + blockIdx.x;
            const int x = L & 7, q = total >> 3, r = total & 7;
            const int logical = x * q + (x < r ? x : r) + (L >> 3);
            bz = logical / gridDim.x;
            bx = logical % gridDim.x;
        }
    }
    const int tile_m = bx / g.tiles_n, tile_n = bx % g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (g.dyn) {                        // wave-uniform: sizes decided on the device (deflation count)
        const int kk = g.dyn[bz];
        g.N = kk;
        g.K = kk;
        if (n0 >= kk) return;
    }
    const double *__restrict__ A = g.A + bz * g.sA;
    const double *__restrict__ B = g.B + bz * g.sB;

    d4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    double ra[TileA::PER_THREAD], rb[TileB::PER_THREAD];
    const double *pa[TileA::PER_THREAD], *pb[TileB::PER_THREAD];
    TileA::setup(pa, A, g.lda, m0, g.M, tid);
    TileB::setup(pb, B, g.ldb, n0, g.N, tid);
    const long stepA = TileA::step(g.lda), stepB = TileB::step(g.ldb);
    const int nk = (g.K + BK - 1) / BK;
    const int nfull = g.K / BK;                  // tiles [0, nfull) are complete

    if (nfull > 0) {
        TileA::gload(ra, pa, 0);
        TileB::gload(rb, pb, 0);
        TileA::template sstore<false>(ra, ldsA(0), 0, g.K, tid);
        TileB::template sstore<false>(rb, ldsB(0), 0, g.K, tid);
    } else {
        TileA::gload_tail(ra, pa, g.lda, 0, g.K, tid);
        TileB::gload_tail(rb, pb, g.ldb, 0, g.K, tid);
        TileA::template sstore<true>(ra, ldsA(0), 0, g.K, tid);
        TileB::template sstore<true>(rb, ldsB(0), 0, g.K, tid);
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    int cur = 0;
    auto mma_tile = [&](int buf) {
        const double *sa = ldsA(buf);
        const double *sb = ldsB(buf);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int k = kk * 4 + fq;
            double a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = sa[TileA::lds_index(wr * 16 * FM + i * 16 + fr, k)];
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = sb[TileB::lds_index(wc * 16 * FN + j * 16 + fr, k)];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // steady state: the next tile is complete -> straight-line loads, MFMAs, LDS stores (no selects, no branches)
    int kt = 0;
    for (; kt + 1 < nfull; ++kt) {
        TileA::gload(ra, pa, (kt + 1) * stepA);
        TileB::gload(rb, pb, (kt + 1) * stepB);
        mma_tile(cur);
        TileA::template sstore<false>(ra, ldsA(cur ^ 1), 0, g.K, tid);
        TileB::template sstore<false>(rb, ldsB(cur ^ 1), 0, g.K, tid);
        __syncthreads();
        cur ^= 1;
    }
    if (kt + 1 < nk) {                         // one partial K tile follows
        TileA::gload_tail(ra, pa, g.lda, (kt + 1) * BK, g.K, tid);
        TileB::gload_tail(rb, pb, g.ldb, (kt + 1) * BK, g.K, tid);
        mma_tile(cur);
        TileA::template sstore<true>(ra, ldsA(cur ^ 1), (kt + 1) * BK, g.K, tid);
        TileB::template sstore<true>(rb, ldsB(cur ^ 1), (kt + 1) * BK, g.K, tid);
        __syncthreads();
        cur ^= 1;
    }
    mma_tile(cur);

    // ---- epilogue ----
    double qsum = 0.0, qsum2 = 0.0;
    double *__restrict__ C = (EPI == EPI_QUAD) ? nullptr : g.C + bz * g.sC;
    double *__restrict__ C2 = (EPI == EPI_DUAL || EPI == EPI_GRAD) ? g.C2 + bz * g.sC : nullptr;
    double *__restrict__ C3 = (EPI == EPI_GRAD) ? g.C3 + bz * g.sC : nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wr * 16 * FM + i * 16 + fq + 4 * r;
            if (row >= g.M) continue;
            long drow = 0;
            if (EPI == EPI_DIV_D || EPI == EPI_QUAD || EPI == EPI_GRAD) drow = (long)(row / g.rdiv) * g.ldd;
            const double rsc = (EPI == EPI_GRAD) ? g.rowscale[row / g.rdiv] : 0.0;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int col = n0 + wc * 16 * FN + j * 16 + fr;
                if (col >= g.N) continue;
                const double v = acc[i][j][r];
                if (EPI == EPI_STORE) {
                    C[(long)row * g.ldc + col] = g.alpha * v;
                } else if (EPI == EPI_ACCUM) {
                    C[(long)row * g.ldc + col] += g.alpha * v;
                } else if (EPI == EPI_DUAL) {
                    const double o = g.alpha * v;
                    C[(long)row * g.ldc + col] = o;
                    C2[(long)row * g.ldc + col] += o;
                } else if (EPI == EPI_DIV_D) {
                    C[(long)row * g.ldc + col] = v / g.D[drow + col];
                } else if (EPI == EPI_GRAD) {
                    // b = alpha / D; also b * et[col], b * es[row / rdiv]; partial sums of alpha*b and b*b
                    const double bq = v / g.D[drow + col];
                    C[(long)row * g.ldc + col] = bq;
                    C2[(long)row * g.ldc + col] = bq * g.colscale[col];
                    C3[(long)row * g.ldc + col] = bq * rsc;
                    qsum += v * bq;
                    qsum2 += bq * bq;
                } else {
                    qsum += v * v / g.D[drow + col];
                }
            }
        }
    }
    if (EPI == EPI_QUAD || EPI == EPI_GRAD) {
        // wave reduction (64 lanes) then the waves through LDS; fixed order -> deterministic
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            qsum += __shfl_down(qsum, off, 64);
            if (EPI == EPI_GRAD) qsum2 += __shfl_down(qsum2, off, 64);
        }
        __shared__ double wsum[2 * WM * WN];
        if (lane == 0) {
            wsum[wid] = qsum;
            wsum[WM * WN + wid] = qsum2;
        }
        __syncthreads();
        if (tid == 0) {
            double t = 0.0, t2 = 0.0;
#pragma unroll
            for (int i = 0; i < WM * WN; ++i) {
                t += wsum[i];
                t2 += wsum[WM * WN + i];
            }
            const long nb = (long)gridDim.x * gridDim.z, me = bz * gridDim.x + bx;
            g.partials[me] = t;
            if (EPI == EPI_GRAD) g.partials[nb + me] = t2;
        }
    }
}

// Deterministic final reduction of per-block partials (single workgroup, fixed tree).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ p, long n, double *out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

template <int WM, int WN, int FM, int FN, int BK, bool TA, bool TB>
static void launch_epi(const GemmK &k, int epi, dim3 grid, hipStream_t s) {
    const dim3 blk(64 * WM * WN);
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_STORE>), grid, blk, 0, s, k); break;
        case EPI_DIV_D: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_DIV_D>), grid, blk, 0, s, k); break;
        case EPI_QUAD: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_QUAD>), grid, blk, 0, s, k); break;
        case EPI_ACCUM: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_ACCUM>), grid, blk, 0, s, k); break;
        case EPI_DUAL: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_DUAL>), grid, blk, 0, s, k); break;
        case EPI_GRAD: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_GRAD>), grid, blk, 0, s, k); break;
        default: throw HipError{-3, "gemm_f64: bad epilogue"};
    }
}

template <int WM, int WN, int FM, int FN, int BK>
static void launch_trans(const GemmK &k, bool ta, bool tb, int epi, dim3 grid, hipStream_t s) {
    if (!ta && !tb) launch_epi<WM, WN, FM, FN, BK, false, false>(k, epi, grid, s);
    else if (ta && !tb) launch_epi<WM, WN, FM, FN, BK, true, false>(k, epi, grid, s);
    else if (!ta && tb) launch_epi<WM, WN, FM, FN, BK, false, true>(k, epi, grid, s);
    else launch_epi<WM, WN, FM, FN, BK, true, true>(k, epi, grid, s);
}

void gemm_f64(gpcsd_ctx *c, const GemmDesc &g, hipStream_t s) {
    if (!s) s = c->stream;
    GP_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, -3, "gemm_f64: empty problem %dx%dx%d", g.M, g.N, g.K);
    GemmK k;
    k.M = g.M; k.N = g.N; k.K = g.K;
    k.A = g.A; k.lda = g.lda; k.B = g.B; k.ldb = g.ldb; k.C = g.C; k.ldc = g.ldc; k.C2 = g.C2; k.C3 = g.C3;
    k.colscale = g.colscale; k.rowscale = g.rowscale;
    k.sA = g.sA; k.sB = g.sB; k.sC = g.sC;
    k.alpha = g.alpha; k.D = g.D; k.rdiv = g.rdiv > 0 ? g.rdiv : 1; k.ldd = g.ldd;
    k.partials = nullptr;
    k.dyn = g.dyn;

    // Tile configurations (block tile, waves, K depth).  Large flat GEMMs want many resident workgroups per CU so the
    // hardware dispatcher balances the tail; tiny GEMMs are latency-bound and want deep K tiles.
    //   1: 128x128, 8 waves, BK16   2: 128x64, 4 waves, BK8    3: 64x64, 4 waves, BK16
    //   4: 64x64, 4 waves, BK32     5: 32x32, 4 waves, BK64    6: 128x64, 4 waves, BK16
    static const int CFG_BM[7] = {0, 128, 128, 64, 64, 32, 128}, CFG_BN[7] = {0, 128, 64, 64, 64, 32, 64};
    int cfg = g.cfg;
    if (cfg <= 0 || cfg > 6) {
        auto tiles = [&](int bm, int bn) { return (long)ceil_div(g.M, bm) * ceil_div(g.N, bn) * g.batch; };
        // measured on MI355X (tools/gemm_sweep.py): 64x64 tiles reach the same ~40 TF/s as 128x128 on the large
        // flat GEMMs and balance the tail better; everything smaller is latency-bound and wants 32x32 / BK64
        cfg = (tiles(64, 64) >= 512) ? 3 : 5;
    }
    const int bm = CFG_BM[cfg], bn = CFG_BN[cfg];
    const int tm = ceil_div(g.M, bm), tn = ceil_div(g.N, bn);
    k.tiles_n = tn;
    dim3 grid(tm * tn, 1, g.batch);
    const long nblocks = (long)tm * tn * g.batch;
    if (g.epi == EPI_QUAD || g.epi == EPI_GRAD) k.partials = c->buf<double>("gemm_partials", 2 * nblocks);

    const double flops = 2.0 * g.M * (double)g.N * g.K * g.batch;
    {
        ProfScope ps(c, g.prof_name, flops, s);
        switch (cfg) {
            case 1: launch_trans<4, 2, 2, 4, 16>(k, g.transA, g.transB, g.epi, grid, s); break;
            case 2: launch_trans<2, 2, 4, 2, 8>(k, g.transA, g.transB, g.epi, grid, s); break;
            case 3: launch_trans<2, 2, 2, 2, 16>(k, g.transA, g.transB, g.epi, grid, s); break;
            case 4: launch_trans<2, 2, 2, 2, 32>(k, g.transA, g.transB, g.epi, grid, s); break;
            case 5: launch_trans<2, 2, 1, 1, 64>(k, g.transA, g.transB, g.epi, grid, s); break;
            default: launch_trans<2, 2, 4, 2, 16>(k, g.transA, g.transB, g.epi, grid, s); break;
        }
        GP_HIP(hipGetLastError());
    }
    if (g.epi == EPI_QUAD || g.epi == EPI_GRAD) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, (const double *)k.partials, nblocks, g.quad_out);
        if (g.epi == EPI_GRAD)
            hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, (const double *)(k.partials + nblocks), nblocks,
                               g.quad_out + 1);
        GP_HIP(hipGetLastError());
    }
}

}  // namespace gpcsd
