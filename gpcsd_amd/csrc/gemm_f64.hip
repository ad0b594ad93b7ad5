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
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

namespace gpcsd {

typedef double d4 __attribute__((ext_vector_type(4)));
// Fragment reads as volatile LDS loads: the compiler otherwise fuses neighbouring reads into ds_read2_b64, which is banked
// modulo 32 (two-way conflicts on these layouts, 25 % of the LDS cycles) instead of ds_read_b64's 64 banks.
typedef const volatile double __attribute__((address_space(3))) *lds_frag_ptr;
#define LDS_FRAG(p) (*(lds_frag_ptr)(p))

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
    long sColscale;
    long sA, sB, sC;
    double alpha;
    const double *D;
    int rdiv;
    long ldd, sD;
    double *partials;
    int tiles_n, tiles_m;
    int m_fastest;      // logical tile order: 1 = tile_m varies fastest (few row tiles, many column tiles)
    int n_group;        // otherwise: n-tiles per sweep over the row panels
    const int *dyn;     // optional device scalar: effective N and K (= *dyn) of this launch (D&C merge GEMMs)
    int lower;          // 1: tiles strictly above the diagonal of C are skipped (symmetric rank-k update, lower triangle wanted)
    int lower_shift;    // rows of the update in front of this launch's first row
    int prio;           // > 0: the launch's waves raise their issue priority (a small product on a dependent chain beside a flood of tiles)
    const double *kscale;   // optional: A's element at contracted index k is multiplied by kscale[k] on its way to LDS
    long sKscale, sKscale2;
    // outer batch level (GemmDesc::batch2): grid z = z2 * batch1 + z1
    int batch1;
    long sA2, sB2, sC2, sD2, sColscale2, sRowscale2, sDyn2;
};

// One operand tile: BO "outer" rows/cols (M or N side) x BK, staged by NT threads.  KMAJOR: global storage is [K][O].
// Each thread owns PER_THREAD fixed (o, k) slots of the tile.
//
// On MI355X the fp64 MFMA runs at the plain fp64 vector rate (78.6 TF/s both), i.e. it occupies the SIMD's vector ALU
// for its 64 cycles, and every other VALU instruction of ANY wave on that SIMD is time taken from the MFMAs (measured:
// 3.2 VALU per MFMA in the old loop = 72 % MFMA duty).  So the steady-state loop is written to need no VALU at all:
//   * global loads use a wave-uniform 64-bit tile base (advanced on the scalar ALU) + loop-invariant 32-bit per-thread
//     byte offsets (outer index clamped into range once: a duplicated edge row only feeds accumulators the epilogue
//     never stores), i.e. `global_load_dwordx2 v, v_off, s[base]` with nothing to compute per tile;
//   * LDS addresses are one per-thread base + compile-time immediates (the double buffer index is a template constant);
//   * loaded registers are first touched by the LDS store after the MFMA block, so the loads overlap the MFMAs.
// Only the last, partial K tile takes a clamped + zero-masked path.
template <int BO, bool KMAJOR, int NT, int BK>
struct Tile {
    static constexpr int LD_OK = BK + 2;       // (BK+2) mod 32 == 2 for BK = 16, 64: rows land on distinct even slots
    static constexpr int LDS_ELEMS = KMAJOR ? BK * (BO + PAD_KO) : BO * LD_OK;
    static constexpr int PER_THREAD = BO * BK / NT;
    static_assert(BO * BK % NT == 0, "tile must divide evenly over the workgroup");
    static_assert(NT % BO == 0 && NT % BK == 0, "slot i of a thread must sit a constant distance from its slot 0");

    __device__ static __forceinline__ constexpr int lds_index(int o, int k) {
        return KMAJOR ? k * (BO + PAD_KO) + o : o * LD_OK + k;
    }
    // slot 0 of thread tid; slot i adds (DO * i, DK * i)
    static constexpr int DO = KMAJOR ? 0 : NT / BK, DK = KMAJOR ? NT / BO : 0;
    __device__ static __forceinline__ void slot0(int tid, int &o, int &k) {
        if (KMAJOR) {
            k = tid / BO;
            o = tid % BO;
        } else {
            o = tid / BK;
            k = tid % BK;
        }
    }
    // wave-uniform base of the tile's first outer row/column, and the buffer resource of K tile t over it (raw buffer, no
    // bounds).  The resource is re-based per K tile with 64-bit scalar arithmetic, so the only 32-bit quantities are the
    // per-thread offsets INSIDE one K tile -- an operand may be as large as memory (a K-major operand of ld = ntrials * nt
    // was limited to 2^28 elements while the K offset travelled in the 32-bit soffset).
    __device__ static __forceinline__ const double *tile_base(const double *p, long ld, int o0) {
        return p + (KMAJOR ? (long)o0 : (long)o0 * ld);
    }
    __device__ static __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const double *tile0, long ld, int t) {
        const double *b = tile0 + (long)t * (KMAJOR ? (long)BK * ld : (long)BK);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(b), 0, 0xFFFFFFFF, 0x00020000);
    }
    // byte offsets of this thread's slots from the tile base (host side checks they fit 32 bits)
    __device__ static __forceinline__ void setup(unsigned (&off)[PER_THREAD], long ld, int o0, int Olim, int tid) {
        int o, k;
        slot0(tid, o, k);
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int go = o0 + o + DO * i, rel = (go < Olim ? go : Olim - 1) - o0;
            off[i] = (unsigned)((KMAJOR ? (long)(k + DK * i) * ld + rel : (long)rel * ld + k) * 8);
        }
    }
    __device__ static __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
    // full tile at scalar byte offset soff: buffer_load_dwordx2 v, voff, s[rsrc], soff offen -- no VALU at all
    __device__ static __forceinline__ void gload(double (&r)[PER_THREAD], __amdgpu_buffer_rsrc_t rs, int soff,
                                                 const unsigned (&off)[PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) r[i] = bload(rs, off[i], soff);
    }
    // partial tile with kleft (>= 1) valid k: out-of-range k reads the last valid one instead (always a legal address)
    __device__ static __forceinline__ void gload_tail(double (&r)[PER_THREAD], __amdgpu_buffer_rsrc_t rs, int soff, long ld, int o0,
                                                      int Olim, int kleft, int tid) {
        int o, k;
        slot0(tid, o, k);
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int go = o0 + o + DO * i, rel = (go < Olim ? go : Olim - 1) - o0;
            const int kk = k + DK * i, kc = kk < kleft ? kk : kleft - 1;
            r[i] = bload(rs, (unsigned)((KMAJOR ? (long)kc * ld + rel : (long)rel * ld + kc) * 8), soff);
        }
    }
    // thr = this thread's slot-0 address inside the destination buffer
    template <bool MASK>
    __device__ static __forceinline__ void sstore(const double (&r)[PER_THREAD], double *thr, int k_slot0, int kleft) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i)
            thr[lds_index(DO * i, DK * i)] = (!MASK || k_slot0 + DK * i < kleft) ? r[i] : 0.0;
    }
};

template <int WM, int WN, int FM, int FN, int BK, bool TA, bool TB, int EPI, bool KS = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f64_kernel(GemmK g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 16 * FM * WM, BN = 16 * FN * WN;
    using TileA = Tile<BM, TA, NT, BK>;    // transA: global [K][M]
    using TileB = Tile<BN, !TB, NT, BK>;   // !transB: global [K][N]
    __shared__ double lds[2 * (TileA::LDS_ELEMS + TileB::LDS_ELEMS)];
    // operand A buffers at lds + b*TileA::LDS_ELEMS; operand B buffers follow the A buffers
    if (g.prio) __builtin_amdgcn_s_setprio(3);
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
        if (total >= 64 && !g.lower) {
            const int L = blockIdx.z * gridDim.x + blockIdx.x;
            const int x = L & 7, q = total >> 3, r = total & 7;
            const int logical = x * q + (x < r ? x : r) + (L >> 3);
            bz = logical / gridDim.x;
            bx = logical % gridDim.x;
        } else if (total >= 64) {
            // lower-triangular update: the tiles above the diagonal exit at once, and they are not spread evenly over the logical
            // order (the first n-groups run nearly all their tiles, the last ones nearly none) -- one contiguous range per XCD
            // left XCD 0 with twice the average work and XCD 7 with none (31 TF/s).  Chunks of one row panel's n-group (the
            // tiles that share an A panel) are dealt round-robin to the XCDs instead.
            const int L = blockIdx.x;                        // (within one batch entry: blockIdx.z stays)
            const int x = L & 7, slot = L >> 3, ch = g.n_group;
            const int full = ((int)gridDim.x >> 3) / ch * ch * 8;     // tiles covered by whole rounds of 8 chunks
            if (L < full) bx = ((slot / ch) * 8 + x) * ch + slot % ch;
        }
    }
    // consecutive logical tiles share the operand panel of the LONGER tile dimension, so that panel is fetched from HBM
    // once per XCD neighbourhood while the short dimension's operand stays L2 resident as a whole
    int tile_m, tile_n;
    if (g.m_fastest) {
        tile_m = bx % g.tiles_m;
        tile_n = bx / g.tiles_m;
    } else {
        // n-tiles in groups of n_group: one sweep over all row panels per group, so the group's slice of B (<= ~2 MB)
        // stays in the 4 MB L2 for the whole sweep instead of being evicted by the streaming A panels
        const int per_group = g.tiles_m * g.n_group;
        const int gi = bx / per_group, rem = bx - gi * per_group;
        const int n_first = gi * g.n_group;
        const int ng = (g.tiles_n - n_first < g.n_group) ? g.tiles_n - n_first : g.n_group;
        tile_m = rem / ng;
        tile_n = n_first + rem - tile_m * ng;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (g.lower && n0 > m0 + BM - 1 + g.lower_shift) return;      // wave-uniform: no entry of this tile lies on or below the diagonal
    // two batch levels: z1 the inner entry (stride s?), z2 the outer one (stride s?2; replicas of a hyper-parameter batch)
    long z1 = bz, z2 = 0;
    if (g.batch1 > 0) {                 // wave-uniform; only launches with an outer level pay the division
        z2 = bz / g.batch1;
        z1 = bz - z2 * g.batch1;
    }
    if (g.dyn) {                        // wave-uniform: sizes decided on the device (deflation count)
        const int kk = g.dyn[z2 * g.sDyn2 + z1];
        g.N = kk;
        g.K = kk;
        if (n0 >= kk) return;
    }
    const double *__restrict__ A = g.A + z1 * g.sA + z2 * g.sA2;
    const double *__restrict__ B = g.B + z1 * g.sB + z2 * g.sB2;

    d4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    if (EPI == EPI_SUB) {                 // accumulators start as -C: issued in front of the first operand tile's loads
        const double *__restrict__ Cin = g.C + z1 * g.sC + z2 * g.sC2;
        const int colb = n0 + wc * 16 * FN + (lane & 15), rowb = m0 + wr * 16 * FM + (lane >> 4);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + 16 * i + 4 * r;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int col = colb + 16 * j;
                    if (row < g.M && col < g.N) acc[i][j][r] = -Cin[(long)row * g.ldc + col];
                }
            }
    }

    double ra[TileA::PER_THREAD], rb[TileB::PER_THREAD];
    unsigned offA[TileA::PER_THREAD], offB[TileB::PER_THREAD];
    TileA::setup(offA, g.lda, m0, g.M, tid);
    TileB::setup(offB, g.ldb, n0, g.N, tid);
    const double *const baseA = TileA::tile_base(A, g.lda, m0), *const baseB = TileB::tile_base(B, g.ldb, n0);   // wave-uniform
    const int nk = (g.K + BK - 1) / BK;
    const int nfull = g.K / BK;                  // tiles [0, nfull) are complete

    const int fr = lane & 15, fq = lane >> 4;
    int oa0, ka0, ob0, kb0;
    TileA::slot0(tid, oa0, ka0);
    TileB::slot0(tid, ob0, kb0);
    double *const swA = lds + TileA::lds_index(oa0, ka0);                                   // LDS store bases (buffer 0)
    double *const swB = lds + 2 * TileA::LDS_ELEMS + TileB::lds_index(ob0, kb0);
    const double *const srA = lds + TileA::lds_index(wr * 16 * FM + fr, fq);                // fragment read bases (buffer 0)
    const double *const srB = lds + 2 * TileA::LDS_ELEMS + TileB::lds_index(wc * 16 * FN + fr, fq);
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;

    // optional scaling of A along K (GemmDesc::kscale; instantiations with KS only -- as a run-time branch in every
    // instantiation it cost the cfg3 step 1.6 %): the factors of a thread's slots travel with the tile's loads and are applied
    // right before the LDS store, i.e. behind the MFMA block like the store itself
    const double *__restrict__ ksc = KS ? g.kscale + z1 * g.sKscale + z2 * g.sKscale2 : nullptr;
    double rk[KS ? TileA::PER_THREAD : 1];
    auto load_ks = [&](int t) {
        if constexpr (KS) {
#pragma unroll
            for (int i = 0; i < TileA::PER_THREAD; ++i) {
                const int kk = t * BK + ka0 + TileA::DK * i;
                rk[i] = ksc[kk < g.K ? kk : g.K - 1];
            }
        }
    };
    auto apply_ks = [&]() {
        if constexpr (KS) {
#pragma unroll
            for (int i = 0; i < TileA::PER_THREAD; ++i) ra[i] *= rk[i];
        }
    };
    auto load_full = [&](int t) {
        TileA::gload(ra, TileA::rsrc(baseA, g.lda, t), 0, offA);
        TileB::gload(rb, TileB::rsrc(baseB, g.ldb, t), 0, offB);
        load_ks(t);
    };
    auto load_any = [&](int t) {
        if (t < nfull) {
            load_full(t);
        } else {
            TileA::gload_tail(ra, TileA::rsrc(baseA, g.lda, t), 0, g.lda, m0, g.M, g.K - t * BK, tid);
            TileB::gload_tail(rb, TileB::rsrc(baseB, g.ldb, t), 0, g.ldb, n0, g.N, g.K - t * BK, tid);
            load_ks(t);
        }
    };
    auto store_full = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        apply_ks();
        TileA::template sstore<false>(ra, swA + buf * TileA::LDS_ELEMS, 0, 0);
        TileB::template sstore<false>(rb, swB + buf * TileB::LDS_ELEMS, 0, 0);
    };
    auto store_any = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        if (t < nfull) {
            store_full(bufc);
        } else {
            apply_ks();
            TileA::template sstore<true>(ra, swA + buf * TileA::LDS_ELEMS, ka0, g.K - t * BK);
            TileB::template sstore<true>(rb, swB + buf * TileB::LDS_ELEMS, kb0, g.K - t * BK);
        }
    };
    // k-steps (of 4) the LAST K tile needs: a partial tile is zero-filled in LDS, but its all-zero steps are skipped
    // (K = 500, BK = 16: one step instead of four, 2.3 % of the MFMAs of the whole GEMM)
    const int last_steps = (nk > nfull) ? (g.K - nfull * BK + 3) / 4 : BK / 4;
    auto mma_last = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const double *sa = srA + buf * TileA::LDS_ELEMS;
        const double *sb = srB + buf * TileB::LDS_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            if (kk < last_steps) {                  // wave-uniform
                double a[FM], b[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) a[i] = LDS_FRAG(sa + TileA::lds_index(i * 16, kk * 4));
#pragma unroll
                for (int j = 0; j < FN; ++j) b[j] = LDS_FRAG(sb + TileB::lds_index(j * 16, kk * 4));
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto mma_tile = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const double *sa = srA + buf * TileA::LDS_ELEMS;
        const double *sb = srB + buf * TileB::LDS_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = LDS_FRAG(sa + TileA::lds_index(i * 16, kk * 4));
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = LDS_FRAG(sb + TileB::lds_index(j * 16, kk * 4));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    load_any(0);
    store_any(Buf0{}, 0);
    __syncthreads();
    // steady state, two K tiles per trip so the buffer index is a compile-time constant
    int kt = 0;
    for (; kt + 2 < nfull; kt += 2) {
        load_full(kt + 1);
        __builtin_amdgcn_sched_barrier(0);     // keep the order loads | MFMAs | LDS stores: the MFMAs are the loads' cover
        mma_tile(Buf0{});
        __builtin_amdgcn_sched_barrier(0);
        store_full(Buf1{});
        __syncthreads();
        load_full(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(Buf1{});
        __builtin_amdgcn_sched_barrier(0);
        store_full(Buf0{});
        __syncthreads();
    }
    // tile kt sits in buffer 0 and at most two more follow (the last one possibly partial)
    if (kt + 1 < nk) {
        load_any(kt + 1);
        mma_tile(Buf0{});
        store_any(Buf1{}, kt + 1);
        __syncthreads();
        if (kt + 2 < nk) {
            load_any(kt + 2);
            mma_tile(Buf1{});
            store_any(Buf0{}, kt + 2);
            __syncthreads();
            mma_last(Buf0{});
        } else {
            mma_last(Buf1{});
        }
    } else {
        mma_last(Buf0{});
    }

    // ---- epilogue ----
    double qsum = 0.0, qsum2 = 0.0;
    const long offC = z1 * g.sC + z2 * g.sC2;
    double *C = (EPI == EPI_QUAD) ? nullptr : g.C + offC;
    const double *__restrict__ Dz = g.D + z1 * g.sD + z2 * g.sD2;
    double *__restrict__ C2 = (EPI == EPI_GRAD && g.C2) ? g.C2 + offC : nullptr;      // (optional: callers that scale at the
    double *__restrict__ C3 = (EPI == EPI_GRAD && g.C3) ? g.C3 + offC : nullptr;      //  consumer's load -- kscale -- skip the copies)
    const double *__restrict__ colscale = g.colscale ? g.colscale + z1 * g.sColscale + z2 * g.sColscale2 : nullptr;
    const double *__restrict__ rowscale = (EPI == EPI_GRAD) ? g.rowscale + z2 * g.sRowscale2 : nullptr;
    if (EPI == EPI_STORE || EPI == EPI_SUB) {
        // The plain store is the hot epilogue: every VALU instruction here is taken from the MFMAs of the workgroups that
        // share the SIMD, so the per-column factor (alpha, optional column scale) and the validity of the FN columns are
        // formed once, rows advance by pointer increments, and the FN stores of a row use immediate offsets.
        const int colb = n0 + wc * 16 * FN + fr;
        double asc[FN];
        bool cok[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            cok[j] = colb + 16 * j < g.N;
            asc[j] = g.alpha;
            if (colscale) asc[j] *= colscale[cok[j] ? colb + 16 * j : 0];       // wave-uniform test
        }
        const int rowb = m0 + wr * 16 * FM + fq;
        double *pr = C + (long)rowb * g.ldc + colb;
        const long step4 = 4 * g.ldc;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (rowb + 16 * i + 4 * r < g.M) {
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        if (cok[j]) pr[16 * j] = asc[j] * acc[i][j][r];
                }
                pr += step4;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wr * 16 * FM + i * 16 + fq + 4 * r;
            if (row >= g.M) continue;
            long drow = 0;
            if (EPI == EPI_DIV_D || EPI == EPI_QUAD || EPI == EPI_GRAD) drow = (long)(row / g.rdiv) * g.ldd;
            const double rsc = (EPI == EPI_GRAD) ? rowscale[row / g.rdiv] : 0.0;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int col = n0 + wc * 16 * FN + j * 16 + fr;
                if (col >= g.N) continue;
                const double v = acc[i][j][r];
                if (EPI == EPI_ACCUM) {
                    C[(long)row * g.ldc + col] += g.alpha * v;
                } else if (EPI == EPI_DIV_D) {
                    C[(long)row * g.ldc + col] = v * Dz[drow + col];
                } else if (EPI == EPI_GRAD) {
                    // b = alpha / D; also b * et[col], b * es[row / rdiv]; partial sums of alpha*b and b*b
                    const double bq = v * Dz[drow + col];
                    C[(long)row * g.ldc + col] = bq;
                    if (C2) C2[(long)row * g.ldc + col] = bq * colscale[col];        // wave-uniform
                    if (C3) C3[(long)row * g.ldc + col] = bq * rsc;
                    qsum += v * bq;
                    qsum2 += bq * bq;
                } else {
                    qsum += v * v * Dz[drow + col];
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

// ------------------------------------------------------------------------------------------------------------------
// Last product of a folded prediction with the unfold fused in (replaces gemm_pred_tstar + unfold_swap_sum_kernel).
//
//   comp~[(zq, r)][tp][cc][b] = sum_i' S~[(zq, r)][tp block i'] Pcat_tp[i'][cc * npP + b]        (the GEMM, K = nts / nta)
//   out[cc][z][t][r] = sum over the site parity sp of zq and the time parity tp of b of (+-) weights * comp~   (the unfold)
//
// Every output element combines FOUR entries of comp~ -- (site-symmetric, site-antisymmetric row) x (time-symmetric,
// time-antisymmetric column block) of one (site orbit a, trial r, time orbit b) -- and yields the four mirror images
// (z_i | z_j) x (t_k | t_l).  The kernel computes the transposed product, rows = (cc, b), columns = (a, r), so that the lanes
// of a wave run along r, the innermost index of the outputs: a workgroup owns 32 time orbits b (all CC components of them:
// fragment i of a wave = component i, so the sum over components is a register add) x 32 columns (a, r) in BOTH site parities
// (fragment j = 0 the symmetric row of S~, j = 1 the antisymmetric one), runs the K loop once per time parity with its own
// accumulators, and then holds ss / as / sa / aa of the same (b, a, r) in one lane: the unfold is register arithmetic and every
// store instruction writes 16 consecutive trials (128 bytes) of four output rows.  comp~ (154 MB at 384 x 500 x 50) is never
// written or read, and the relayout launch is gone.
template <int CC>
struct PredUnfoldK {
    const double *S;             // S~ [(zq, r)][nt] row-major
    long lds;
    const double *Pc[2];         // Pcat_tp [K_tp][C * npP_tp]
    long ldp[2];
    int npP[2], K[2], kcol0[2];
    int nb, nba;                 // time orbits; those below nba have an antisymmetric partner
    long ncolS, ncolA, anti_row0;   // columns (a, r) of the symmetric site block, of the antisymmetric one, its first row in S~
    int R, nt;
    SymDev sz, st;
    double *list;                // [cc][z][t][r] (nullptr: sums only)
    long list_stride;
    double *sum;                 // [z][t][r]
    int tiles_b;
    long tile_c0;                // first column tile of this launch (chunked launches: gpcsd_predict's copy pipeline)
};

// BF = fragments of 16 time orbits per wave (workgroup = 32 * BF orbits): BF = 2 halves the number of workgroups that stream
// one S~ panel and issues 8 MFMAs per 6 fragment reads instead of 4 per 4; every element's K order is the same in both.
template <int CC, int BF>
__global__ __launch_bounds__(256) void gemm_pred_unfold_kernel(PredUnfoldK<CC> g) {
    constexpr int NT = 256, BK = 16, BB = 32 * BF, BM = BB * CC, BN = 64;
    using TileA = Tile<BM, true, NT, BK>;     // Pcat: global [K][rows]
    using TileB = Tile<BN, false, NT, BK>;    // S~: global [cols][K]
    __shared__ double lds[2 * (TileA::LDS_ELEMS + TileB::LDS_ELEMS)];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    // XCD-aware order as in gemm_f64_kernel: consecutive logical tiles share the S~ panel (the long dimension)
    int L = blockIdx.x;
    {
        const int total = gridDim.x;
        if (total >= 64) {
            const int x = L & 7, q = total >> 3, r = total & 7;
            L = x * q + (x < r ? x : r) + (L >> 3);
        }
    }
    const int tile_b = L % g.tiles_b;
    const long tile_c = g.tile_c0 + L / g.tiles_b;
    const int b0 = tile_b * BB;
    const long n0 = tile_c * 32;
    const int fr = lane & 15, fq = lane >> 4;

    d4 acc[2][CC * BF][2];                                     // [time parity][component * BF + orbit fragment][site parity]
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int i = 0; i < CC * BF; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[tp][i][j] = d4{0.0, 0.0, 0.0, 0.0};

    int oa0, ka0, ob0, kb0;
    TileA::slot0(tid, oa0, ka0);
    TileB::slot0(tid, ob0, kb0);
    double *const swA = lds + TileA::lds_index(oa0, ka0);
    double *const swB = lds + 2 * TileA::LDS_ELEMS + TileB::lds_index(ob0, kb0);
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;

#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
        const int K = g.K[tp];
        if (K <= 0) continue;                                  // (no antisymmetric time block: wave-uniform)
        const int nbt = tp ? g.nba : g.nb;                     // valid b of this parity
        // per-thread slots: A' slot i = (row o = oa0 [+ 0], k = ka0 + DK i), row o -> component o / BB, time orbit b0 + o % BB
        unsigned offA[TileA::PER_THREAD], offB[TileB::PER_THREAD];
        int colA;
        {
            const int cc = oa0 / BB, bb = b0 + (oa0 % BB);
            colA = cc * g.npP[tp] + (bb < nbt ? bb : nbt - 1);
#pragma unroll
            for (int i = 0; i < TileA::PER_THREAD; ++i) offA[i] = (unsigned)(((long)(ka0 + TileA::DK * i) * g.ldp[tp] + colA) * 8);
        }
        long rowB[TileB::PER_THREAD];
#pragma unroll
        for (int i = 0; i < TileB::PER_THREAD; ++i) {          // slot 0: symmetric row n0 + o, slot 1 (o + 32): its antisymmetric row
            const int o = ob0 + TileB::DO * i;
            const long c = n0 + (o & 31);
            if (o < 32) rowB[i] = c < g.ncolS ? c : g.ncolS - 1;
            else rowB[i] = g.anti_row0 + (c < g.ncolA ? c : (g.ncolA > 0 ? g.ncolA - 1 : -g.anti_row0));
            offB[i] = (unsigned)((rowB[i] * g.lds + kb0) * 8);
        }
        const double *const baseA = g.Pc[tp];
        const double *const baseB = g.S + g.kcol0[tp];
        const int nk = (K + BK - 1) / BK, nfull = K / BK;
        double ra[TileA::PER_THREAD], rb[TileB::PER_THREAD];
        const double *const srA = lds + TileA::lds_index(wr * 16 + fr, fq);
        const double *const srB = lds + 2 * TileA::LDS_ELEMS + TileB::lds_index(wc * 16 + fr, fq);

        auto load_full = [&](int t) {
            TileA::gload(ra, TileA::rsrc(baseA, g.ldp[tp], t), 0, offA);
            TileB::gload(rb, TileB::rsrc(baseB, g.lds, t), 0, offB);
        };
        auto load_any = [&](int t) {
            if (t < nfull) {
                load_full(t);
            } else {                                           // partial K tile: out-of-range k reads the last valid one
                const int kleft = K - t * BK;
                const __amdgpu_buffer_rsrc_t rsa = TileA::rsrc(baseA, g.ldp[tp], t), rsb = TileB::rsrc(baseB, g.lds, t);
#pragma unroll
                for (int i = 0; i < TileA::PER_THREAD; ++i) {
                    const int kk = ka0 + TileA::DK * i, kc = kk < kleft ? kk : kleft - 1;
                    ra[i] = TileA::bload(rsa, (unsigned)(((long)kc * g.ldp[tp] + colA) * 8), 0);
                }
                const int kc = kb0 < kleft ? kb0 : kleft - 1;
#pragma unroll
                for (int i = 0; i < TileB::PER_THREAD; ++i) rb[i] = TileB::bload(rsb, (unsigned)((rowB[i] * g.lds + kc) * 8), 0);
            }
        };
        auto store_any = [&](auto bufc, int t) {
            constexpr int buf = decltype(bufc)::value;
            if (t < nfull) {
                TileA::template sstore<false>(ra, swA + buf * TileA::LDS_ELEMS, 0, 0);
                TileB::template sstore<false>(rb, swB + buf * TileB::LDS_ELEMS, 0, 0);
            } else {
                TileA::template sstore<true>(ra, swA + buf * TileA::LDS_ELEMS, ka0, K - t * BK);
                TileB::template sstore<true>(rb, swB + buf * TileB::LDS_ELEMS, kb0, K - t * BK);
            }
        };
        const int last_steps = (nk > nfull) ? (K - nfull * BK + 3) / 4 : BK / 4;
        auto mma = [&](auto bufc, int steps) {
            constexpr int buf = decltype(bufc)::value;
            const double *sa = srA + buf * TileA::LDS_ELEMS;
            const double *sb = srB + buf * TileB::LDS_ELEMS;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                if (kk < steps) {                              // wave-uniform
                    double a[CC * BF], b[2];
#pragma unroll
                    for (int i = 0; i < CC * BF; ++i) a[i] = LDS_FRAG(sa + TileA::lds_index(i * 32, kk * 4));
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = LDS_FRAG(sb + TileB::lds_index(j * 32, kk * 4));
#pragma unroll
                    for (int i = 0; i < CC * BF; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[tp][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[tp][i][j], 0, 0, 0);
                }
            }
        };
        __syncthreads();                                       // the previous parity's last fragment reads are done
        load_any(0);
        store_any(Buf0{}, 0);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 < nfull; kt += 2) {
            load_full(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(Buf0{}, BK / 4);
            __builtin_amdgcn_sched_barrier(0);
            store_any(Buf1{}, kt + 1);
            __syncthreads();
            load_full(kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(Buf1{}, BK / 4);
            __builtin_amdgcn_sched_barrier(0);
            store_any(Buf0{}, kt + 2);
            __syncthreads();
        }
        if (kt + 1 < nk) {
            load_any(kt + 1);
            mma(Buf0{}, BK / 4);
            store_any(Buf1{}, kt + 1);
            __syncthreads();
            if (kt + 2 < nk) {
                load_any(kt + 2);
                mma(Buf1{}, BK / 4);
                store_any(Buf0{}, kt + 2);
                __syncthreads();
                mma(Buf0{}, last_steps);
            } else {
                mma(Buf1{}, last_steps);
            }
        } else {
            mma(Buf0{}, last_steps);
        }
    }

    // ---- epilogue: unfold in site and time, sum over components, the (up to) four mirror images of every (b, a, r) ----
    const long rho = n0 + wc * 16 + fr;                        // this lane's column (a, r) of the symmetric site block
    if (rho >= g.ncolS) return;
    const int a = (int)(rho / g.R), rr = (int)(rho - (long)a * g.R);
    const int zi = g.sz.rep_i[a], zj = g.sz.rep_j[a];
    const double isq2 = 0.70710678118654752440;
    const double wz = (zi == zj) ? 1.0 : isq2;
    const bool has_za = (zi != zj) && rho < g.ncolA;
    const long rowlen = (long)g.nt * g.R;
    double *const sum_i = g.sum + (long)zi * rowlen + rr, *const sum_j = g.sum + (long)zj * rowlen + rr;
#pragma unroll
    for (int f = 0; f < BF; ++f)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int bb = b0 + f * 32 + wr * 16 + fq + 4 * r4;
        if (bb >= g.nb) continue;
        const int tk = g.st.rep_i[bb], tl = g.st.rep_j[bb];
        const double wt = (tk == tl) ? 1.0 : isq2;
        const bool has_ta = (tk != tl) && bb < g.nba;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const double ss = wz * wt * acc[0][cc * BF + f][0][r4];
            const double sa = has_ta ? wz * isq2 * acc[1][cc * BF + f][0][r4] : 0.0;
            const double as = has_za ? isq2 * wt * acc[0][cc * BF + f][1][r4] : 0.0;
            const double aa = (has_za && has_ta) ? 0.5 * acc[1][cc * BF + f][1][r4] : 0.0;
            const double v0 = (ss + sa) + (as + aa);           // (zi, tk)
            const double v1 = (ss - sa) + (as - aa);           // (zi, tl)
            const double v2 = (ss + sa) - (as + aa);           // (zj, tk)
            const double v3 = (ss - sa) - (as - aa);           // (zj, tl)
            s0 += v0;                                          // components summed in index order, as the reference does
            s1 += v1;
            s2 += v2;
            s3 += v3;
            if (g.list) {
                double *const li = g.list + (long)cc * g.list_stride + (long)zi * rowlen + rr;
                double *const lj = g.list + (long)cc * g.list_stride + (long)zj * rowlen + rr;
                li[(long)tk * g.R] = v0;
                if (tl != tk) li[(long)tl * g.R] = v1;
                if (zj != zi) {
                    lj[(long)tk * g.R] = v2;
                    if (tl != tk) lj[(long)tl * g.R] = v3;
                }
            }
        }
        sum_i[(long)tk * g.R] = s0;
        if (tl != tk) sum_i[(long)tl * g.R] = s1;
        if (zj != zi) {
            sum_j[(long)tk * g.R] = s2;
            if (tl != tk) sum_j[(long)tl * g.R] = s3;
        }
    }
}

bool gemm_pred_unfold_supported(int C, long nrows_S, int nt) {
    static const bool off = getenv("GPCSD_FUSED_UNFOLD") && getenv("GPCSD_FUSED_UNFOLD")[0] == '0';      // A/B switch
    // 32-bit byte offsets span all rows of S~ (symmetric and antisymmetric halves are addressed from one base)
    return !off && (C == 1 || C == 2) && nrows_S * (long)nt * 8 < (1L << 32);
}

void gemm_pred_unfold(gpcsd_ctx *c, const PredUnfoldDesc &d, hipStream_t s) {
    GP_REQUIRE(gemm_pred_unfold_supported(d.C, d.anti_row0 + d.ncolA, d.nt), -3, "gemm_pred_unfold: unsupported shape");
    // GPCSD_UNFOLD_BF=2: 64 time orbits per workgroup.  Measured at 384 x 500 x 50 (round 6): S~ is fetched 2.0 instead of
    // 2.8 times (the panels in flight on one XCD plus Pcat are about the size of its L2 either way), 80 MB less traffic per
    // step, but one wave per SIMD and 234 instead of 205 us per launch; the step is the same.  Default stays 32.
    const char *const bf_env = getenv("GPCSD_UNFOLD_BF");      // read per call: tests switch it
    const bool wide = bf_env && atoi(bf_env) == 2;
    const int tiles_b = ceil_div(d.nb, wide ? 64 : 32);
    const long tiles_all = (d.ncolS + 31) / 32;
    const long tc0 = d.tile_c1 < 0 ? 0 : d.tile_c0, tc1 = d.tile_c1 < 0 ? tiles_all : std::min(d.tile_c1, tiles_all);
    if (tc1 <= tc0) return;
    const long tiles_c = tc1 - tc0;
    const long nblocks = tiles_b * tiles_c;
    GP_REQUIRE(nblocks < (1L << 31), GPCSD_ERR_CAPACITY, "gemm_pred_unfold: too many tiles");
    const double flops = 2.0 * (double)(d.ncolS + d.ncolA) * d.C * ((double)d.nb * d.K[0] + (double)d.nba * d.K[1]) *
                         ((double)tiles_c / (double)tiles_all);
    ProfScope ps(c, "gemm_pred_tstar_unfold", flops, s);
    auto fill = [&](auto &k) {
        k.S = d.S; k.lds = d.lds;
        for (int tp = 0; tp < 2; ++tp) {
            k.Pc[tp] = d.Pc[tp]; k.ldp[tp] = d.ldp[tp]; k.npP[tp] = d.npP[tp]; k.K[tp] = d.K[tp]; k.kcol0[tp] = d.kcol0[tp];
        }
        k.nb = d.nb; k.nba = d.nba; k.ncolS = d.ncolS; k.ncolA = d.ncolA; k.anti_row0 = d.anti_row0;
        k.R = d.R; k.nt = d.nt; k.sz = d.sz; k.st = d.st; k.list = d.list; k.list_stride = d.list_stride; k.sum = d.sum;
        k.tiles_b = tiles_b;
        k.tile_c0 = tc0;
    };
    if (d.C == 1) {
        PredUnfoldK<1> k;
        fill(k);
        if (wide) hipLaunchKernelGGL((gemm_pred_unfold_kernel<1, 2>), dim3((unsigned)nblocks), dim3(256), 0, s, k);
        else hipLaunchKernelGGL((gemm_pred_unfold_kernel<1, 1>), dim3((unsigned)nblocks), dim3(256), 0, s, k);
    } else {
        PredUnfoldK<2> k;
        fill(k);
        if (wide) hipLaunchKernelGGL((gemm_pred_unfold_kernel<2, 2>), dim3((unsigned)nblocks), dim3(256), 0, s, k);
        else hipLaunchKernelGGL((gemm_pred_unfold_kernel<2, 1>), dim3((unsigned)nblocks), dim3(256), 0, s, k);
    }
    GP_HIP(hipGetLastError());
}

// Deterministic final reduction of per-block partials (single workgroup, fixed tree).
// A second workgroup may carry an unrelated reduction of the same shape (p2, n2 -> out2: the sum of log D partials of the
// likelihood, which would otherwise be a launch of its own in the dependent tail of the call).
// blockIdx.y = segment of an outer batch level: partials [y * n, (y + 1) * n) -> out[y * ostride].
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ p, long n, double *out,
                                                              const double *__restrict__ p2 = nullptr, long n2 = 0,
                                                              double *out2 = nullptr, long ostride = 0) {
    if (blockIdx.x == 1) {
        if (blockIdx.y != 0) return;
        p = p2;
        n = n2;
        out = out2;
    } else {
        p += blockIdx.y * n;
        out += blockIdx.y * ostride;
    }
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
        case EPI_GRAD: hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_GRAD>), grid, blk, 0, s, k); break;
        case EPI_SUB:
            if constexpr (!TA && TB) {
                hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI_SUB>), grid, blk, 0, s, k);
                break;
            } else {
                throw HipError{-3, "gemm_f64: EPI_SUB is instantiated for A (M,K), B (N,K) only"};
            }
        default: throw HipError{-3, "gemm_f64: bad epilogue"};
    }
}

// operand A scaled along K (GemmDesc::kscale): plain-store epilogue, the two operand layouts the gradient sums use
template <int WM, int WN, int FM, int FN, int BK>
static void launch_kscale(const GemmK &k, bool ta, bool tb, int epi, dim3 grid, hipStream_t s) {
    const dim3 blk(64 * WM * WN);
    if (epi != EPI_STORE || ta == tb) throw HipError{-3, "gemm_f64: kscale is implemented for the plain-store epilogue with exactly one transposed operand"};
    if (ta) hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, true, false, EPI_STORE, true>), grid, blk, 0, s, k);
    else hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, FM, FN, BK, false, true, EPI_STORE, true>), grid, blk, 0, s, k);
}

template <int WM, int WN, int FM, int FN, int BK>
static void launch_trans(const GemmK &k, bool ta, bool tb, int epi, dim3 grid, hipStream_t s) {
    if (!ta && !tb) launch_epi<WM, WN, FM, FN, BK, false, false>(k, epi, grid, s);
    else if (ta && !tb) launch_epi<WM, WN, FM, FN, BK, true, false>(k, epi, grid, s);
    else if (!ta && tb) launch_epi<WM, WN, FM, FN, BK, false, true>(k, epi, grid, s);
    else launch_epi<WM, WN, FM, FN, BK, true, true>(k, epi, grid, s);
}

// the tile configuration gemm_f64 picks by itself for a plain product of this shape (GemmDesc::cfg == 0; see the rule below)
int gemm_auto_cfg(int M, int N, int K, int batch) {
    const long tiles = (long)ceil_div(M, 64) * ceil_div(N, 64) * std::max(batch, 1);
    return (tiles >= 512) ? ((K <= 320) ? 2 : 3) : 5;
}

void gemm_f64(gpcsd_ctx *c, const GemmDesc &g, hipStream_t s) {
    if (!s) s = c->stream;
    GP_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, -3, "gemm_f64: empty problem %dx%dx%d", g.M, g.N, g.K);
    // 32-bit byte offsets inside one batch entry's operand (buffer_load voffset + soffset)
    // 32-bit byte offsets exist only inside one K tile of one block tile: (rows of the tile) x (leading dimension).  The
    // widest tile has 128 outer rows / 64 K rows, so a leading dimension below 2^22 doubles is always safe; K-major operands
    // (the flat projections: ld = ntrials * nt) only span BK <= 64 rows per tile and get 2^23
    const long ld_lim_a = g.transA ? GPCSD_MAX_GEMM_LD_KMAJOR : (1L << 22), ld_lim_b = g.transB ? (1L << 22) : GPCSD_MAX_GEMM_LD_KMAJOR;
    GP_REQUIRE(g.lda < ld_lim_a && g.ldb < ld_lim_b, GPCSD_ERR_CAPACITY,
               "gemm_f64: leading dimension %ld exceeds the capacity of one flat GEMM operand row (%ld doubles; "
               "GPCSD_MAX_GEMM_LD_KMAJOR: ntrials * nt of a resident block of trials)",
               g.lda >= ld_lim_a ? g.lda : g.ldb, g.lda >= ld_lim_a ? ld_lim_a : ld_lim_b);
    GemmK k;
    k.M = g.M; k.N = g.N; k.K = g.K;
    k.A = g.A; k.lda = g.lda; k.B = g.B; k.ldb = g.ldb; k.C = g.C; k.ldc = g.ldc; k.C2 = g.C2; k.C3 = g.C3;
    k.colscale = g.colscale; k.rowscale = g.rowscale; k.sColscale = g.sColscale;
    k.sA = g.sA; k.sB = g.sB; k.sC = g.sC;
    k.alpha = g.epi == EPI_SUB ? -1.0 : g.alpha; k.D = g.D; k.rdiv = g.rdiv > 0 ? g.rdiv : 1; k.ldd = g.ldd; k.sD = g.sD;
    k.partials = nullptr;
    k.dyn = g.dyn;
    k.lower = g.lower ? 1 : 0;
    k.lower_shift = g.lower ? g.lower_shift : 0;
    k.prio = g.prio;
    {
        // products on the chains' streams (spatial Gram assembly, divide & conquer merges, the prediction's side products) are small
        // launches on a dependent chain beside the main stream's floods of tiles: their waves issue first (GPCSD_CHAIN_GEMM_PRIO=0: A/B)
        static const bool off = getenv("GPCSD_CHAIN_GEMM_PRIO") && getenv("GPCSD_CHAIN_GEMM_PRIO")[0] == '0';
        if (!off && k.prio == 0 && (s == c->stream2 || s == c->stream3 || s == c->stream4)) k.prio = 1;
    }
    k.kscale = g.kscale; k.sKscale = g.sKscale; k.sKscale2 = g.sKscale2;
    const int batch2 = g.batch2 > 1 ? g.batch2 : 1;
    k.batch1 = batch2 > 1 ? g.batch : 0;
    k.sA2 = g.sA2; k.sB2 = g.sB2; k.sC2 = g.sC2; k.sD2 = g.sD2; k.sColscale2 = g.sColscale2; k.sRowscale2 = g.sRowscale2;
    k.sDyn2 = g.sDyn2;

    // Tile configurations (block tile, waves, K depth).  Large flat GEMMs want many resident workgroups per CU so the
    // hardware dispatcher balances the tail; tiny GEMMs are latency-bound and want deep K tiles.
    //   1: 128x128, 8 waves, BK16   2: 64x64, 4 waves, BK8   3: 64x64, 4 waves, BK16   5: 32x32, 4 waves, BK64
    // (128x64 / BK8 / BK32 / single-LDS-buffer variants were measured and dropped: tools/gemm_sweep.py)
    static const int CFG_BM[6] = {0, 128, 64, 64, 0, 32}, CFG_BN[6] = {0, 128, 64, 64, 0, 32};
    int cfg = g.cfg;
    if (cfg != 1 && cfg != 2 && cfg != 3 && cfg != 5) {
        // the outer batch level (hyper-parameter sets) does NOT enter the choice: a set must run the same tile configuration --
        // hence the same summation order of the fused reductions -- whether it is evaluated alone or in a batch
        auto tiles = [&](int bm, int bn) { return (long)ceil_div(g.M, bm) * ceil_div(g.N, bn) * g.batch; };
        // measured on MI355X (tools/gemm_sweep.py): 64x64 tiles reach the same ~40 TF/s as 128x128 on the large
        // flat GEMMs and balance the tail better; everything smaller is latency-bound and wants 32x32 / BK64
        // short K (the folded GEMMs: 192 / 250): the same tile with BK 8 -- half the pipeline fill per tile and half the LDS,
        // 46-48 TF/s against 40-41 with BK 16 at 192x25000x192 / 19200x250x250; equal from K = 500 up
        cfg = (tiles(64, 64) >= 512) ? ((g.K <= 320) ? 2 : 3) : 5;
    }
    if (g.kscale && (cfg == 1 || cfg == 2)) cfg = 3;           // (the K-scaled variant exists for configurations 3 and 5)
    const int bm = CFG_BM[cfg], bn = CFG_BN[cfg];
    const int tm = ceil_div(g.M, bm), tn = ceil_div(g.N, bn);
    k.tiles_n = tn;
    k.tiles_m = tm;
    k.m_fastest = (tm < tn) ? 1 : 0;
    {
        const long bytes_per_ntile = (long)g.K * bn * 8;                      // one column panel of B
        long ngp = bytes_per_ntile > 0 ? (2L << 20) / bytes_per_ntile : tn;  // ~2 MB of B per sweep
        if ((g.sB != 0 && g.batch > 1) || (g.sB2 != 0 && batch2 > 1)) ngp = tn;   // B differs per batch entry: nothing to keep
        k.n_group = (int)std::max(1L, std::min<long>(tn, ngp));
    }
    dim3 grid(tm * tn, 1, g.batch * batch2);
    const long nblocks = (long)tm * tn * g.batch * batch2;
    if (g.epi == EPI_QUAD || g.epi == EPI_GRAD) k.partials = c->buf<double>("gemm_partials", 2 * nblocks);

    double flops = 2.0 * g.M * (double)g.N * g.K * g.batch * batch2;
    if (g.lower) {                                  // count the tiles that run (on or below the diagonal), whole tiles
        long run = 0;
        for (int i = 0; i < tm; ++i) run += std::min<long>(tn, ((long)i * bm + bm - 1 + g.lower_shift) / bn + 1);
        flops = 2.0 * (double)run * bm * bn * g.K * g.batch * batch2;
    }
    {
        ProfScope ps(c, g.prof_name, flops, s);
        if (k.kscale) {
            switch (cfg) {
                case 3: launch_kscale<2, 2, 2, 2, 16>(k, g.transA, g.transB, g.epi, grid, s); break;
                case 5: launch_kscale<2, 2, 1, 1, 64>(k, g.transA, g.transB, g.epi, grid, s); break;
                default: throw HipError{-3, "gemm_f64: kscale is implemented for tile configurations 3 and 5"};
            }
        } else switch (cfg) {
            case 1: launch_trans<4, 2, 2, 4, 16>(k, g.transA, g.transB, g.epi, grid, s); break;
            case 2: launch_trans<2, 2, 2, 2, 8>(k, g.transA, g.transB, g.epi, grid, s); break;
            case 3: launch_trans<2, 2, 2, 2, 16>(k, g.transA, g.transB, g.epi, grid, s); break;
            default: launch_trans<2, 2, 1, 1, 64>(k, g.transA, g.transB, g.epi, grid, s); break;
        }
        GP_HIP(hipGetLastError());
    }
    if (g.epi == EPI_QUAD || g.epi == EPI_GRAD) {
        // one sum per outer batch entry: the partials of entry z2 are contiguous (grid z = z2 * batch + z1)
        const long per = nblocks / batch2;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(g.extra_sum_out ? 2 : 1, batch2), dim3(256), 0, s, (const double *)k.partials,
                           per, g.quad_out, g.extra_sum_in, (long)g.extra_sum_n, g.extra_sum_out, g.sQuad2);
        if (g.epi == EPI_GRAD)
            hipLaunchKernelGGL(reduce_partials_kernel, dim3(1, batch2), dim3(256), 0, s, (const double *)(k.partials + nblocks), per,
                               g.quad_out + 1, (const double *)nullptr, 0L, (double *)nullptr, g.sQuad2);
        GP_HIP(hipGetLastError());
    }
}

}  // namespace gpcsd
