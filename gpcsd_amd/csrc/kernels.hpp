// Device-pointer launchers shared between translation units of libgpcsd_hip.so.
#pragma once
#include <functional>
#include "ctx.hpp"

namespace gpcsd {

// ---------------------------------------------------------------- fp64 MFMA GEMM core
enum Epi : int {
    EPI_STORE = 0,   // C = alpha * acc
    EPI_DIV_D = 1,   // C = acc * D[(row / rdiv) * ldd + col]   with D = the RECIPROCALS 1/D_xi (k_build_D's Dinv)
    EPI_QUAD = 2,    // no store; partial sums of acc^2 * D[...] (reciprocals; deterministic two-stage reduce)
    EPI_ACCUM = 3,   // C += alpha * acc
    EPI_GRAD = 4,    // b = acc * D (reciprocals): C = b, C2 = b*colscale[col], C3 = b*rowscale[row/rdiv]; sums of acc*b and b*b
    EPI_DUAL_INIT = 6, // C = alpha*acc and C2 = alpha*acc (first term of a running sum: no read, no zero fill)
    EPI_SUB = 7        // C = C - acc with C read at the START of the tile (accumulators initialised to -C, plain-store epilogue with
                       // alpha = -1): the rank-k trailing updates of the Cholesky -- EPI_ACCUM's read-modify-write sits at the end
                       // of a tile, where nothing hides its latency.  A (M,K) row-major, B stored (N,K) only.
};

struct GemmDesc {
    int M = 0, N = 0, K = 0;
    const double *A = nullptr;
    long lda = 0;
    bool transA = false;   // false: A is (M,K) row-major; true: A stored (K,M) row-major
    const double *B = nullptr;
    long ldb = 0;
    bool transB = false;   // false: B is (K,N) row-major; true: B stored (N,K) row-major
    double *C = nullptr;
    long ldc = 0;
    double *C2 = nullptr;  // EPI_GRAD
    double *C3 = nullptr;  // EPI_GRAD
    const double *colscale = nullptr, *rowscale = nullptr;   // EPI_GRAD; EPI_STORE: optional C = alpha*acc*colscale[col]
    long sColscale = 0;                                      // batch stride of colscale (EPI_STORE)
    // optional scaling of operand A along the contracted index: C = (A diag(kscale)) B, applied to A's tile on its way to LDS
    // (one rounded product per element: the same bits as a pre-scaled copy of A, without writing and re-reading that copy)
    const double *kscale = nullptr;
    long sKscale = 0, sKscale2 = 0;                          // batch strides
    int batch = 1;
    long sA = 0, sB = 0, sC = 0;
    // second (outer) batch level: entry z = z2 * batch + z1 adds z2 * s?2 to the operand bases.  Used to run the same
    // product for several hyper-parameter sets (replicas) whose buffers sit a fixed stride apart.
    int batch2 = 1;
    long sA2 = 0, sB2 = 0, sC2 = 0, sD2 = 0, sColscale2 = 0, sRowscale2 = 0, sDyn2 = 0;
    long sQuad2 = 0;              // EPI_QUAD / EPI_GRAD: quad_out of outer entry z2 is quad_out + z2 * sQuad2
    double alpha = 1.0;
    int epi = EPI_STORE;
    const double *D = nullptr;
    int rdiv = 1;
    long ldd = 0;
    long sD = 0;                  // batch stride of D
    double *quad_out = nullptr;   // EPI_QUAD: one double (EPI_GRAD: two), written by the final reduce
    const double *extra_sum_in = nullptr;   // EPI_QUAD: an unrelated vector of partials summed by the same reduce launch
    int extra_sum_n = 0;                     //   (sum_small_kernel's association: bit-identical to k_build_D's own sum)
    double *extra_sum_out = nullptr;
    const int *dyn = nullptr;     // device int per batch entry: effective N = K = dyn[batch] (tiles beyond it exit)
    bool lower = false;           // square symmetric update (C and its mirror image equal): tiles strictly above the diagonal exit at once
    int lower_shift = 0;          // ... of a launch that covers rows lower_shift.. of such an update (row i of C is row lower_shift + i)
    int prio = 0;                 // 1: the launch's waves run at raised issue priority (small products of a dependent chain that share
                                  // their CUs with another stream's flood of tiles: Cholesky's diagonal chain)
    int cfg = 0;                  // 0 = choose the tile configuration automatically, 1 / 2 / 3 / 5 = force (see gemm_f64.hip)
    const char *prof_name = "gemm_f64";
};

int gemm_auto_cfg(int M, int N, int K, int batch);      // the configuration gemm_f64 picks for a plain product when GemmDesc::cfg == 0
void gemm_f64(gpcsd_ctx *c, const GemmDesc &g, hipStream_t s = nullptr);

// Last product of a folded prediction with the unfold, the (r, t) -> (t, r) relayout and the sum over components fused into
// its epilogue (gemm_f64.hip: gemm_pred_unfold_kernel).  S~ [(zq, r)][nt]; Pcat_tp [K_tp][C * npP_tp], tp = time parity.
struct PredUnfoldDesc {
    const double *S;
    long lds;
    const double *Pc[2];
    long ldp[2];
    int npP[2], K[2], kcol0[2];
    int nb, nba;                    // time orbits (symmetric block); those with an antisymmetric partner
    long ncolS, ncolA, anti_row0;   // (site orbit, trial) columns of the symmetric / antisymmetric site block; first anti row of S~
    int R, nt, C;
    SymDev sz, st;
    double *list;
    long list_stride;
    double *sum;
    long tile_c0 = 0, tile_c1 = -1;     // column tiles (32 columns (site orbit, trial) each) of this launch; tile_c1 < 0: all
};
constexpr int PRED_UNFOLD_BN = 32;      // columns (site orbit, trial) per tile of the fused last product
bool gemm_pred_unfold_supported(int C, long nrows_S, int nt);
void gemm_pred_unfold(gpcsd_ctx *c, const PredUnfoldDesc &d, hipStream_t s);

// ---------------------------------------------------------------- batched hyper-parameter sets
// Device image of gpcsd_hparams for batched evaluations (gpcsd_loglik_grad_batch): one entry per hyper-parameter set.  The
// Gram builders / derivative kernels take an optional table: with `tab` they run once for all B sets (the set index is a grid
// dimension, scalars come from tab[set], outputs are `s_out` apart), without it they are the single-set kernels as before.
struct HpDev {
    double R, eps, ell_s[2];
    int ncomp;
    int kind[GPCSD_MAX_TEMPORAL];
    double ell_t[GPCSD_MAX_TEMPORAL], sigma2_t[GPCSD_MAX_TEMPORAL];
    double sig2n, jitter;
    __host__ __device__ double ell_of(int c) const { return ell_t[c]; }          // (the accessors TemporalSet has too: kernels
    __host__ __device__ double sigma2_of(int c) const { return sigma2_t[c]; }    //  templated on the parameter source)
};

// ---------------------------------------------------------------- elementwise / Gram builders (gram.hip)
void k_b_fwd_1d(gpcsd_ctx *c, const double *r, long n, double R, double *out, hipStream_t s);
void k_second_diff(gpcsd_ctx *c, const double *in, long n_outer, long n_axis, long n_inner, double edge, double *out, hipStream_t s);
void k_b_fwd_2d(gpcsd_ctx *c, const double *d1, const double *d2, const double *w, long n, double R, double eps,
                double *out, hipStream_t s);
// out(n,m) = sum_c sigma2_c k_c(t_i - tp_j); ncomp components fused (K1-K3 of SURVEY 2a)
void k_temporal_gram(gpcsd_ctx *c, int ncomp, const int *kind, const double *ell, const double *sigma2,
                     const double *t, int n, const double *tp, int m, double *out, hipStream_t s, const HpDev *tab = nullptr,
                     int B = 1, long s_out = 0);
// A(nx, G) = gl_w[g] * b_fwd_1d(gl_x[g] - x[i], R)
void k_fwd_weights_1d(gpcsd_ctx *c, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl, double R,
                      double *A, hipStream_t s, const HpDev *tab = nullptr, int B = 1, long s_out = 0);
// A(nx, G=ngl1*ngl2) = w1[g1] w2[g2] * b_fwd_2d(|gl - x|)
void k_fwd_weights_2d(gpcsd_ctx *c, const double *xy, int nx, const double *gx1, const double *gw1, int ngl1,
                      const double *gx2, const double *gw2, int ngl2, double R, double eps, double *A, hipStream_t s,
                      const HpDev *tab = nullptr, int B = 1, long s_out = 0);
// SE kernel between two 1D point sets: out(n,m) = exp(-0.5 ((a_i - b_j)/ell)^2)
void k_se_1d(gpcsd_ctx *c, const double *a, int n, const double *b, int m, double ell, double *out, hipStream_t s,
             const HpDev *tab = nullptr, int B = 1, long s_out = 0);
// anisotropic SE between two 2D point sets given as separate coordinate generators:
// point i of set A = (a1[i / na2], a2[i % na2]) if na2 > 0 (tensor grid) else (a[2i], a[2i+1]) (explicit list)
void k_se_2d(gpcsd_ctx *c, const double *a1, const double *a2, int na, int na2, const double *b1, const double *b2,
             int nb, int nb2, double ell1, double ell2, double *out, hipStream_t s, const HpDev *tab = nullptr, int B = 1,
             long s_out = 0);
// one axis factor of the tensor-grid SE kernel: out(n,n) = exp(-0.5 (a_i - a_j)^2 / ell^2)
void k_se_axis(gpcsd_ctx *c, const double *a, int n, double ell, double *out, hipStream_t s);
// ... and with its derivative w.r.t. the length scale, for the B sets of a batched evaluation (ell = tab[set].ell_s[axis])
void k_se_axis_tab(gpcsd_ctx *c, const double *a, int n, int axis, const HpDev *tab, int B, double *K, double *dK, hipStream_t s);
// Log-likelihood pieces in the basis U (x) Q (gram.hip): with Kt_p = amax_p Q_p T_p Q_p^T (T_p tridiagonal: d_p, e_p) the
// block of Ks (x) Kt + sig2 I of spatial eigen-row x' and temporal parity p is Q_p (es[x'] amax_p T_p + sig2 I) Q_p^T;
// *out_sumlog = sum of the log pivots of all these tridiagonal matrices (= sum log D), *out_quad = sum over the rows w of
// W = U^T Y Q ([x'][r][t~], rows of nt) of w^T (.)^-1 w by one forward recurrence each.  No temporal eigenvectors.
// host_slot (pinned, device-accessible) != null: the final sums' launch also writes {sum log D, quadratic form} to host_slot[0..1]
// and the status_doubles doubles at status_src to host_slot + status_at -- returns true when it did (the caller then skips its copy)
bool k_ll_tridiag(gpcsd_ctx *c, const double *W, const double *es, const double *const d[2], const double *const e[2],
                  const double *const amax[2], const double *sig, int nx, int R, int nt, const int np[2], const int c0[2],
                  double *out_sumlog, double *out_quad, hipStream_t s, double *host_slot = nullptr, const double *status_src = nullptr,
                  int status_at = 0, int status_doubles = 0);
// The same systems solved: B[x'][r][p block] = (es[x'] amax_p T_p + sig2 I)^-1 W[x'][r][p block] (B may be W).  The posterior mean
// in the basis U (x) Q -- what (W V) / D is in the basis U (x) V -- without the temporal eigenvectors.
// k_tridiag_solve_pass: trials per pass of the kernel for column blocks of up to npmax (its z lives in LDS); 0 = unsupported.
int k_tridiag_solve_pass(int npmax, int R);
void k_tridiag_solve(gpcsd_ctx *c, const double *W, double *B, const double *es, const double *const d[2], const double *const e[2],
                     const double *const amax[2], const double *sig, int nx, int R, int nt, const int np[2], const int c0[2],
                     hipStream_t s);
void k_add_diag(gpcsd_ctx *c, double *A, int n, double v, hipStream_t s, const HpDev *tab = nullptr, int B = 1, long s_out = 0);
void k_shift_copy(gpcsd_ctx *c, const double *src, double *dst, int n, double v, hipStream_t s);     // dst = src + v
void k_sum_partials(gpcsd_ctx *c, double *out, const double *P, long n, int parts, hipStream_t s);
// D[x*nt + i] = es[x]*et[i] + sig[x or 0]; also sumlog -> *sumlog_out (deterministic)
// with a table (B sets): scalar noise from tab[b].sig2n when nsig == 1, else set b's list at sig + b * nx
// Dinv (optional) = 1/D elementwise.  sumlog_out == nullptr: no final sum; the per-block partials stay in the ctx buffer
// "buildD_partials" and the return value is their count (a later launch may fold the sum in, or nobody needs it)
int k_build_D(gpcsd_ctx *c, const double *es, int nx, const double *et, int nt, const double *sig, int nsig, double *D,
              double *Dinv, double *sumlog_out, hipStream_t s, const HpDev *tab = nullptr, int B = 1, long s_sumlog = 0);
// lfp host layout [x][t][r] -> device layout [x][r][t] (and back for predictions [z][r][t] -> [z][t][r])
void k_swap_last2(gpcsd_ctx *c, const double *in, double *out, int n0, int n1, int n2, hipStream_t s);
void k_fill(gpcsd_ctx *c, double *p, long n, double v, hipStream_t s);
// in[(z, r)][c*n2 + t] (C components side by side) -> list[c][z][t][r] (optional) and sum[z][t][r] = sum_c
void k_swap_last2_sum(gpcsd_ctx *c, const double *in, int C, double *list, long list_stride, double *sum, int n0, int n1, int n2,
                      hipStream_t s);

// folded-basis helpers (gram.hip): rectangular fold of a covariance that commutes with the reflections of its row and
// column grids (out_ss: rs.ns x cs.ns, out_aa: rs.na x cs.na), the fold of the resident data in both indices, and the
// final pass of a folded prediction (unfold in site and time, (r, t) -> (t, r), sum over components)
void k_sym_fold_rect(gpcsd_ctx *c, const double *K, long ldk, const SymDev &rs, const SymDev &cs, double *out_ss, double *out_aa,
                     hipStream_t s);
// G (n,n) = F^T diag(Gss, Gaa) F for B sets (inputs s_in apart: Gss (ns,ns) followed by Gaa (na,na); outputs n*n apart)
void k_sym_unfold_mat(gpcsd_ctx *c, const double *Gf, long s_in, const SymDev &sy, int n, double *out, hipStream_t s, int B = 1);
void k_fold_lfp(gpcsd_ctx *c, const double *Y, int nx, int R, int nt, const SymDev &ss, const SymDev &st, double *out,
                hipStream_t s);
// nsP / naP > 0: the (parity, component) column blocks of `in` are padded to that many columns (128-byte aligned blocks)
void k_unfold_swap_sum(gpcsd_ctx *c, const double *in, int C, double *list, long list_stride, double *sum, int R, int nt,
                       const SymDev &sz, const SymDev &st, hipStream_t s, int nsP = 0, int naP = 0, long ldin = 0);

// ---------------------------------------------------------------- eigensolver (eigh.hip)
// Symmetric eigendecomposition of A (n,n) on device.  evals ascending; evecs (n,n) row-major with
// eigenvectors in COLUMNS (numpy.linalg.eigh convention).  A is destroyed.  status: device int (0 ok).
void eigh_device(gpcsd_ctx *c, double *A, int n, double *evals, double *evecs, int *d_status, hipStream_t s,
                 const char *tag);
constexpr int MAX_EIG_BATCH = 4;
// One CLASS of eigenproblems: `count` replicas of the same order n whose inputs / outputs sit a fixed stride apart
// (replica r reads A + r*sA, writes w + r*sw and Z + r*sZ).  The launches of a chain are shared by up to MAX_EIG_BATCH
// classes (the symmetric / antisymmetric halves of Ks and Kt) times any number of replicas (hyper-parameter sets of a
// batched evaluation): kernel arguments describe the classes by value, a workgroup derives its replica's pointers by
// adding replica * stride -- no per-problem table in the arguments or in memory.
struct EigReq {
    double *A;
    int n;
    double *w, *Z;
    const char *tag;
    int count = 1;
    long sA = 0, sw = 0, sZ = 0;
    bool prefilled = false;      // the scaled matrix is already in the class arena (eigh_arena_view): no scaling / copy pass
};
// Where the tridiagonalisation of class `tag` (order n, `count` replicas) expects its input: A0 = the matrix divided by
// *amax (replica r at A0 + r*blk, *amax at amax + r*blk), with the reflector storage V ((n + 64) x n) and tau (n + 64) zeroed.
// A caller that can generate the scaled matrix itself (k_temporal_fold_fill) writes it there and submits the class as
// `prefilled`, which takes the fold / absmax / scale launches out of the dependent chain.
struct EigArenaView {
    double *A0, *V, *tau, *amax;
    long blk;
    double *d, *e;               // the tridiagonal of A0 = Q T Q^T once the tridiagonalisation has run (n entries each)
    bool *psd;                   // host flag of the class (gpcsd_ctx::arena_psd): a fill that writes a positive semi-definite matrix sets it
};
int eigh_regtail_rows();         // rows of a whole problem the register tail (sytrd_regtail.hpp) holds: stage 5 applies up to here
EigArenaView eigh_arena_view(gpcsd_ctx *c, const char *tag, int n, int count);
// tags of the two half-size classes of problem `slot` (0 / 1) of eigh_pair_device: [0] symmetric, [1] antisymmetric
const char *const *eigh_fold_tags(const gpcsd_ctx *c, int slot);     // (slot 1: the tag set of the context's current generation, gpcsd_ctx::tgen)
// fold of a PSD matrix (+ diagonal shift per replica) straight into the class arenas, scaled: see eigh.hip
void k_psd_fold_fill(gpcsd_ctx *c, const double *K, int n, long sK, int nrep, const double *shift, const SymDev &sy,
                     const EigArenaView &as, const EigArenaView &aa, int *status, int status_stride, hipStream_t s,
                     const HpDev *tab = nullptr,        // tab: replica r adds tab[r].jitter instead of shift[r] (any nrep)
                     bool tab_shifts_nonneg = false);   // ... all of which are >= 0 (the arenas then count as PSD: EigArenaView::psd)
// The temporal chain's input in ONE launch: the symmetric / antisymmetric fold of Kt = sum_c sigma2_c k_c(t_i - t_j) for
// `nrep` hyper-parameter sets, evaluated entry by entry from the time grid (the same expressions, in the same order, as
// k_temporal_gram followed by the eigensolver's fold), divided by a power of two >= 2 sum_c sigma2_c (>= every entry of
// either block) and written where the tridiagonalisation reads it (eigh_arena_view of the two half-size classes), reflector
// storage zeroed, *amax set.  Replaces temporal Gram -> fold -> absmax -> scale/copy/zero (five dependent launches).
// Non-finite entries are zeroed and reported in status[r * status_stride] (4), as the scaling pass does.
struct TemporalSet {
    int ncomp;
    int kind[GPCSD_MAX_TEMPORAL];
    double ell[GPCSD_MAX_TEMPORAL], sigma2[GPCSD_MAX_TEMPORAL];
    __host__ __device__ double ell_of(int c) const { return ell[c]; }
    __host__ __device__ double sigma2_of(int c) const { return sigma2[c]; }
};
void k_temporal_fold_fill(gpcsd_ctx *c, const TemporalSet *sets, int nrep, const double *t, int n, const SymDev &sy,
                          const EigArenaView &as, const EigArenaView &aa, int *status, int status_stride, hipStream_t s);
// ... for B sets from a device table of hyper-parameters (scale formed on the device by the same rule)
void k_temporal_fold_fill_tab(gpcsd_ctx *c, const HpDev *tab, int B, const double *t, int n, const SymDev &sy,
                              const EigArenaView &as, const EigArenaView &aa, int *status, int status_stride, hipStream_t s,
                              bool variances_nonneg);   // every sigma2 of the table is >= 0: the blocks are PSD (EigArenaView::psd)
// flat problem index g -> (class, replica) from the prefix sums start[0..MAX_EIG_BATCH] (unused classes repeat the total)
__device__ __forceinline__ void class_of(const int *start, int g, int &cls, int &rep) {
    cls = (g >= start[1]) + (g >= start[2]) + (g >= start[3]);
    rep = g - start[cls];
}
// Two independent problems at once (Ks and Kt of one likelihood evaluation): every stage is batched so they share
// launches; with a known symmetry each splits into two half-size problems first.  Either n may be <= 0 to skip.
// need_merged = false: a caller that stays in the folded basis (eigh_fold_view) skips the unfold + rank merge of folded problems
// count > 1: `count` replicas of both problems (inputs n*n apart, eigenvalues n apart, eigenvectors n*n apart); replica r
// reports numerical failure in d_status[r * status_stride] (status_stride 0: one shared word)
// prefolded_mask bit p: the folded halves of problem p are already in their class arenas, scaled (see EigArenaView); A_p is
// then not read (symmetry folding must apply to that problem: eigh_fold_view(...).on)
// stage: 0 = the whole solve.  Staged (capi_fused.inl: the log-likelihood's shifted tridiagonal systems need the temporal side
// only as far as A / amax = Q T Q^T), same arguments every time:
//   1 = the tridiagonalisation (chain stream);
//   3 = the T factors of its reflector panels and the orthogonal factor Q itself (eigh_Q_view; T in EigArenaView::d / e) -- on
//       ANY stream behind stage 1: it reads the reflectors, which stages 2 and 4 leave alone;
//   2 = the divide & conquer of the tridiagonal matrix (chain stream, behind stage 1);
//   4 = the back-transformation (chain stream, behind stage 2 AND stage 3: it needs the T factors).
// Staged solves need folded (prefolded or not), tridiagonalisation-path problems within the fused back-transformation's size
// (eigh_stageable).
void eigh_pair_device(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, const SymDev *sym0, double *A1, int n1,
                      double *w1, double *Z1, const SymDev *sym1, int *d_status, hipStream_t s, bool need_merged = true,
                      int count = 1, int status_stride = 0, int count1 = -1,   // count1 > 0: replicas of problem 1 (else = count)
                      int prefolded_mask = 0, int stage = 0);
bool eigh_stageable(const SymDev *sy, int n);
// Q of class `tag` after stage 3: (n, n) row-major per replica, replicas n*n apart
double *eigh_Q_view(gpcsd_ctx *c, const char *tag, int n, int count);
// Half-size results of a symmetry-folded problem, in fold order (see eigh.hip); on == false: the problem is not folded.
struct FoldView {
    bool on = false;
    int ns = 0, na = 0;
    double *w = nullptr, *U = nullptr;      // w = (ws | wa);  U = (Us, ns x ns | Ua, na x na), eigenvectors in columns
    long sw = 0, sU = 0;                    // replica strides of w (= n) and U (= ns^2 + na^2)
};
FoldView eigh_fold_view(gpcsd_ctx *c, int slot, const SymDev *sy, int n, int count = 1);
// fused compact-WY back-transformation (wy.hip): all panels of all problems in two launches
struct WyProb {
    const double *V, *tau;   // reflectors by rows ((n + 64) x n, zero padded), tau (n + 64)
    double *T, *Z;           // T factors (npanels x 64 x 64 workspace), eigenvector matrix updated in place
    int n, npanels, nrefl;
    double *w_scale = nullptr;       // optional: eigenvalues of the SCALED matrix, multiplied by *amax in the apply launch
    const double *amax = nullptr;    // (saves the separate rescale launch at the end of the dependent chain)
    long blk = 0, sZ = 0, sw = 0;    // replica strides: V / tau / T / amax live in the class arena (blk), Z and w_scale are the caller's
    int z_identity = 0;              // the apply launch starts from Z = I (and writes Q itself): Z is only written
};
struct WyBatch {
    WyProb p[MAX_EIG_BATCH];         // one entry per class
    int start[MAX_EIG_BATCH + 1];    // prefix sums of the replica counts (see class_of)
    unsigned long long *clk = nullptr;   // measurement aid (GPCSD_WY_CLK=1): wall-clock stamps of workgroup (0, 0) at its phase boundaries
    int *status = nullptr;               // stage 5 (wy_q_pipeline): a gate whose time ran out reports 7 here (a scheduling miss: the
                                         // collecting call evaluates again, unpipelined -- gpcsd_ctx::q_pipe_timeouts)
    unsigned long long gate_ticks = 20000000ull;   // the gate's patience, 100 MHz ticks; 0: give up at once (test aid)
};
__device__ __forceinline__ WyProb wy_resolve(const WyBatch &b, int g) {
    int cls, rep;
    class_of(b.start, g, cls, rep);
    WyProb P = b.p[cls];
    const long o = rep * P.blk;
    P.V += o; P.tau += o; P.T += o;
    if (P.amax) P.amax += o;
    P.Z += rep * P.sZ;
    if (P.w_scale) P.w_scale += rep * P.sw;
    return P;
}
bool wy_fused_supported(int nmax);
// prep_done: the T factors were already formed by the D&C leaf launch (stedc_batch_device with a WyBatch)
void wy_batch_device(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s, bool prep_done = false);
void wy_prep_device(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s);     // the T factors only
// Q panel by panel behind the progress words of a register tail that is still running (wy.hip); chunk: the caller's work on the
// columns [col0[i], col1[i]) of class i's Q that the panel just applied completes
void wy_q_pipeline(gpcsd_ctx *c, const WyBatch &b, int nclass, hipStream_t s,
                   const std::function<void(const int *col0, const int *col1)> &chunk, bool one_launch = false);
// stages of the large-n solver, exposed for tests / diagnostics
void sytrd_device(gpcsd_ctx *c, double *A, int n, double *d, double *e, double *V, double *tau, hipStream_t s);
void stedc_device(gpcsd_ctx *c, const double *d, const double *e, int n, double *w, double *Z, int *d_status,
                  hipStream_t s, const char *tag);

// ---------------------------------------------------------------- analytic gradient pieces (grad.hip)
// a_x = sum_i et_i/D_xi, b_i = sum_x es_x/D_xi, s1 = sum 1/D
// B > 1: B hyper-parameter sets, every array contiguous per set (D nx*nt, es nx, et nt, a nx, b nt apart), s1 s_s1 apart
void k_D_sums(gpcsd_ctx *c, const double *D, const double *es, const double *et, int nx, int nt, double *a, double *b,
              double *s1_out, hipStream_t s, int B = 1, long s_s1 = 0);
// Ghs[y][x] += -1/2 (sig_y - sig_x)/(es_x - es_y) Ssum[x][y] for x != y, |es_x - es_y| > tiny
// B > 1: per hyper-parameter set (Ghs, Ssum nx*nx apart, es and sig nx apart)
void k_siglist_eigvec_term(gpcsd_ctx *c, double *Ghs, const double *Ssum, const double *es, const double *sig, int nx,
                           double tiny, hipStream_t s, int B = 1);
// out[b] = sum_{x,i} alpha[(x*nb + b)*nt + i]^2 / D[x*nt + i]
void k_per_trial_quad(gpcsd_ctx *c, const double *alpha, const double *D, int nx, int nb, int nt, double *out, hipStream_t s);
// out[x] = sum_k B[x*rowlen + k]^2
void k_rowgroup_sumsq(gpcsd_ctx *c, const double *B, int nrows, long rowlen, double *out, hipStream_t s);
// out (n,n) = scale * sum_b in[b*stride + e] + dscale * diag(dvec)
// B > 1: per hyper-parameter set, inputs s_in apart, dvec s_dvec apart (default n), out s_out apart (default n*n)
void k_batch_reduce(gpcsd_ctx *c, const double *in, int nb, long stride, int n, double scale, const double *dvec, double dscale,
                    double *out, hipStream_t s, int B = 1, long s_in = 0, long s_dvec = -1, long s_out = -1);
// out[2c] = <Gt, dKt/d ell_c>, out[2c+1] = <Gt, dKt/d sigma2_c>
// with a table: B sets (Gt nt*nt apart, outputs s_out apart); hp still supplies n_temporal
void k_temporal_grad(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *Gt, const double *t, int nt, double *out2C,
                     hipStream_t s, const HpDev *tab = nullptr, int B = 1, long s_out = 0);
// out[b][i * R + r] = in[b * s_in + i], i < n, r < R, b < B
void k_repeat_rows(gpcsd_ctx *c, const double *in, long s_in, int n, int R, int B, double *out, hipStream_t s);
// out[k] = <Gt, dK_k> for nm matrices dK_k (n2 doubles each, contiguous): user-defined temporal covariances
void k_frob_inner(gpcsd_ctx *c, const double *Gt, const double *dK, long n2, int nm, double *out, hipStream_t s);
// out[0..1] = <M, dKgl/d ell_1>, <M, dKgl/d ell_2>   (ngl2 == 0: 1D, only out[0] meaningful)
// out2[set * s_out + q] = <P, T_q>, q = 0, 1: the two spatial length-scale derivatives in the Kronecker form (grad.hip)
void k_frob_pair(gpcsd_ctx *c, const double *P, const double *T1, const double *T2, long n, double *out2, hipStream_t s, int B,
                 long s_out);
void k_kgl_grad(gpcsd_ctx *c, const double *M, const double *Kgl, const double *gx1, const double *gx2, int G, int ngl2,
                double ell1, double ell2, double *out2, hipStream_t s, const HpDev *tab = nullptr, int B = 1, long s_out = 0);
// out[0] = 2 <S, dA/dR>
void k_fwdR_grad(gpcsd_ctx *c, const double *S, const double *x, int nx, const double *gx1, const double *gw1, const double *gx2,
                 const double *gw2, int G, int ngl2, double R, double eps, double *out1, hipStream_t s, const HpDev *tab = nullptr,
                 int B = 1, long s_out = 0);

// ---------------------------------------------------------------- Cholesky (chol.hip)
// In-place lower Cholesky of A (n,n) row-major; strictly-upper part zeroed.  d_status: 0 ok, k+1 = pivot k <= 0.
void potrf_device(gpcsd_ctx *c, double *A, int n, int *d_status, hipStream_t s);
void potrf_diag128_probe(gpcsd_ctx *c, double *A, int n, double *X, int *d_status, unsigned long long *clk_dev, hipStream_t s);
// X = L^{-1} B in place (B (n,nrhs) row-major)
void trsm_lower_device(gpcsd_ctx *c, const double *L, int n, double *B, int nrhs, hipStream_t s);
void logdet_chol_device(gpcsd_ctx *c, const double *L, int n, double *out, hipStream_t s);
void sumsq_device(gpcsd_ctx *c, const double *x, long n, double *out, hipStream_t s);

}  // namespace gpcsd
