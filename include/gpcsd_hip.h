/*
 * gpcsd_hip.h -- C ABI of libgpcsd_hip.so, the MI355X (gfx950) implementation of the
 * GPCSD covariance-assembly / Kronecker-eigen marginal-likelihood / posterior-predict
 * hot path.
 *
 * The reference (natalieklein/gpcsd) is pure Python and has no FFI; the boundary it
 * offers is its Python class + operator surface.  Every entry point below names the
 * reference function (file:line under /root/reference/) whose arithmetic it replaces;
 * the Python package `gpcsd_amd` mirrors the reference classes and calls these through
 * ctypes (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *   - All arrays are C-contiguous float64 HOST buffers owned by the caller unless a
 *     name ends in `_dev`.  The library never retains a host pointer past the call.
 *   - Every function returns int: 0 ok; >0 numerical failure (eigensolver did not
 *     converge, Cholesky pivot <= 0 at column k -> return k+1) which the Python shim
 *     turns into numpy.linalg.LinAlgError exactly where the reference would raise it
 *     (gpcsd1d.py:219, gpcsd2d.py:217,258); <0 usage / HIP runtime error.
 *     NaN/Inf in results are returned, not trapped (reference runs under
 *     np.seterr(all='ignore'), gpcsd1d.py:7).
 *   - One ctx = one device + its four HIP streams (main, temporal chain, spatial
 *     chain, side products: DESIGN.md 4.8) + the resident data; calls on one ctx must
 *     be serialised by the caller, different contexts may be used from different
 *     threads.  Contexts are created lazily by the Python layer (fork safety).
 */
#ifndef GPCSD_HIP_H
#define GPCSD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPCSD_MAX_TEMPORAL 8
#define GPCSD_KIND_SE      0   /* covariances.py:257-271 */
#define GPCSD_KIND_MATERN  1   /* covariances.py:291-305 */
#define GPCSD_KIND_HOST    2   /* user-defined GPCSDTemporalCov subclass (covariances.py:235-238): its Gram matrices are
                                  evaluated by the caller's compute_Kt and handed over with gpcsd_set_host_temporal_gram */

/* Capacity limits of this build (no counterpart in the reference, which is bounded by host memory only).  Exceeding one
 * returns GPCSD_ERR_CAPACITY (-7), which the Python layer raises as gpcsd_amd.GPCSDCapacityError -- a RuntimeError, so
 * that fit()'s LinAlgError / ValueError handlers (gpcsd1d.py:219, gpcsd2d.py:217,258) do not swallow it.
 *   GPCSD_MAX_EIG_N           rows of one symmetric eigenproblem after symmetry folding (a mirror-symmetric grid of up
 *                           to 8192 points): nx and nt of the fused calls, n of gpcsd_eigh / gpcsd_eig_D
 *   GPCSD_MAX_GEMM_LD_KMAJOR  doubles in one row of a flat GEMM operand whose rows are the contracted index: ntrials * nt
 *                           of the resident block of trials (8.4 M: 16 777 trials of 500 samples, 25 GB of fp64 at 384
 *                           electrodes; shard trials over ranks beyond that)                                          */
#define GPCSD_MAX_EIG_N           4096
#define GPCSD_MAX_GEMM_LD_KMAJOR  (1L << 23)
#define GPCSD_ERR_CAPACITY      (-7)

#define GPCSD_PRED_CSD  1
#define GPCSD_PRED_LFP  2
#define GPCSD_PRED_BOTH 3

typedef struct gpcsd_ctx gpcsd_ctx;

/* Hyper-parameters read from the reference's mutable param dicts at every call
 * (gpcsd1d.py:53-62, gpcsd2d.py:67-79, covariances.py:48,174,254,288). */
typedef struct gpcsd_hparams {
    double R;                               /* forward-model radius                      */
    double eps;                             /* 2D singularity offset (ignored in 1D)     */
    double ell_s[2];                        /* spatial length scale(s): 1D uses [0]      */
    int    n_temporal;                      /* number of temporal components (<= 8)      */
    int    kind[GPCSD_MAX_TEMPORAL];        /* GPCSD_KIND_*                              */
    double ell_t[GPCSD_MAX_TEMPORAL];
    double sigma2_t[GPCSD_MAX_TEMPORAL];
    int    n_sig2n;                         /* 1 = scalar noise, nx = per-electrode list */
    const double *sig2n;                    /* host pointer, n_sig2n doubles             */
    double jitter;                          /* JITTER added to Ks in loglik (gpcsd1d.py:17, gpcsd2d.py:16) */
} gpcsd_hparams;

/* ---- context -------------------------------------------------------------------- */
int  gpcsd_ctx_create(int device, gpcsd_ctx **out);
int  gpcsd_ctx_destroy(gpcsd_ctx *ctx);
/* The HIP streams a context queues on, as integers (hipStream_t): which = 0 the main stream (every fused call's products and
 * results; an integrator orders its own HIP work against the library with an event on this one), 1 / 2 the temporal / spatial
 * eigen chains (utility_functions.py:58-59), 3 the side stream.  A closed context's four streams go back to a per-device pool and
 * the next context of the process takes them over (the stream -> hardware queue mapping is fixed when a stream is created: models
 * opened one after another run on the same queues); which = -1 / -2 (ctx may be NULL): stream sets created / taken over so far in
 * this process.  GPCSD_STREAM_POOL=0 switches the pool off. */
int  gpcsd_ctx_stream_handle(gpcsd_ctx *ctx, int which, unsigned long long *out);
const char *gpcsd_last_error(gpcsd_ctx *ctx);     /* ctx may be NULL: last global error */
int  gpcsd_version(void);
/* waits for everything queued on the context; also reports (rc > 0) a pending asynchronous gpcsd_predict_resident's failure */
int  gpcsd_device_synchronize(gpcsd_ctx *ctx);
/* Page-locked host blocks for result arrays (gpcsd2d.py:328-334 returns host arrays of (1+C)*nz*nt*ntrials doubles: 231 MB
 * at 384 x 500 x 50).  Outputs handed to gpcsd_predict / gpcsd_fetch / gpcsd_sample_prior may live in such a block, in which
 * case the device-to-host copy is a single DMA at link speed; any other host pointer still works (staged, slower). */
int  gpcsd_host_alloc(size_t bytes, void **out);
int  gpcsd_host_free(void *p);
/* PCI address of device `device` ("0000:c5:00.0", NUL-terminated into buf[len], len >= 16): what a host needs to find the
 * device's NUMA node (/sys/bus/pci/devices/<address>/numa_node) and keep its threads there.  On the two-socket hosts of the
 * MI355X pool a host process free to migrate between the sockets ran the queued step of bench.py at 1.12 .. 1.25 ms from run to
 * run, one kept on either socket's CPUs at 1.13 (gpcsd_amd._hip.bind_host_to_device_numa; `numactl`/`taskset` do the same from
 * outside).  No context needed. */
int  gpcsd_device_pci_bus_id(int device, char *buf, int len);

/* ---- resident data (copied; re-laid-out on device) ----------------------------- */
/* lfp is (nx, nt, ntrials) C-order as held by GPCSD{1,2}D.lfp (gpcsd1d.py:34, gpcsd2d.py:36);
 * stored on device electrode-major / trial / time so both projections are flat GEMMs. */
int gpcsd_set_lfp(gpcsd_ctx *ctx, const double *lfp, int nx, int nt, int ntrials);
/* GPCSD1DSpatialCov.__init__ state (covariances.py:12-27): electrodes + mapped GL rule */
int gpcsd_set_geometry_1d(gpcsd_ctx *ctx, const double *x, int nx,
                          const double *gl_x, const double *gl_w, int ngl);
/* GPCSD2DSpatialCov.__init__/reset_x state (covariances.py:99-137): xy is (nx,2) */
int gpcsd_set_geometry_2d(gpcsd_ctx *ctx, const double *xy, int nx,
                          const double *gl_x1, const double *gl_w1, int ngl1,
                          const double *gl_x2, const double *gl_w2, int ngl2);
int gpcsd_set_time(gpcsd_ctx *ctx, const double *t, int nt);   /* GPCSDTemporalCov.t */
/* User-defined temporal covariances (covariances.py:235-238; gpcsd1d.py:118-120 accepts any object with compute_Kt):
 * Kt = sum_c cov_c.compute_Kt() (nt, nt) and, for predict, Kt_cross[c] = cov_c.compute_Kt(tstar) (ncomp, ntstar, nt;
 * may be NULL for loglik), evaluated by the caller.  Copied; used by every later fused call whose hparams carry a
 * component of kind GPCSD_KIND_HOST, in place of the built-in Gram builders (the eigensolver, projections and predict
 * chain are unchanged; the reflection fold of the time axis is skipped since such a kernel need not be stationary).
 * Kt == NULL clears it.  gpcsd_loglik_grad needs the derivatives of such a Gram as well (gpcsd_set_host_temporal_dgram); without
 * them it refuses such hparams (-3). */
int gpcsd_set_host_temporal_gram(gpcsd_ctx *ctx, const double *Kt, int nt, const double *Kt_cross, int ncomp, int ntstar);
/* Derivatives of that Gram matrix with respect to the natural temporal hyper-parameters, evaluated by the caller (the
 * reference differentiates any GPCSDTemporalCov subclass by tracing it with autograd, gpcsd1d.py:211; here the class offers
 * compute_dKt(name)): dKt is (nmat, nt, nt) with nmat = 2 * n_temporal, ordered (d/d ell_c, d/d sigma2_c) per component.
 * Copied; belongs to the Gram handed over last (a new gpcsd_set_host_temporal_gram clears it).  With it gpcsd_loglik_grad
 * accepts GPCSD_KIND_HOST components: its temporal entries are <Gt, dKt_k>, one evaluation per gradient.  dKt == NULL clears. */
int gpcsd_set_host_temporal_dgram(gpcsd_ctx *ctx, const double *dKt, int nt, int nmat);

/* ---- operator surface (stand-alone; host in / host out) ------------------------- */
/* b_fwd_1d(r, R)                        forward_models.py:9-17   (elementwise, n values) */
int gpcsd_b_fwd_1d(gpcsd_ctx *ctx, const double *r, long n, double R, double *out);
/* predictcsd_trad_1d / _2d  predict_csd.py:3-16 / :19-31: minus the second difference along the electrode axis of host data
 * viewed as (n_outer, n_axis, n_inner) -- 1-D: (1, nx, nt*ntrials), ends of the axis -0.0 (edge_nan = 0); 2-D, column-wise on
 * gridded data: (nx1, nx2, nt*ntrials), first and last column NaN (edge_nan = 1).  out has the shape of lfp. */
int gpcsd_trad_csd(gpcsd_ctx *ctx, const double *lfp, long n_outer, long n_axis, long n_inner, int edge_nan, double *out);
/* b_fwd_2d(delta1, delta2, R, eps, w)   forward_models.py:42-54  (w != NULL -> d1,d2 ignored) */
int gpcsd_b_fwd_2d(gpcsd_ctx *ctx, const double *d1, const double *d2, const double *w, long n,
                   double R, double eps, double *out);
/* GPCSDTemporalCov{SE,Matern}.compute_Kt(t, tprime)   covariances.py:257-271 / :291-305; out (n, m) */
int gpcsd_gram_temporal(gpcsd_ctx *ctx, int kind, const double *t, int n, const double *tp, int m,
                        double ell, double sigma2, double *out);
/* GPCSD1DSpatialCovSE.compute_Ks        covariances.py:50-56 ; out (nx, nx) */
int gpcsd_ks_csd_1d(gpcsd_ctx *ctx, const double *x, int nx, double ell, double *out);
/* GPCSD2DSpatialCovSE.compute_Ks        covariances.py:177-186 */
int gpcsd_ks_csd_2d(gpcsd_ctx *ctx, const double *xy, int nx, double ell1, double ell2, double *out);
/* compKphi_1d(R, xp)                    covariances.py:74-96 ; out (nx, nxp); xp NULL -> x */
int gpcsd_kphi_1d(gpcsd_ctx *ctx, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl,
                  double R, double ell, const double *xp, int nxp, double *out);
/* compKphig_1d(z, R)                    covariances.py:58-72 ; out (nx, nz) */
int gpcsd_kphig_1d(gpcsd_ctx *ctx, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl,
                   const double *z, int nz, double R, double ell, double *out);
/* compKphi_2d(R, eps, xp)               covariances.py:204-232 ; xy (nx,2), xp (nxp,2) or NULL */
int gpcsd_kphi_2d(gpcsd_ctx *ctx, const double *xy, int nx,
                  const double *gl_x1, const double *gl_w1, int ngl1,
                  const double *gl_x2, const double *gl_w2, int ngl2,
                  double R, double eps, double ell1, double ell2,
                  const double *xp, int nxp, double *out);
/* compKphig_2d(z, R, eps)               covariances.py:188-202 ; z (nz,2) */
int gpcsd_kphig_2d(gpcsd_ctx *ctx, const double *xy, int nx,
                   const double *gl_x1, const double *gl_w1, int ngl1,
                   const double *gl_x2, const double *gl_w2, int ngl2,
                   const double *z, int nz, double R, double eps, double ell1, double ell2, double *out);
/* numpy.linalg.eigh as used by comp_eig_D (utility_functions.py:58-59): ascending evals, evecs in columns */
int gpcsd_eigh(gpcsd_ctx *ctx, const double *A, int n, double *evals, double *evecs);
/* the same for a matrix the caller vouches is positive semi-definite (a Gram matrix such as Ks, Kt handed to comp_eig_D,
 * utility_functions.py:44-64): it may take the rank-revealing early exit of gpcsd_tail_early_exit (orders <= 192) */
int gpcsd_eigh_psd(gpcsd_ctx *ctx, const double *A, int n, double *evals, double *evecs);
/* `count` matrices of the same order through one shared chain of launches (batched hyper-parameter evaluations decompose
 * all their Gram matrices this way): A (count,n,n) -> evals (count,n), evecs (count,n,n); status[i] > 0: matrix i failed. */
int gpcsd_eigh_batch(gpcsd_ctx *ctx, const double *A, int n, int count, double *evals, double *evecs, int *status);
/* Diagnostics for the large-n eigensolver stages (no reference counterpart; LAPACK does these inside dsyevd):
 * Householder tridiagonalisation A = Q T Q^T: d (n), e (n, last unused), reflectors V (n,n) by rows, tau (n) */
int gpcsd_debug_sytrd(gpcsd_ctx *ctx, const double *A, int n, double *d, double *e, double *V, double *tau);
/* divide & conquer eigen-decomposition of tridiag(d (n), e (n-1)): w ascending, Z (n,n) eigenvectors in columns */
int gpcsd_debug_stedc(gpcsd_ctx *ctx, const double *d, const double *e, int n, double *w, double *Z);
/* comp_eig_D(Ks, Kt, sig2n)            utility_functions.py:44-64 ; Dvec has nx*nt entries */
int gpcsd_eig_D(gpcsd_ctx *ctx, const double *Ks, int nx, const double *Kt, int nt,
                const double *sig2n, int n_sig, double *Qs, double *Qt, double *Dvec);
/* numpy.linalg.cholesky (gpcsd1d.py:303-304, gpcsd2d.py:343-350): lower factor, upper part zeroed */
int gpcsd_potrf(gpcsd_ctx *ctx, const double *A, int n, double *L);
/* sum(log(diag(L))) * 2 = log det(A) for A = L L^T */
int gpcsd_logdet_chol(gpcsd_ctx *ctx, const double *L, int n, double *out);
/* solve L X = B (lower, non-transposed), B and X are (n, nrhs) */
int gpcsd_trsm_lower(gpcsd_ctx *ctx, const double *L, int n, const double *B, int nrhs, double *X);
/* C(M,N) = op(A) op(B), row-major; transA: A stored (K,M); transB: B stored (N,K).  The fp64 MFMA core. */
int gpcsd_gemm(gpcsd_ctx *ctx, int transA, int transB, int M, int N, int K,
               const double *A, const double *B, double *C);
/* dense cross-check path (scalar sig2n): chol(kron(Ks,Kt)+sig2n I) log-det + triangular solves.
 * lfp (nx,nt,ntrials) host.  Not a reference code path; see DESIGN.md. */
int gpcsd_loglik_dense_chol(gpcsd_ctx *ctx, const double *Ks, int nx, const double *Kt, int nt, double sig2n,
                            const double *lfp, int ntrials, double *out);

/* ---- fused hot calls on resident data ------------------------------------------- */
/* GPCSD1D.loglik() gpcsd1d.py:113-128 / GPCSD2D.loglik() gpcsd2d.py:136-151 */
int gpcsd_loglik(gpcsd_ctx *ctx, const gpcsd_hparams *hp, double *out);
/* Sharded form: out[0] = sum(log D) (identical on every shard), out[1] = sum_r sum alpha^2/D over the
 * resident trials; loglik = -0.5*R_total*out[0] - 0.5*sum_over_shards(out[1]) */
int gpcsd_loglik_parts(gpcsd_ctx *ctx, const gpcsd_hparams *hp, double *out2);
/* The same evaluation split in two: _async validates its arguments, queues the work and returns; _wait blocks until that
 * evaluation (and only that: not whatever was queued behind it) has finished and returns its out2 / rc.  Between the two
 * the caller may queue further calls on the context -- a gpcsd_predict_resident at the same hyper-parameters, say -- whose
 * eigen-chains then run beside this evaluation's GEMMs instead of after the host has come back for the result.  Up to
 * four evaluations may be outstanding per context (rc -3 beyond that); _wait collects them oldest first.  hp is read during
 * the _async call only.  rc > 0 from _wait:
 * numerical failure of this evaluation or of an earlier asynchronous call nobody has collected yet (status is sticky until
 * a synchronising call -- gpcsd_device_synchronize, any call that returns values -- has reported it).  A _wait that reports
 * a failure drains the context and clears the status, so evaluations queued AFTER it returned start clean; evaluations that
 * were already outstanding copied the status as it stood and report the failure too (a failed wait poisons those), and
 * the decomposition cache forgets both sides (a retry with the same hp re-solves). */
int gpcsd_loglik_parts_async(gpcsd_ctx *ctx, const gpcsd_hparams *hp);
int gpcsd_loglik_parts_wait(gpcsd_ctx *ctx, double *out2);
/* gpcsd_loglik_parts_async(hp_loglik) followed by gpcsd_predict_resident(hp_predict, ...) as ONE queued call: same results
 * bit for bit, collected the same way (gpcsd_loglik_parts_wait, gpcsd_fetch), but the two temporal eigenproblems go through
 * one chain of launches as two replicas and the two spatial ones through another -- replicas inside a chain are nearly free
 * on this GPU, independent chains are not.  The usual pair is one hyper-parameter set with and without the
 * jitter (GPCSD1D.loglik gpcsd1d.py:113-128 adds it, predict gpcsd1d.py:258 does not); any two sets are accepted.  Every
 * decomposition is still computed, unless the decomposition cache is on and the temporal hyper-parameters of the two sets
 * coincide: then that side is solved once, exactly as the cache would serve the second of two separate calls.  Falls back
 * to the two calls in sequence where the folded-basis path does not apply (per-electrode noise lists, user-defined temporal
 * covariances, grids or prediction sites without the mirror symmetry). */
int gpcsd_loglik_predict_async(gpcsd_ctx *ctx, const gpcsd_hparams *hp_loglik, const gpcsd_hparams *hp_predict,
                               const double *z, int nz, const double *tstar, int ntstar, int type, int want_lists);
/* Announce the NEXT gpcsd_loglik_predict_async.  A caller that knows the hyper-parameters of its next paired evaluation before
 * the current one has been collected -- a grid or a chain of proposals fixed in advance, a benchmark loop; NOT an optimiser, whose
 * next point depends on the value it is waiting for (gpcsd1d.py:211) -- passes them here right after queueing the current call:
 * the two decomposition chains of the next call (utility_functions.py:58-59 for Kt and Ks) are queued at once and start as soon as
 * their streams are free, i.e. under the current call's products instead of behind the host's collection of its log-likelihood.
 * The next gpcsd_loglik_predict_async with the same hyper-parameters, sites and times takes them over (same launches, same
 * buffers, same bits); any other evaluation on the context in between drops them (they have then run for nothing).
 * Returns 1 when queued, 0 when the paired form does not apply to these arguments (nothing queued), < 0 on error. */
int gpcsd_prefetch_pair(gpcsd_ctx *ctx, const gpcsd_hparams *hp_loglik, const gpcsd_hparams *hp_predict, const double *z, int nz,
                        const double *tstar, int ntstar);
/* how many front halves gpcsd_prefetch_pair queued, and how many of them a paired call took over */
int gpcsd_prefetch_stats(gpcsd_ctx *ctx, long *queued, long *taken);
/* Local log-likelihood pieces and the gradient of  L_loc = -0.5*ntrials_resident*out2[0] - 0.5*out2[1]  with respect
 * to the natural hyper-parameters [R, ell_s (dim), (ell_t, sigma2_t) per component, sig2n]; with a per-electrode
 * noise list (hp->n_sig2n == nx, indexed by eigen-row as utility_functions.py:54-63) the tail holds nx entries and the
 * spatial entries include the eigenvector-rotation term (the objective is then not a function of Ks alone).
 * Replaces the autograd tape of gpcsd1d.py:211 / gpcsd2d.py:250.  Both L_loc and grad are sums over trials plus a
 * term linear in the resident trial count, so shards combine by plain summation over ranks. */
int gpcsd_loglik_grad(gpcsd_ctx *ctx, const gpcsd_hparams *hp, double *out2, double *grad, int ngrad);
/* The same evaluation for `nsets` hyper-parameter sets in ONE chain of launches -- the restarts of fit() (gpcsd1d.py:193-220,
 * gpcsd2d.py:230-262) are independent optimiser chains whose evaluations are latency-bound, so sets evaluated together
 * share every launch.  out2 is (nsets, 2), grad (nsets, ngrad); status[i] > 0 reports a numerical failure of set i alone
 * (the others are valid).  Every set gets exactly the bits a gpcsd_loglik_grad call of its own returns.  All sets must have
 * the same temporal kernel kinds and the same number of noise entries: a scalar sig2n each, or (round 5) a per-electrode list
 * of nx entries each -- the restarts of auditory_lfp/fit_gpcsd_baseline.py:85-101, evaluated on the merged-order path with the
 * eigenvector-rotation term per set. */
int gpcsd_loglik_grad_batch(gpcsd_ctx *ctx, const gpcsd_hparams *hp, int nsets, double *out2, double *grad, int ngrad,
                            int *status);
/* GPCSD{1,2}D.predict(z, t, type) gpcsd1d.py:248-293 / gpcsd2d.py:289-334.
 * z (nz, dim), tstar (ntstar) with ntstar == nt (the reference raises ValueError otherwise -> rc -22).
 * Outputs (any may be NULL): *_list is (n_temporal, nz, ntstar, ntrials), sums are (nz, ntstar, ntrials). */
int gpcsd_predict(gpcsd_ctx *ctx, const gpcsd_hparams *hp, const double *z, int nz,
                  const double *tstar, int ntstar, int type,
                  double *csd_list, double *csd, double *lfp_list, double *lfp);
/* Same computation, results left in ctx-owned device buffers in the output layout (z, t, trial); nothing crosses
 * PCIe.  Buffers: "pred_out_csd", "pred_out_lfp" (nz*ntstar*ntrials) and, when want_lists,
 * "pred_out_csd_list", "pred_out_lfp_list" (n_temporal times that).  Read them back with gpcsd_fetch.
 * ASYNCHRONOUS on the symmetric-grid (folded) path: the call validates its arguments, queues the work and returns 0 with
 * the GEMM tail still in flight; any later call on the context is ordered behind it (the next call's temporal
 * eigen-chain runs beside that tail, which is the point).  A numerical failure (rc > 0) of such a call is reported by the
 * next call on the context that synchronises: gpcsd_fetch, gpcsd_device_synchronize, or any call that returns values
 * (gpcsd_loglik*, gpcsd_predict, ...).  Argument errors (rc < 0) are still reported by the call itself. */
int gpcsd_predict_resident(gpcsd_ctx *ctx, const gpcsd_hparams *hp, const double *z, int nz,
                           const double *tstar, int ntstar, int type, int want_lists);
/* copy `count` doubles of the named ctx-owned device buffer to host; rc -2 if the name is unknown; rc > 0 if the
 * asynchronous gpcsd_predict_resident that produced the buffer failed numerically */
int gpcsd_fetch(gpcsd_ctx *ctx, const char *name, double *host, long count);
/* Bytes this context has moved between the caller's PAGEABLE host memory and the device through its own page-locked bounce blocks.
 * The library never passes pageable caller memory to the HIP runtime: the runtime would register those pages with the driver and
 * keep the registration cached, and when the pages are later freed or moved the driver evicts every queue of the process for
 * 10-30 ms (measured: one such stall in every second later step loop of a process, none with this in place).  Arrays in
 * page-locked memory (gpcsd_host_alloc; the class API's results) are copied directly.  No reference counterpart (NumPy arrays
 * in, NumPy arrays out: gpcsd1d.py:21-62). */
int gpcsd_bounce_stats(gpcsd_ctx *ctx, long *bytes);
/* The device address and size in bytes of the same named buffer, for callers that go on working on the GPU instead of copying out:
 * the posterior means gpcsd_predict_resident leaves in HBM ("pred_out_csd", "pred_out_csd_list", "pred_out_lfp",
 * "pred_out_lfp_list": gpcsd1d.py:286-293 / gpcsd2d.py:327-334, layout (nz, nt, ntrials) / (C, nz, nt, ntrials)) gathered over the
 * ranks of a trial-sharded job by an RCCL all-gather (gpcsd_amd/dist.py), or handed to another library (__cuda_array_interface__,
 * DLPack).  Collects a queued prediction's deferred status (rc > 0 as gpcsd_fetch) and drains the context's streams first: the
 * buffer is complete on return and stays valid until the next call that writes it; the library keeps ownership. */
int gpcsd_device_buffer(gpcsd_ctx *ctx, const char *name, unsigned long long *dev_ptr, unsigned long long *bytes);
/* sample_prior with host-supplied standard normals (nx, nt, ntrials): Ls Z_r Lt^T
 * gpcsd1d.py:295-309 / gpcsd2d.py:336-360.  which: GPCSD_PRED_CSD (compute_Ks) or GPCSD_PRED_LFP (compKphi) */
int gpcsd_sample_prior(gpcsd_ctx *ctx, const gpcsd_hparams *hp, int which,
                       const double *normals, int ntrials, double *out);

/* Per-trial whitened quadratic forms  out[b] = sum( (Qs^T resid_b Qt)^2 / Dvec )  for nb residual matrices at once, given a
 * cached decomposition (Qs, Qt, Dvec) from gpcsd_eig_D.  resid is (nx, nt, nb) C-order like lfp.  This is the projection
 * kernel of loglik (gpcsd1d.py:124-127) exposed for downstream per-trial consumers, e.g. the shift-optimisation objective
 * of auditory_lfp/fit_mean_function.py:311-321 (alpha = Qs^T resid Qt; quad = -0.5 * sum(alpha^2 / Dvec)). */
int gpcsd_whitened_quad(gpcsd_ctx *ctx, const double *Qs, int nx, const double *Qt, int nt, const double *Dvec,
                        const double *resid, int nb, double *out);

/* Precision of the Gram builders (temporal SE / Matern, spatial SE, forward-model weights; covariances.py:50-96,177-232,
 * 257-305): bits = 64 (default) evaluates them exactly as the reference does; bits = 32 is the "fp32 kernel build + fp64
 * factor" variant of the fit benchmark: coordinates and hyper-parameters are rounded to float, exp/log/sqrt are
 * single precision, the result is widened and every later stage (GEMMs, eigensolver, likelihood, gradient) stays fp64.
 * Applies to the operator entry points and to the fused calls of this context. */
int gpcsd_set_gram_precision(gpcsd_ctx *ctx, int bits);

/* Folded-basis GEMMs: with mirror-symmetric electrode and time grids (detected in set_geometry / set_time) loglik and predict
 * run their projections (gpcsd1d.py:124-127, 262-279) as half-size products in the symmetric / antisymmetric basis; same
 * results to rounding, half the flops.  on = 0 / 1 switches the path for this context (default 1; GPCSD_NO_FOLD_GEMM=1 in
 * the environment disables it process-wide), on < 0 only queries.  *calls (optional) receives how many loglik / predict
 * calls of this context have taken the folded path so far. */
/* Multi-GPU from C (one process per GPU): trials are independent, every rank recomputes the (deterministic)
 * decompositions and owns a contiguous block of trials.  gpcsd_shard_block gives rank `rank` of `world` its block
 * [first, first+count) of `ntrials`; the rank uploads lfp[:, :, first:first+count] with gpcsd_set_lfp, calls
 * gpcsd_loglik_parts -> (sum log D, partial quad), sums the partial quad over ranks with the collective of its choice
 * (RCCL / MPI all-reduce of one double) and gpcsd_combine_loglik forms -R/2 sum log D - 1/2 quad (gpcsd1d.py:122,127-128).
 * predict / gpcsd_loglik_grad work on the rank's block the same way (gradient entries and L_loc add over ranks). */
int gpcsd_shard_block(int ntrials, int rank, int world, int *first, int *count);
int gpcsd_combine_loglik(int ntrials_total, double sumlog, double quad_sum_over_ranks, double *out);

/* Decomposition cache: a fused call whose spatial and / or temporal hyper-parameters, grids and precision equal those of
 * the previous fused call on this context reuses that side's eigendecomposition instead of repeating it (predict() right
 * after loglik() / fit(): neuropixels/fit_gpcsd2d.py:101-107; repeated predict() calls).  Kernels are deterministic, so the
 * results are bit-identical either way.  on = 1 / 0 switches it (default on), -1 only queries; *hits counts reused sides. */
int gpcsd_decomposition_cache(gpcsd_ctx *ctx, int on, long *hits);
int gpcsd_fold_gemm(gpcsd_ctx *ctx, int on, long *calls);
/* The paired call gpcsd_loglik_predict_async (loglik gpcsd2d.py:136-151 + predict :289-334 of one step) with EQUAL temporal
 * hyper-parameters in both sets -- the case of every loglik -> predict pair, whose only difference is the spatial jitter -- forms
 * the product X = Y~ Q of the resident data with the temporal basis once and lets the prediction read the log-likelihood's copy:
 * the two replicas of the temporal problem are the same matrix reduced by the same deterministic launches, so the second product
 * would recompute the same bits.  on = 1 / 0 switches it (default 1; GPCSD_PAIR_SHARE_X=0 for new contexts), < 0 only queries;
 * *calls counts the paired calls that shared the product.  Results are bit-identical either way. */
int gpcsd_pair_share_x(gpcsd_ctx *ctx, int on, long *calls);
/* ... and ONE spatial decomposition for the pair when the two sets have equal spatial hyper-parameters and differ by the jitter
 * only -- every loglik -> predict pair: the reference adds jitter * I to Ks in loglik (gpcsd1d.py:116, gpcsd2d.py:139) and not in
 * predict (gpcsd1d.py:258, gpcsd2d.py:296).  eigh(Ks + j I) has the eigenvectors of eigh(Ks) and its spectrum shifted by j
 * (utility_functions.py:58-59 applied to either), so the prediction takes the log-likelihood's eigenvectors, its spectrum minus j
 * and -- with X shared as well -- its projected data.  Exact in exact arithmetic; in floating point the prediction then differs from
 * the separately decomposed one by the eigensolver's own backward error (1e-13 relative at 384 x 500, tests/test_pair_share_s.py).
 * on = 1 / 0 switches it (default 0: it saves a large product and 154 MB of traffic per 384 x 500 x 50 step, but the step measured
 * slower, DESIGN 4.13, and the pair then differs from its fenced calls in the last bits; GPCSD_PAIR_SHARE_S=1 for new contexts),
 * < 0 only queries; *calls counts the pairs that took it. */
int gpcsd_pair_share_s(gpcsd_ctx *ctx, int on, long *calls);
/* Pipelined orthogonal factor (round 5).  The tridiagonal forms (gpcsd_ll_tridiag below) need Q of Kt = Q T Q^T
 * (utility_functions.py:58) and X = Y~ Q before anything else of their tails can start, and both used to wait for the END of the
 * tridiagonalisation -- the longest launch of a step.  Column j of Q only depends on the reflectors in front of it, so the
 * single-workgroup reduction publishes its progress every 64 reflectors and the T factor of that panel, the 64 finished columns of Q
 * and the matching 64 columns of X follow on another stream while the reduction is still at work on the next panel; behind it only
 * the last panel's share is left.  Same reflectors, same T factors; Q and X agree with the unpipelined form to rounding (the products
 * are summed in another order).  Applies to temporal blocks of at most 256 rows after symmetry folding.
 * on = 1 / 0 switches it (default 1; GPCSD_Q_PIPE=0 for new contexts -- cheaper under a profiler that serialises kernels, where a
 * launch that waits for a running one cannot make progress: gpcsd_q_pipeline_stats below), < 0 only queries;
 * *calls counts the temporal chains that took it. */
int gpcsd_q_pipeline(gpcsd_ctx *ctx, int on, long *calls);
/* The one place where a launch waits for a RUNNING kernel of another stream is bounded (0.2 s).  A wait that runs out is a
 * scheduling miss, not a numerical failure (kernels serialised by a profiler or a debugger, an oversubscribed card): the launch
 * leaves the unfinished reflectors alone, and the call that collects the evaluation -- the synchronous call itself,
 * gpcsd_loglik_parts_wait, or whichever call next synchronises behind a gpcsd_predict_resident -- switches the pipeline off for
 * the context (latched), evaluates again behind the end of the reduction (same T, Q, X as with on = 0) and returns that result.
 * *on: the switch as it stands; *timeouts: evaluations repeated for this reason; gate_ticks >= 0 sets the bound in 100 MHz ticks
 * (0: every wait gives up at once -- the test aid that drives the repeat path; GPCSD_QPIPE_GATE_TICKS for new contexts), < 0
 * leaves it.  No reference counterpart (utility_functions.py:58-59 is one LAPACK call). */
int gpcsd_q_pipeline_stats(gpcsd_ctx *ctx, int *on, long *timeouts, long long gate_ticks);
/* gpcsd_predict with host outputs (what the class API's predict() returns, gpcsd1d.py:286-293 / gpcsd2d.py:327-334): the last
 * product of a folded prediction is launched in chunks of prediction sites and every chunk's finished output rows are copied to
 * the caller's arrays while the next chunk computes (231 MB per call at 384 x 500 x 50: the copy is 4 of the call's 5 ms and no
 * longer waits for the whole product).  Applies to outputs of at least 32 MB through the fused last product; same bits.
 * on = 1 / 0 switches it (default 1; GPCSD_PRED_CHUNKED=0 for new contexts), < 0 only queries; *calls counts the calls that took it. */
int gpcsd_predict_chunked_copy(gpcsd_ctx *ctx, int on, long *calls);
/* The log-likelihood of the folded path in the basis U (x) Q instead of U (x) V.  With Kt = m Q T Q^T from the
 * tridiagonalisation alone (Q orthogonal, T tridiagonal) and the spatial side fully decomposed, Ks (x) Kt + sig2 I is a set of
 * SHIFTED TRIDIAGONAL matrices es[x'] m T + sig2 I: sum log D is the sum of the logs of their LDL^T pivots and the quadratic
 * form one forward recurrence per (x', trial) row of U^T Y Q -- exact algebra (gpcsd1d.py:113-128), no temporal eigenvectors, so
 * the log-likelihood does not wait for the temporal divide & conquer (which a prediction still needs and the chain still runs).
 * Same results to rounding (1e-15 relative on the log-likelihood at 384 x 500).  mode: 0 = off (the eigenvector form always),
 * 1 = on wherever it applies (mirror-symmetric time grid with halves beyond the Jacobi size, scalar noise, folded path),
 * 2 = on while max(nx, 64) * nt * ntrials <= 11 * 2^20 (the default: above that the step is bound by its GEMMs and the form
 *     costs 2-10 %),
 * < 0 only queries; GPCSD_LL_TRIDIAG=0|1|2 sets the mode of new contexts.  *calls counts the log-likelihoods evaluated this
 * way.  DESIGN.md 9. */
/* Late status words.  A temporal chain all of whose consumers take the tridiagonal form (a log-likelihood in this form; the
 * paired call when its prediction is in that form too, the default) runs its tridiagonalisation, T factors and Q only: the
 * divide & conquer and the back-transformation are not queued, and the context is idle behind every value-returning call.  Only
 * the paired call gpcsd_loglik_predict_async with an EIGENVECTOR-form prediction (GPCSD_PRED_TRIDIAG=0, or shapes the solve
 * kernel does not take) queues those two stages behind a log-likelihood that does not wait for them; they report into status
 * words of their own, which the call that JOINS that chain collects (the prediction's wait / gpcsd_device_synchronize): a failure
 * there is that call's rc > 0, never the log-likelihood's (whose value does not depend on them) and never an unrelated later
 * call's. */
int gpcsd_ll_tridiag(gpcsd_ctx *ctx, int on, long *calls);
/* Rank-revealing early exit of the tridiagonalisation (no reference counterpart -- numpy.linalg.eigh,
 * utility_functions.py:58-59, reduces every column): for matrices the LIBRARY's own Gram fills announce as positive
 * semi-definite (Ks, Kt of the model with non-negative variances and jitter; never a caller's matrix handed to gpcsd_eigh /
 * gpcsd_eig_D / gpcsd_set_host_temporal_gram) of at most 192 rows after folding, the Householder reduction stops once the trace
 * still to be reduced is below 64 unit roundoffs of the matrix's trace (backward error <= that trace).  on = 1 / 0 switches it
 * for this context (default 1; GPCSD_TAIL_EARLY_EXIT=0 for new contexts), < 0 only queries; *previous receives the setting
 * before the call.  Both settings are tested against each other (tests/test_hip_fullsize.py::test_tail_early_exit_*). */
int gpcsd_tail_early_exit(gpcsd_ctx *ctx, int on, int *previous);
/* Test aid, off by default and never read from the environment: with on = 1 the divide & conquer stage of a staged temporal
 * chain reports numerical failure 3 for every replica (drives the late status words above). */
int gpcsd_debug_fault_stage2(gpcsd_ctx *ctx, int on);

/* ---- several devices from ONE process ------------------------------------------------------- */
/* SURVEY 8(b)'s gpcsd_dist_*: one context per device, driven by one host process (the Python layer uses one process per GPU
 * over torch.distributed / RCCL instead: gpcsd_amd/dist.py).  The path has no device-to-device exchange: trials
 * are independent, every device recomputes the deterministic decompositions, and what is combined is one double per device --
 * summed on the host in device order.  devices == NULL: ordinals 0..ndev-1 (an ordinal may repeat: two contexts on one GPU). */
typedef struct gpcsd_dist gpcsd_dist;
int gpcsd_dist_create(int ndev, const int *devices, gpcsd_dist **out);
int gpcsd_dist_destroy(gpcsd_dist *dist);
int gpcsd_dist_size(gpcsd_dist *dist);
int gpcsd_dist_ctx(gpcsd_dist *dist, int i, gpcsd_ctx **ctx);          /* the i-th context, for the per-context knobs */
const char *gpcsd_dist_last_error(gpcsd_dist *dist);
/* the gpcsd_set_* calls on every device */
int gpcsd_dist_set_geometry_1d(gpcsd_dist *dist, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl);
int gpcsd_dist_set_geometry_2d(gpcsd_dist *dist, const double *xy, int nx, const double *gl_x1, const double *gl_w1, int ngl1,
                               const double *gl_x2, const double *gl_w2, int ngl2);
int gpcsd_dist_set_time(gpcsd_dist *dist, const double *t, int nt);
/* lfp (nx, nt, ntrials).  replicate = 0: device i gets the block of trials gpcsd_shard_block gives rank i (trial sharding,
 * BASELINE cfg4); replicate = 1: every device gets all trials (restart sharding, BASELINE cfg5) */
int gpcsd_dist_set_lfp(gpcsd_dist *dist, const double *lfp, int nx, int nt, int ntrials, int replicate);
/* loglik() over all trials  gpcsd1d.py:113-128 / gpcsd2d.py:136-151: queued on every device, partial sums added in device order */
int gpcsd_dist_loglik(gpcsd_dist *dist, const gpcsd_hparams *hp, double *out);
/* as gpcsd_loglik_grad, summed over the devices' blocks of trials */
int gpcsd_dist_loglik_grad(gpcsd_dist *dist, const gpcsd_hparams *hp, double *out2, double *grad, int ngrad);
/* the restarts of fit()  gpcsd1d.py:193-220: set k evaluated on device k mod ndev (needs replicate = 1); arguments as
 * gpcsd_loglik_grad_batch */
int gpcsd_dist_loglik_grad_batch(gpcsd_dist *dist, const gpcsd_hparams *hps, int nsets, double *out2, double *grad, int ngrad,
                                 int *status);
/* predict() over all trials  gpcsd1d.py:248-293 / gpcsd2d.py:289-334: outputs as gpcsd_predict, assembled from the devices' blocks */
int gpcsd_dist_predict(gpcsd_dist *dist, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                       int type, double *csd_list, double *csd, double *lfp_list, double *lfp);

/* ---- measurement --------------------------------------------------------------- */
/* When enabled, every launch of a named hot kernel (or kernel family) is bracketed by hipEvents on the stream it runs on.
 * on = 0 off; 1 fenced: every fused call synchronises, asynchronous / paired calls are evaluated one by one, chains run
 * eagerly; 2 asynchronous: events only, queued and paired calls stay queued and paired (what bench.py's timed step does),
 * chains run eagerly so the scopes inside them see their kernels; 3 as 2 with the chains replayed as hipGraphs (only the
 * scopes around whole chains and the GEMMs record).  gpcsd_prof_get waits for the recorded events. */
int gpcsd_prof_enable(gpcsd_ctx *ctx, int on);
/* Modes 2 / 3: duration (ms) of the last launch of the single-workgroup tridiagonalisation tail -- the kernel with the
 * largest share of GPU time -- in a chain (region 0: temporal, 1: spatial, 2: other), from wall-clock stamps its workgroups
 * write themselves: the one way to time it inside a replayed hipGraph.  Valid once that chain has finished. */
int gpcsd_prof_tail_clock(gpcsd_ctx *ctx, int region, double *ms, int *nwg, double *flops);
int gpcsd_prof_reset(gpcsd_ctx *ctx);
/* total ms, launch count and algorithmic flops accumulated under `name`; rc -2 if unknown */
int gpcsd_prof_get(gpcsd_ctx *ctx, const char *name, double *ms, long *count, double *flops);
/* names, ';'-separated, into buf */
int gpcsd_prof_names(gpcsd_ctx *ctx, char *buf, int buflen);
/* back-to-back v_mfma_f64_16x16x4_f64 microbenchmark: measured TFLOP/s */
int gpcsd_mfma_f64_peak(gpcsd_ctx *ctx, double *tflops);
/* average ms per launch of the fp64 MFMA GEMM on device-resident pseudo-random operands; cfg 0 = automatic tile
 * configuration, 1..6 = forced (tuning aid) */
int gpcsd_gemm_bench(gpcsd_ctx *ctx, int transA, int transB, int M, int N, int K, int cfg, int reps, double *ms_out);
/* average ms of the blocked Cholesky (numpy.linalg.cholesky of gpcsd1d.py:303-304 / gpcsd2d.py:343-350; the dense cross-check's
 * factor at N = nx * nt) on a device-resident, device-generated SPD test matrix of order n; events on the call's own stream
 * around the factorisation alone.  Kernel split through the profiling scopes (potrf_*). */
int gpcsd_potrf_bench(gpcsd_ctx *ctx, int n, int reps, double *ms_out);
/* how many of the factorisation's gate launches (the one-wave launches that hold the trailing update back until the next diagonal
 * block's workgroup is resident, numpy.linalg.cholesky at gpcsd1d.py:303-304) gave up waiting since the context was created:
 * the gates steer the order of execution only, so this is a performance counter, never an error */
int gpcsd_potrf_gate_timeouts(gpcsd_ctx *ctx, long *count);
/* phase split (us) of the one-workgroup 128 x 128 factor + invert launch inside that Cholesky: {load, serial panels, rank-16
 * MFMA updates, store L, diagonal-block inverses, doubling levels 16 / 32 / 64, store X, total} -- a tuning aid */
int gpcsd_potrf_diag_probe(gpcsd_ctx *ctx, double *out10);
/* streaming copy microbenchmark over `bytes` bytes: measured GB/s (read+write counted) */
int gpcsd_hbm_copy_peak(gpcsd_ctx *ctx, long bytes, double *gbs);

#ifdef __cplusplus
}
#endif
#endif /* GPCSD_HIP_H */
