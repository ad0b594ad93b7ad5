// Internal context of libgpcsd_hip.so: device, stream, named device buffers, profiling events.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gpcsd_hip.h"

namespace gpcsd {

struct HipError {
    int code;
    std::string msg;
};

#define GP_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            char _b[512];                                                                     \
            snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                     hipGetErrorString(_e));                                                  \
            throw gpcsd::HipError{-100 - (int)_e, _b};                                        \
        }                                                                                     \
    } while (0)

#define GP_REQUIRE(cond, code, ...)                                                           \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            char _b[512];                                                                     \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                                            \
            throw gpcsd::HipError{(code), _b};                                                \
        }                                                                                     \
    } while (0)

struct ProfEntry {
    double ms = 0.0;
    long count = 0;
    double flops = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<double> pending_flops;
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// An involutive permutation P with P K P = K, as orbit tables on the device: orbit a < ns has representatives
// (rep_i[a], rep_j[a]); the na pairs come first, fixed points have rep_i == rep_j; orb[row] = orbit of a row,
// sgn[row] = 0 (fixed point), +1 (first of a pair), -1 (second of a pair).  ns == 0: no symmetry known.
struct SymDev {
    int ns = 0, na = 0;
    const int *rep_i = nullptr, *rep_j = nullptr, *orb = nullptr, *sgn = nullptr;
};

}  // namespace gpcsd

// hipFuncSetAttribute applies to the CURRENT device's copy of a kernel: raise an attribute once per (call site, device), not once
// per process (a second device of the same process would otherwise launch with the default 64 KB of dynamic LDS)
namespace gpcsd {
struct PerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 63;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};
}  // namespace gpcsd

struct gpcsd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // temporal chain (Kt, its eigen-decomposition)
    hipStream_t stream3 = nullptr;          // spatial chain (Ks assembly, its eigen-decomposition)
    hipStream_t stream4 = nullptr;          // stage 3 / stage 5 of a staged temporal chain; the chunked copy-out of gpcsd_predict
    hipStream_t stream5 = nullptr;          // predict: what needs no decomposition (paired call) and the small Pcat products, beside
                                            // the large GEMMs of the main stream
    // Scalars and status words of the fused calls share one device allocation ("scal_status") and travel in one copy:
    // SCAL_N doubles, then STATUS_N ints.  Words [0..3]: the spatial chain ([0], second replica [2]) and the temporal chain
    // ([1], [3]) up to and including what a log-likelihood in the tridiagonal form consumes (stages 1 and 3 of a staged chain,
    // everything of an unstaged one).  Words [4..7] ("late"): stages 2 and 4 of a staged temporal chain (divide & conquer,
    // back-transformation), which such a log-likelihood neither reads nor waits for -- only calls that join the whole chain
    // (ev_join) collect them, and the next staged chain clears them on its own stream when nothing asynchronous is pending.
    static constexpr int SCAL_N = 64, STATUS_N = 8, STATUS_LATE = 4;
    static constexpr int RESULT_DOUBLES = SCAL_N + STATUS_N / 2;
    double *h_result = nullptr;             // pinned host landing zone for the end-of-call copy (RESULT_DOUBLES)
    bool capturing = false;                 // inside a stream capture: profiling scopes stay silent
    hipEvent_t ev_join = nullptr, ev_sjoin = nullptr;      // temporal / spatial chain finished (recorded on stream2 / stream3)
    // Software pipeline across calls.  The outputs of a side's decomposition (eigenvector blocks, spectra) exist in two
    // generations, used alternately (par[side] = the current one, 0 spatial / 1 temporal): the chain of call N+1 writes the
    // buffers call N-1 read, so it does not have to wait for call N's GEMM tail, which still reads generation N.  What it does
    // have to wait for is every reader of generation N-1; those were all queued on the main stream before chain N was launched,
    // and ev_mark[side][N % 2] was recorded there at that moment.  Chain N+1 therefore waits for ev_mark[side][N % 2] and then
    // records ev_mark[side][(N+1) % 2].  Everything else a chain touches (Gram assembly scratch, solver workspaces) is
    // private to the chain's stream.  Synchronous calls gain nothing from this (the host waits for the end of the call), but
    // gpcsd_predict_resident returns with its GEMM tail in flight, and the next call's two chains then run beside that tail.
    // async_pending: the status words of such a call have not been collected yet -- the next synchronising call
    // (gpcsd_fetch, gpcsd_device_synchronize, any call that returns values) reports them.
    int par[2] = {0, 0};
    hipEvent_t ev_mark[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool async_pending = false;
    // gpcsd_loglik_parts_async / gpcsd_loglik_predict_async: a result lands in a slot of h_ll (pinned, RESULT_DOUBLES like
    // h_result) behind that slot's event; up to LL_SLOTS evaluations may be outstanding, gpcsd_loglik_parts_wait collects
    // them oldest first.  two: the quadratic form came back as two partial sums; done: evaluated at once (profiling on).
    static constexpr int LL_SLOTS = 4;
    // A queued call's arguments, kept until its result has been collected: a call whose pipelined stage 5 missed its
    // tridiagonalisation (status 7: a scheduling miss, not a numerical failure -- wy.hip, q_pipe_timeouts below) is evaluated
    // again, unpipelined, by the call that collects it.
    struct HpKeep {
        gpcsd_hparams hp{};
        std::vector<double> sig;
        void set(const gpcsd_hparams *h) {
            hp = *h;
            sig.assign(h->sig2n, h->sig2n + (h->n_sig2n > 0 ? h->n_sig2n : 0));
            hp.sig2n = sig.data();
        }
    };
    struct PredKeep {
        bool have = false, piped = false, lists = false;
        HpKeep hp;
        std::vector<double> z, ts;
        int nz = 0, nts = 0, type = 0;
    };
    struct LlSlot {
        hipEvent_t ev = nullptr;
        bool two = false, done = false;
        double out[2] = {0.0, 0.0};
        int rc = 0;
        bool piped = false;                 // queued while the Q pipeline was on: status 7 means "run me again"
        HpKeep hp;
        long pred_seq = -1;                 // the paired call's prediction (pred_seq at the time), -1: a log-likelihood alone
    };
    long pred_seq = 0;                      // predictions queued so far: only the LAST one owns the resident outputs
    PredKeep last_pred;
    long q_pipe_timeouts = 0;               // calls evaluated again because a gate of the pipelined stage 5 gave up
    unsigned long long q_gate_ticks = 20000000ull;   // a gate's patience in 100 MHz ticks (0.2 s; a tail takes < 1 ms).
                                            // GPCSD_QPIPE_GATE_TICKS=0: test aid, every gate gives up at once
    double *h_ll = nullptr;                 // LL_SLOTS x RESULT_DOUBLES
    LlSlot ll_slot[LL_SLOTS];
    int ll_head = 0, ll_count = 0;          // oldest outstanding slot, number outstanding
    hipEvent_t ev_aux = nullptr, ev_pc = nullptr;   // predict: small products of the tail on stream2 beside the large ones
    hipEvent_t ev_chol_a = nullptr, ev_chol_d = nullptr;   // potrf look-ahead (chol.hip): next panel updated / next diagonal block factored
    unsigned int *h_chol_flag = nullptr;    // DEVICE words: [0] stamped by the diagonal kernel when its workgroup is resident (potrf's
                                            // gates poll it, agent-scope atomics), [1] gates whose time ran out (gpcsd_potrf_gate_timeouts)
    unsigned int chol_token = 0;            // last token handed out
    // staged temporal chain (capi.hip front_half): ev_t1 = stage 1 (tridiagonalisation) done, recorded on stream2; ev_q = stage 3
    // (T factors and the orthogonal factor Q, on stream4 behind ev_t1) done -- the log-likelihood's tail needs nothing more of
    // that chain, its stage 4 waits for it too.  q_gen: the generation of the temporal solver slot whose Q / tridiagonal are in
    // the buffers (decomposition cache hits reuse them).
    hipEvent_t ev_q[2] = {nullptr, nullptr};
    long q_gen = -1;
    // Those stage-1 / stage-3 outputs (reflectors, T factors, the tridiagonal and its scale, Q: the class arenas of the temporal
    // halves) exist in TWO generations, used alternately by successive temporal chains (tgen = the current one: it selects the
    // arena tag set, eigh_fold_tags): chain N + 1 overwrites what chain N - 1 left, whose readers -- a log-likelihood's or a
    // prediction's tail in the tridiagonal form, which may sit late in a queued GEMM tail -- were queued a whole call earlier.
    // Single-buffered, the next step's tridiagonalisation waited for the previous prediction's solve (0.3 ms per step).
    int tgen = 0;
    hipEvent_t ev_t1 = nullptr;
    bool q_queued[2] = {false, false};      // a stage 3 has been queued on this generation since its last temporal chain started (staged_chain_guard)
    // ... and the last reader of those single-buffered stage-1 outputs (Q, the tridiagonal, its scale) on the main stream: the next
    // temporal chain must not overwrite them before it (a caller may queue several steps deep)
    hipEvent_t ev_tri_done[2] = {nullptr, nullptr};
    bool tri_reader_queued[2] = {false, false};
    std::string last_error;
    std::map<std::string, gpcsd::DevBuf> bufs;
    bool prof_on = false;
    // wall-clock stamps of the tridiagonalisation tail's workgroups (SytrdBatch::clk): three regions (temporal chain, spatial
    // chain, other) of TAIL_CLK_WGS (start, end) pairs in host-mapped memory, allocated when profiling mode 2 / 3 is switched on
    static constexpr int TAIL_CLK_WGS = 64;
    unsigned long long *tail_clk_host = nullptr, *tail_clk_dev = nullptr;
    int tail_clk_count[3] = {0, 0, 0};
    double tail_clk_flops[3] = {0.0, 0.0, 0.0};
    int prof_mode = 0;                      // gpcsd_prof_enable: 0 off, 1 fenced, 2 asynchronous (eager chains), 3 asynchronous (graph chains)
    std::map<std::string, gpcsd::ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, int> int_cache;
    long alloc_epoch = 0;
    struct GraphSlot {
        hipGraphExec_t exec = nullptr;
        long epoch = -1;          // alloc_epoch the executable graph was captured under
        long seen_epoch = -1;     // alloc_epoch after the last eager run of this key
    };
    std::map<std::string, GraphSlot> graphs;   // small host-side memo (e.g. which n a device-side plan table was built for)

    // resident problem
    int dim = 0;                            // 1 or 2 once geometry is set
    int nx = 0, nt = 0, ntrials = 0;        // lfp shape
    int geo_nx = 0, ngl1 = 0, ngl2 = 0;     // geometry
    int time_nt = 0;
    double *d_lfp = nullptr;                // [x][r][t]
    gpcsd::SymDev sym_s, sym_t;             // reflection symmetry of the electrode / time grids (ns == 0: none found)
    // folded-basis GEMMs (capi.hip): what the electrode symmetry reflects about, host copies of the grids to recognise
    // prediction sites / times with the same symmetry, the symmetry of the last prediction sites, and whether the
    // resident data has been folded for the current geometry
    double sym_s_ctr[2] = {0.0, 0.0};
    bool sym_s_refl[2] = {false, false};
    std::vector<double> geo_host, time_host, sym_z_pts;
    gpcsd::SymDev sym_z;
    int lfp_fold_sig = 0;                   // bit k set: the folded copy of the data for FoldMode::sig() == k is current (0: none is)
    bool status_zeroed = false;             // the fused calls' status words were cleared at the end of the previous call
    // Which eigensolver class arenas (by tag) currently hold a POSITIVE SEMI-DEFINITE matrix: set by the fills that know it (the
    // Gram fills k_psd_fold_fill / k_temporal_fold_fill write true through EigArenaView::psd), cleared whenever the library's own
    // scaling pass fills the arena from a caller's matrix.  Only such a class may take the tridiagonalisation's rank-revealing early
    // exit (sytrd_regtail.hpp): the claim comes from the producer of the matrix, not from how the arena happened to be filled.
    std::map<std::string, bool> arena_psd;
    bool claim_psd = false;                 // gpcsd_eigh_psd() in progress: the caller vouches that its matrix is positive semi-definite
    bool tail_early_exit = true;            // gpcsd_tail_early_exit() / GPCSD_TAIL_EARLY_EXIT=0: PSD classes may stop the tridiagonalisation early
    bool fault_stage2 = false;              // gpcsd_debug_fault_stage2(): test aid, the staged divide & conquer reports status 3
    bool gram_fp32 = false;                 // gpcsd_set_gram_precision(): Gram builders evaluate in float (cfg5 variant)
    bool fold_gemm_on = true;               // gpcsd_fold_gemm()
    int ll_tridiag_mode = 2;                // gpcsd_ll_tridiag(): log-likelihood in the basis U (x) Q -- 0 off, 1 on, 2 by size (capi.hip)
    long ll_tridiag_calls = 0;
    long fold_gemm_calls = 0;
    // Pipelined stage 3 (round 5; wy.hip: wy_q_pipeline): the register tail of a staged temporal chain publishes its progress panel by
    // panel; T factors and the finished columns of Q follow on stream4 WHILE the tail reduces the next panel, and the matching
    // columns of X = Y~ Q on the main stream behind an event per panel -- behind the tail only the last panel's share is left (T, Q
    // and X stood for 0.18 ms of a 0.95 ms cfg3 step).
    // q_pipe: the switch (gpcsd_q_pipeline() / GPCSD_Q_PIPE=0).  q_pipe_want: set by a caller around its front half -- it promises to
    // form X through loglik_tri_pre, where stage 5 is queued (EigState::pipe_pending).  pipe_req: set around the stage-1 call
    // (problem set-up, graph key).  q_pipe_x: what stage 5 hangs on every finished block of columns -- in / out (nx R rows of nt), the
    // parity blocks' first columns.
    // gpcsd_prefetch_pair (capi_fused.inl): the next paired call's front half, queued ahead of that call (PairPrefetch, owned here;
    // dropped by any other front half)
    void *pair_prefetch = nullptr;
    long pair_prefetch_queued = 0, pair_prefetch_taken = 0;
    bool q_pipe = true;
    bool q_pipe_want = false;
    int pipe_req = 0;
    long q_pipe_calls = 0;
    struct QPipeX {
        const double *in = nullptr;
        double *out = nullptr;
        int M = 0, ld = 0, c0[2] = {0, 0}, rep = 0;
    } q_pipe_x;
    hipEvent_t ev_t0 = nullptr;             // the temporal chain's inputs are in place (stream2, in front of stage 1): stage 5 starts behind it
    bool t1_wait_pending = false;           // stage 5 was queued: the main stream's first reader of d / e still has to wait for ev_t1
    hipEvent_t ev_stage[8] = {};            // stage 5: panel k's columns of Q are final (stream4) -> the main stream's product on them
    // the paired call's prediction builds what needs no decomposition (cross-covariances, prediction-time Grams: ~0.1 ms of small
    // launches) on stream5 instead of in front of the log-likelihood's products on the main stream
    hipEvent_t ev_prelude = nullptr;        // ... done (stream5)
    hipEvent_t ev_m1 = nullptr;             // the main stream's readers of the folded cross-covariances are done (any predict)
    bool m1_read_queued = false;
    int q_pipe_mask = -1;                   // stage 5: bit p set = the product on Q's finished columns is launched behind panel p
                                            // (always behind the last; a skipped panel's columns ride in the next launch); -1: default
    // gpcsd_predict (host outputs): the caller's arrays while the call runs.  The fused last product of a folded prediction is then
    // launched in chunks of site orbits and every chunk's finished output rows are copied out (stream4: the DMA engine) while the
    // next chunk computes -- predict_sink_done[which] tells gpcsd_predict that nothing is left to download (capi_fused.inl).
    struct PredSink {
        bool active = false;
        double *sum[2] = {nullptr, nullptr}, *list[2] = {nullptr, nullptr};       // [0] csd, [1] lfp
        bool done[2] = {false, false};
    } pred_sink;
    std::vector<hipEvent_t> pred_sink_events;
    bool pred_chunked = true;               // gpcsd_predict_chunked_copy() / GPCSD_PRED_CHUNKED=0
    long pred_chunked_calls = 0;
    std::map<const int *, std::vector<int>> sym_host;       // host copies (rep_i | rep_j) of the orbit tables, by device pointer
    bool pair_share_x = true;               // gpcsd_pair_share_x()
    // ... and ONE spatial decomposition when the two sets differ by their jitter only (every loglik -> predict pair: the reference
    // adds jitter I to Ks in loglik, gpcsd2d.py:139, and not in predict, :296): the eigenvectors of Ks + j I are those of Ks and the
    // spectrum is shifted by j, so the prediction takes the log-likelihood's eigenvectors, its spectrum minus j, and -- with X shared
    // as well -- its projected data W.  gpcsd_pair_share_s() / GPCSD_PAIR_SHARE_S=1.  OFF by default: 154 MB less traffic and one large
    // product less per cfg3 step, but the step measured 6-10 % SLOWER (the prediction's LDS-filling solve then starts 70 us earlier and
    // lands on the next step's spatial Gram assembly: DESIGN 4.13), and the pair no longer has the bits of its fenced calls.
    bool pair_share_s = false;
    int solve_pass = 0;                     // trials per pass of the next tridiagonal solve: 0 / 64 the one-pass form, 32 the two-per-CU form (gram.hip: k_tridiag_solve)
    long pair_shared_s_calls = 0;
    long pair_shared_x_calls = 0;           // paired calls whose prediction read the log-likelihood's X = Y~ Q (capi_fused.inl)
    // Decomposition cache (capi.hip::front_half): predict() right after loglik() / fit() with the same hyper-parameters
    // (neuropixels/fit_gpcsd2d.py:101-107) decomposes the very same Kt again, repeated predict() calls the same Ks as well.
    // The kernels are deterministic, so reusing what the previous call left in the context's buffers gives the same bits.
    // A side is reused when its key (hyper-parameters of that side, grids, precision, output form) matches and nothing has
    // run on that solver slot since (eig_gen counts every eigh_pair_device use of a slot).
    bool decomp_cache_on = true;
    long decomp_cache_hits = 0;
    long eig_gen[2] = {0, 0};
    std::vector<unsigned char> decomp_key[2];
    long decomp_gen[2] = {-1, -1};
    bool decomp_t_full = false;             // the cached temporal side went through all stages (spectrum and eigenvectors exist),
                                            // not only the tridiagonalisation + Q a tridiagonal-form consumer needs
    long grid_epoch = 0;                    // bumped by set_geometry / set_time / set_host_temporal_gram / set_gram_precision
    // user-defined temporal covariances (covariances.py:235-238: any object with compute_Kt): the caller evaluates the
    // Gram matrices on the host and hands them over (gpcsd_set_host_temporal_gram); the fused calls then upload them
    // instead of running the SE / Matern builders.  Copies: no host pointer outlives the setter.
    bool host_kt_on = false;
    std::vector<double> host_kt, host_kt_cross;   // (nt, nt) sum over components; (C, ntstar, nt) per component or empty
    int host_kt_nt = 0, host_kt_C = 0, host_kt_ntstar = 0;
    std::vector<double> host_dkt;           // (nmat, nt, nt): d Kt / d theta_k from the objects' compute_dKt (gpcsd_set_host_temporal_dgram)
    int host_dkt_n = 0;

    // ---- device buffers: grow-only, keyed by name, freed in destroy ----
    template <typename T = double>
    T *buf(const std::string &name, size_t count) {
        size_t bytes = count * sizeof(T);
        if (bytes == 0) bytes = sizeof(T);
        gpcsd::DevBuf &b = bufs[name];
        if (b.bytes < bytes) {
            if (b.p) GP_HIP(hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
            GP_HIP(hipMalloc(&b.p, bytes));
            b.bytes = bytes;
            ++alloc_epoch;                       // captured graphs hold raw pointers: any (re)allocation retires them
        }
        return reinterpret_cast<T *>(b.p);
    }
    // page-locked host blocks: grow-only, keyed by name, freed in destroy (results that come back in ONE true asynchronous copy: a
    // copy into pageable memory is staged and makes the host wait per call)
    std::map<std::string, std::pair<void *, size_t>> pinned_bufs;
    template <typename T = double>
    T *pinned(const std::string &name, size_t count) {
        const size_t bytes = std::max<size_t>(count * sizeof(T), sizeof(T));
        auto &b = pinned_bufs[name];
        if (b.second < bytes) {
            if (b.first) GP_HIP(hipHostFree(b.first));
            b.first = nullptr;
            b.second = 0;
            GP_HIP(hipHostMalloc(&b.first, bytes, hipHostMallocDefault));
            b.second = bytes;
        }
        return reinterpret_cast<T *>(b.first);
    }
    // ---- host <-> device copies never hand the CALLER'S pageable memory to the runtime ----
    // A hipMemcpy whose host side is pageable makes the runtime register the caller's pages with the driver (a "userptr" mapping)
    // and keep that registration in a cache.  When those pages are later unmapped, trimmed or migrated -- the NumPy array is freed,
    // its allocator gives the range back -- the driver's MMU notifier EVICTS EVERY QUEUE OF THE PROCESS and restores them one to
    // three timer ticks later: 9 / 19 / 29 ms in which no kernel of the process runs, somewhere inside a later step loop (DESIGN 6,
    // tools/stall_probe.py: 4 of 8 successive models stalled; 0 of 8 with the trials uploaded from page-locked memory).  Every
    // transfer whose host side is not page-locked therefore goes through two page-locked bounce blocks of the context: one host
    // memcpy per 4 MB chunk, overlapped with the previous chunk's DMA.
    static constexpr size_t BOUNCE_BYTES = 4u << 20;
    unsigned char *bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    bool bounce_busy[2] = {false, false};
    long bounced_bytes = 0;                 // bytes that took the bounce path (tests read it through gpcsd_bounce_stats)
    // the context's own page-locked blocks (results, staging, pool): no query needed for those
    bool own_pinned(const void *p) const {
        auto in = [p](const void *b, size_t n) { return b && p >= b && p < static_cast<const unsigned char *>(b) + n; };
        if (in(h_result, RESULT_DOUBLES * sizeof(double)) || in(h_ll, (size_t)LL_SLOTS * RESULT_DOUBLES * sizeof(double)) ||
            in(stage_ring, STAGE_SLOT * STAGE_SLOTS) || in(bounce[0], BOUNCE_BYTES) || in(bounce[1], BOUNCE_BYTES))
            return true;
        for (const auto &kv : pinned_bufs)
            if (in(kv.second.first, kv.second.second)) return true;
        return false;
    }
    static bool host_is_pinned(const void *p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) {
            (void)hipGetLastError();         // (plain malloc / mmap memory: "invalid value", not an error here)
            return false;
        }
        return a.type == hipMemoryTypeHost;
    }
    void bounce_init() {
        for (int k = 0; k < 2; ++k) {
            if (!bounce[k]) GP_HIP(hipHostMalloc(reinterpret_cast<void **>(&bounce[k]), BOUNCE_BYTES, hipHostMallocDefault));
            if (!bounce_ev[k]) GP_HIP(hipEventCreateWithFlags(&bounce_ev[k], hipEventDisableTiming));
        }
    }
    void bounce_wait(int k) {
        if (bounce_busy[k]) GP_HIP(hipEventSynchronize(bounce_ev[k]));
        bounce_busy[k] = false;
    }
    // host -> device on stream st; the host range may be reused when this returns (as with a pageable hipMemcpyAsync)
    void copy_in(void *dev, const void *host, size_t bytes, hipStream_t st) {
        if (bytes == 0) return;
        if (capturing || own_pinned(host) || host_is_pinned(host)) {   // (nothing is uploaded inside a stream capture; kept as it was if it ever is)
            GP_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
            return;
        }
        if (bytes <= STAGE_SLOT && st == stream) {      // (the ring of small slots is fenced against the main stream)
            GP_HIP(hipMemcpyAsync(dev, stage_small(host, bytes), bytes, hipMemcpyHostToDevice, st));
            return;
        }
        bounce_init();
        bounced_bytes += (long)bytes;
        int k = 0;
        for (size_t off = 0; off < bytes; off += BOUNCE_BYTES, k ^= 1) {
            const size_t n = std::min(BOUNCE_BYTES, bytes - off);
            bounce_wait(k);
            memcpy(bounce[k], static_cast<const unsigned char *>(host) + off, n);
            GP_HIP(hipMemcpyAsync(static_cast<unsigned char *>(dev) + off, bounce[k], n, hipMemcpyHostToDevice, st));
            GP_HIP(hipEventRecord(bounce_ev[k], st));
            bounce_busy[k] = true;
        }
    }
    // device -> host on stream st.  Page-locked destination: queued, the caller synchronises (as before).  Pageable destination:
    // through the bounce blocks, complete on return (what a pageable hipMemcpyAsync amounts to as well).
    void copy_out(void *host, const void *dev, size_t bytes, hipStream_t st) {
        if (bytes == 0) return;
        if (own_pinned(host) || host_is_pinned(host)) {
            GP_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
            return;
        }
        bounce_init();
        bounced_bytes += (long)bytes;
        size_t prev_off = 0, prev_n = 0;
        int k = 0, prev_k = -1;
        for (size_t off = 0; off < bytes; off += BOUNCE_BYTES, k ^= 1) {
            const size_t n = std::min(BOUNCE_BYTES, bytes - off);
            bounce_wait(k);
            GP_HIP(hipMemcpyAsync(bounce[k], static_cast<const unsigned char *>(dev) + off, n, hipMemcpyDeviceToHost, st));
            GP_HIP(hipEventRecord(bounce_ev[k], st));
            bounce_busy[k] = true;
            if (prev_k >= 0) {
                bounce_wait(prev_k);
                memcpy(static_cast<unsigned char *>(host) + prev_off, bounce[prev_k], prev_n);
            }
            prev_k = k; prev_off = off; prev_n = n;
        }
        bounce_wait(prev_k);
        memcpy(static_cast<unsigned char *>(host) + prev_off, bounce[prev_k], prev_n);
    }
    template <typename T = double>
    T *upload(const std::string &name, const T *host, size_t count) {
        T *d = buf<T>(name, count);
        if (count) copy_in(d, host, count * sizeof(T), stream);
        if (!upload_shadow.empty()) upload_shadow.erase(name);       // a plain upload makes any cached image of this buffer stale
        return d;
    }
    // upload that is skipped when the named device buffer already holds exactly these bytes (hyper-parameter vectors,
    // prediction sites and times repeat from call to call; a pageable host-to-device copy costs ~10 us of host time)
    std::map<std::string, std::vector<unsigned char>> upload_shadow;
    long upload_count = 0;                  // uploads upload_cached() really queued (on the main stream)
    template <typename T = double>
    T *upload_cached(const std::string &name, const T *host, size_t count) {
        const size_t bytes = count * sizeof(T);
        std::vector<unsigned char> &sh = upload_shadow[name];
        const long epoch_before = alloc_epoch;
        T *d = buf<T>(name, count);
        if (alloc_epoch == epoch_before && sh.size() == bytes && bytes > 0 && memcmp(sh.data(), host, bytes) == 0) return d;
        if (count) copy_in(d, host, bytes, stream);
        ++upload_count;
        sh.assign(reinterpret_cast<const unsigned char *>(host), reinterpret_cast<const unsigned char *>(host) + bytes);
        return d;
    }
    // Small uploads go through a ring of pinned slots: a host-to-device copy from pageable memory makes the host wait for
    // everything queued on the stream before it, which would serialise a caller that changes hyper-parameters every call
    // against the previous call's GEMM tail.  The ring is fenced once per lap (every STAGE_SLOTS small uploads).
    static constexpr size_t STAGE_SLOT = 8192, STAGE_SLOTS = 128;
    unsigned char *stage_ring = nullptr;
    size_t stage_next = 0;
    const void *stage_small(const void *host, size_t bytes) {
        if (bytes > STAGE_SLOT) return host;
        if (!stage_ring) {
            if (hipHostMalloc(reinterpret_cast<void **>(&stage_ring), STAGE_SLOT * STAGE_SLOTS, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                stage_ring = nullptr;
                return host;
            }
        }
        if (stage_next == STAGE_SLOTS) {
            GP_HIP(hipStreamSynchronize(stream));
            stage_next = 0;
        }
        unsigned char *slot = stage_ring + STAGE_SLOT * stage_next++;
        memcpy(slot, host, bytes);
        return slot;
    }
    void download(void *host, const void *dev, size_t bytes) { copy_out(host, dev, bytes, stream); }
    void sync() { GP_HIP(hipStreamSynchronize(stream)); }

    hipEvent_t get_event() {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        GP_HIP(hipEventCreate(&e));
        return e;
    }
    void prof_collect();

    // Stream timeline (GPCSD_TIMELINE=1, a measurement aid): timestamped events at the phase boundaries of the fused calls on
    // the streams they run on -- unlike the profiling scopes it leaves the asynchronous behaviour alone, unlike a profiler's
    // kernel trace it adds no host time per launch.  Dumped (relative times, stderr) by gpcsd_device_synchronize.
    bool timeline_on = false;
    std::vector<std::pair<std::string, hipEvent_t>> timeline;
    void tl(const char *label, hipStream_t s) {
        if (!timeline_on || capturing) return;
        hipEvent_t e = get_event();
        (void)hipEventRecord(e, s);
        timeline.emplace_back(label, e);
    }
    void timeline_dump();
};

namespace gpcsd {

// RAII scope that brackets kernel launches on a stream with events when profiling is on.
struct ProfScope {
    gpcsd_ctx *c;
    const char *name;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double flops;
    ProfScope(gpcsd_ctx *ctx, const char *nm, double fl = 0.0, hipStream_t st = nullptr)
        : c(ctx), name(nm), s(st ? st : ctx->stream), flops(fl) {
        if (c->prof_on && !c->capturing) {
            e0 = c->get_event();
            e1 = c->get_event();
            (void)hipEventRecord(e0, s);
        }
    }
    ~ProfScope() {
        if (c->prof_on && e0 && !c->capturing) {
            (void)hipEventRecord(e1, s);
            ProfEntry &p = c->prof[name];
            p.pending.emplace_back(e0, e1);
            p.pending_flops.push_back(flops);
        }
    }
};

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace gpcsd
