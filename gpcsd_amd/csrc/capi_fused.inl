// Part of capi.hip (included there: one translation unit, so the file-local helpers of capi.hip are in scope) --
// fused hot calls on resident data: loglik (synchronous, queued), predict (folded / full-size, resident), the paired call, sample_prior.

// ------------------------------------------------------------------------------------------------
// fused hot calls
// ------------------------------------------------------------------------------------------------
// End of an asynchronous loglik: the scalars and status words go to the pinned block behind an event; nothing is waited for
// and the status words are left alone (the chains of later calls may already be reporting into them).
// the pinned slot the next asynchronous log-likelihood lands in (the tail's last launch may write it itself: k_ll_tridiag)
static double *next_ll_slot(gpcsd_ctx *c) { return c->h_ll + gpcsd_ctx::RESULT_DOUBLES * ((c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS); }

static int finish_loglik_async(gpcsd_ctx *c, const EigState &e, bool two, bool slot_written = false) {
    const int k = (c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS;       // callers have checked that a slot is free
    gpcsd_ctx::LlSlot &sl = c->ll_slot[k];
    if (!slot_written)
        GP_HIP(hipMemcpyAsync(c->h_ll + gpcsd_ctx::RESULT_DOUBLES * k, e.scal, gpcsd_ctx::RESULT_DOUBLES * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    GP_HIP(hipEventRecord(sl.ev, c->stream));
    c->tl("loglik result copied (main)", c->stream);
    ++c->ll_count;
    sl.done = false;
    sl.two = two;
    c->async_pending = true;
    c->status_zeroed = false;
    return 0;
}

// Folded-basis tail of the log-likelihood: the two projections as 2 + 2 half-size GEMMs, the quadratic form as two partial
// sums (one when the parity blocks went out as one batched launch: returns true), sum(log D) folded into the reduce launch.
// Shifted-tridiagonal form (EigState::tri): X = Y~ Q needs stage 1 of the temporal chain only and is queued BEFORE the main
// stream waits for the spatial chain; after that wait W = diag(U)^T X, and one forward recurrence per row gives the quadratic
// form, the pivots the log-determinant (k_ll_tridiag).  The temporal eigenvectors are never read.
// Order of the log-likelihood's two products: X = Y~ Q then W = U^T X (0: the temporal product first, behind Q and in front of the
// wait for the spatial chain), or W0 = U^T Y~ then W = W0 Q (1: the spatial product first, behind the spatial chain and in front
// of the wait for Q).  Same flops, same shapes; whichever chain ends first should have its product first.  GPCSD_LL_ORDER.
// (ll_order(): capi.hip, beside q_pipe_applies)

// out[(x, r)][t~ block p] = in[(x, r)][t~ block p] Q_p, both parity blocks in one launch (Q of replica e.tri_rep; waits for stage 3)
static void tri_times_Q(gpcsd_ctx *c, EigState &e, const FoldMode &fm, const double *in, double *out, const char *prof) {
    const int nx = c->nx, nt = c->nt, R = c->ntrials;
    hipStream_t s = c->stream;
    if (e.wait_q) GP_HIP(hipStreamWaitEvent(s, c->ev_q[c->tgen], 0));
    e.wait_q = false;
    const char *const *tg = eigh_fold_tags(c, 1);
    GemmDesc g[2];
    for (int p = 0; p < 2; ++p) {
        const int np = p ? fm.ft.na : fm.ft.ns, c0 = p ? fm.ft.ns : 0;
        g[p].M = nx * R; g[p].N = np; g[p].K = np;
        g[p].A = in + c0; g[p].lda = nt;
        g[p].B = eigh_Q_view(c, tg[p], np, e.tri_count) + (size_t)e.tri_rep * np * np; g[p].ldb = np;
        g[p].C = out + c0; g[p].ldc = nt;
        g[p].prof_name = prof;
    }
    gemm_pair(c, g[0], g[1], s);
}

static void loglik_tri_pre(gpcsd_ctx *c, EigState &e, const FoldMode &fm, const double *Yf, const char *xname = "ll_X",
                           const char *prof = "gemm_ll_YQ", bool is_ll = true) {
    if (is_ll && ll_order() == 1) return;              // spatial product first: everything happens in the tail
    if (e.pipe_pending) {                              // stage 1 ran with progress words: T, Q and X panel by panel (queue_q_pipeline)
        queue_q_pipeline(c, e, Yf, xname);
        return;
    }
    tri_times_Q(c, e, fm, Yf, c->buf<double>(xname, (size_t)c->nx * c->ntrials * c->nt), prof);
}

// host_slot: the caller's pinned result slot when the evaluation is asynchronous (else null); returns true when the tail wrote it
static bool loglik_tri_tail(gpcsd_ctx *c, EigState &e, const FoldMode &fm, const double *Yf, double *W, double *host_slot = nullptr) {
    const int nx = c->nx, nt = c->nt, R = c->ntrials;
    hipStream_t s = c->stream;
    ++c->fold_gemm_calls;
    ++c->ll_tridiag_calls;
    join_spatial(c, e);
    double *X = c->buf<double>("ll_X", (size_t)nx * R * nt);
    if (ll_order() == 1) {
        fold_proj_spatial(c, fm.fs, Yf, X, (long)R * nt, s);
        tri_times_Q(c, e, fm, X, W, "gemm_ll_WQ");
    } else {
        fold_proj_spatial(c, fm.fs, X, W, (long)R * nt, s);
    }
    const char *const *tg = eigh_fold_tags(c, 1);
    const double *d[2], *ee[2], *am[2];
    int np[2], c0[2];
    for (int p = 0; p < 2; ++p) {
        np[p] = p ? fm.ft.na : fm.ft.ns;
        c0[p] = p ? fm.ft.ns : 0;
        const EigArenaView av = eigh_arena_view(c, tg[p], np[p], e.tri_count);
        const long o = (long)e.tri_rep * av.blk;
        d[p] = av.d + o; ee[p] = av.e + o; am[p] = av.amax + o;
    }
    if (c->t1_wait_pending) {              // stage 5: nothing on this stream is ordered behind the END of stage 1 yet (d, e, the scale)
        GP_HIP(hipStreamWaitEvent(s, c->ev_t1, 0));
        c->t1_wait_pending = false;
    }
    const bool wrote = k_ll_tridiag(c, W, fm.fs.w, d, ee, am, e.d_sig, nx, R, nt, np, c0, e.scal, e.scal + 1, s, host_slot,
                                    e.scal + gpcsd_ctx::SCAL_N, gpcsd_ctx::SCAL_N, gpcsd_ctx::RESULT_DOUBLES - gpcsd_ctx::SCAL_N);
    GP_HIP(hipEventRecord(c->ev_tri_done[c->tgen], s));
    c->tri_reader_queued[c->tgen] = true;
    return wrote;
}

static bool loglik_fold_tail(gpcsd_ctx *c, EigState &e, const FoldMode &fm, const double *Yf, double *W, double *host_slot = nullptr,
                             bool *slot_written = nullptr) {
    const int nx = c->nx, nt = c->nt, R = c->ntrials;
    hipStream_t s = c->stream;
    if (e.tri) {
        loglik_tri_pre(c, e, fm, Yf);
        const bool wrote = loglik_tri_tail(c, e, fm, Yf, W, host_slot);
        if (slot_written) *slot_written = wrote;
        return true;
    }
    ++c->fold_gemm_calls;
    join_spatial(c, e);
    fold_proj_spatial(c, fm.fs, Yf, W, (long)R * nt, s);
    const int nparts = join_temporal(c, e, &fm, false);         // sum(log D): summed by the reduce launch of the GEMM below
    GemmDesc g2[2];
    g2[0].extra_sum_in = c->buf<double>("buildD_partials", 256);
    g2[0].extra_sum_n = nparts;
    g2[0].extra_sum_out = e.scal;
    for (int p = 0; p < 2; ++p) {
        const int np = p ? fm.ft.na : fm.ft.ns, c0 = p ? fm.ft.ns : 0;
        g2[p].M = nx * R; g2[p].N = np; g2[p].K = np;
        g2[p].A = W + c0; g2[p].lda = nt;
        g2[p].B = fm.ft.U + (p ? (size_t)fm.ft.ns * fm.ft.ns : 0); g2[p].ldb = np;
        g2[p].epi = EPI_QUAD; g2[p].D = e.Dinv + c0; g2[p].rdiv = R; g2[p].ldd = nt; g2[p].quad_out = e.scal + 1 + p;
        g2[p].prof_name = "gemm_proj_temporal_quad";
    }
    return gemm_pair(c, g2[0], g2[1], s);
}

static int loglik_parts_impl(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out2, bool async) {
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    if (async && c->prof_mode == 1) {     // fenced profiling (mode 1): evaluate now, hand the result over at the wait
        gpcsd_ctx::LlSlot &sl = c->ll_slot[(c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS];
        sl.rc = loglik_parts_impl(c, hp, sl.out, false);
        sl.done = true;
        ++c->ll_count;
        return 0;
    }
    const FoldMode fm0 = fold_mode(c, hp);                          // the decision; its views are of the previous generation
    const double *Yf = fm0.on ? folded_lfp(c, fm0) : nullptr;
    // (the spatial chain is joined by the tail: the tridiagonal form queues its first product in front of that wait)
    c->q_pipe_want = fm0.on && fm0.ft.on;                           // (X = Y~ Q goes through loglik_tri_pre below: stage 5 may apply)
    EigState e;
    try {
        e = front_half(c, hp, hp->jitter, !fm0.on, /*join_s=*/!fm0.on, /*want_tri=*/fm0.on);
    } catch (...) {
        c->q_pipe_want = false;
        throw;
    }
    c->q_pipe_want = false;
    const FoldMode fm = fold_mode(c, hp);                           // views of the generation the front half just launched
    const int nx = c->nx, nt = c->nt, R = c->ntrials;
    hipStream_t s = c->stream;
    double *W = c->buf<double>("proj_W", (size_t)nx * R * nt);
    if (fm.on) {
        bool wrote = false;
        const bool batched = loglik_fold_tail(c, e, fm, Yf, W, async ? next_ll_slot(c) : nullptr, &wrote);
        if (async) return finish_loglik_async(c, e, !batched, wrote);
        double h3[3] = {0.0, 0.0, 0.0};
        const int rc = finish_call(c, e, h3, 3);
        out2[0] = h3[0];
        out2[1] = batched ? h3[1] : h3[1] + h3[2];
        return rc;
    }
    GemmDesc g1;                          // W[x'][(r,t)] = sum_x Qs[x][x'] Y[x][(r,t)]        (gpcsd1d.py:125 inner dot)
    g1.M = nx; g1.N = R * nt; g1.K = nx;
    g1.A = e.Qs; g1.lda = nx; g1.transA = true;
    g1.B = c->d_lfp; g1.ldb = (long)R * nt;
    g1.C = W; g1.ldc = (long)R * nt;
    g1.prof_name = "gemm_proj_spatial";
    gemm_f64(c, g1, s);
    join_temporal(c, e);
    GemmDesc g2;                          // alpha[(x',r)][i'] = sum_t W[(x',r)][t] Qt[t][i'];  quad = sum alpha^2 / D
    g2.M = nx * R; g2.N = nt; g2.K = nt;
    g2.A = W; g2.lda = nt;
    g2.B = e.Qt; g2.ldb = nt;
    g2.epi = EPI_QUAD; g2.D = e.Dinv; g2.rdiv = R; g2.ldd = nt; g2.quad_out = e.scal + 1;
    g2.prof_name = "gemm_proj_temporal_quad";
    gemm_f64(c, g2, s);
    if (async) return finish_loglik_async(c, e, false);
    return finish_call(c, e, out2, 2);
}

// Status 7 is not a numerical failure: a gate of the pipelined stage 5 (wy.hip: wy_qstage_kernel) gave up waiting for the
// tridiagonalisation that should have been running beside it -- kernels serialised by a profiler or a debugger, an oversubscribed
// card, a tail that could not get a CU.  The launch did nothing with the unfinished reflectors, so the evaluation is void, not
// wrong.  The call that collects it switches the pipeline off for this context (latched: the conditions that made one gate miss
// make the next one miss), counts it (gpcsd_q_pipeline_stats) and evaluates again: behind the END of the tail queue_stage5_plain
// forms the same T, Q and X.  `piped`: the call was queued while the pipeline was on.
static bool q_pipe_missed(gpcsd_ctx *c, int rc, bool piped) {
    if (rc != 7 || !piped) return false;
    c->q_pipe = false;
    ++c->q_pipe_timeouts;
    drain_after_failure(c);                // (also drops an announced front half: it was queued with the pipeline on)
    if (int *dst = reinterpret_cast<int *>(c->buf<double>("scal_status", gpcsd_ctx::RESULT_DOUBLES) + gpcsd_ctx::SCAL_N)) {
        GP_HIP(hipMemsetAsync(dst, 0, gpcsd_ctx::STATUS_N * sizeof(int), c->stream));
        GP_HIP(hipStreamSynchronize(c->stream));
        c->status_zeroed = true;
    }
    c->last_error.clear();
    return true;
}

static void keep_prediction(gpcsd_ctx *c, bool piped, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                            int type, bool want_lists) {
    gpcsd_ctx::PredKeep &k = c->last_pred;
    ++c->pred_seq;
    k.have = true;
    k.piped = piped;
    k.hp.set(hp);
    k.z.assign(z, z + (size_t)nz * c->dim);
    k.ts.assign(tstar, tstar + ntstar);
    k.nz = nz; k.nts = ntstar; k.type = type; k.lists = want_lists;
}

extern "C" int gpcsd_loglik_parts(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out2) {
    GP_API_BEGIN(c)
    GP_REQUIRE(out2 != nullptr, -3, "null output");
    const bool piped = c->q_pipe;
    int rc = loglik_parts_impl(c, hp, out2, false);
    if (q_pipe_missed(c, rc, piped)) rc = loglik_parts_impl(c, hp, out2, false);
    return rc;
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_parts_async(gpcsd_ctx *c, const gpcsd_hparams *hp) {
    if (c && c->ll_count >= gpcsd_ctx::LL_SLOTS)   // refused before anything is touched: the outstanding ones stay collectable
        return fail(c, HipError{-3, "loglik_parts_async: too many asynchronous evaluations outstanding (collect with "
                                    "gpcsd_loglik_parts_wait)"});
    GP_API_BEGIN(c)
    gpcsd_ctx::LlSlot &sl = c->ll_slot[(c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS];
    sl.piped = c->q_pipe;
    sl.pred_seq = -1;
    if (hp) sl.hp.set(hp);
    return loglik_parts_impl(c, hp, nullptr, true);
    GP_API_END(c)
}

static int predict_impl(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                        int type, bool want_lists, bool async);

extern "C" int gpcsd_loglik_parts_wait(gpcsd_ctx *c, double *out2) {
    if (c && (!out2 || c->ll_count == 0))
        return fail(c, HipError{-3, out2 ? "loglik_parts_wait: no asynchronous evaluation pending" : "null output"});
    GP_API_BEGIN(c)
    const int k = c->ll_head;
    gpcsd_ctx::LlSlot &sl = c->ll_slot[k];
    c->ll_head = (k + 1) % gpcsd_ctx::LL_SLOTS;
    --c->ll_count;
    if (sl.done) {
        out2[0] = sl.out[0];
        out2[1] = sl.out[1];
        return sl.rc;
    }
    {
        // GPCSD_LL_WAIT=spin: poll the event instead of blocking in hipEventSynchronize (A/B for DESIGN 6's stall: does a blocked
        // host wait miss its wake-up?)
        static const bool spin = getenv("GPCSD_LL_WAIT") && !strcmp(getenv("GPCSD_LL_WAIT"), "spin");
        if (spin) {
            hipError_t q;
            while ((q = hipEventQuery(sl.ev)) == hipErrorNotReady) {}
            if (q != hipSuccess) GP_HIP(q);
        } else {
            GP_HIP(hipEventSynchronize(sl.ev));
        }
    }
    const double *host = c->h_ll + gpcsd_ctx::RESULT_DOUBLES * k;
    out2[0] = host[0];
    out2[1] = sl.two ? host[1] + host[2] : host[1];
    int st[gpcsd_ctx::STATUS_N];
    memcpy(st, host + gpcsd_ctx::SCAL_N, sizeof(st));
    // (asynchronous calls: every word as it stood when this evaluation's copy ran, the late stages' of EARLIER chains included)
    const int bad = fold_status(st, true);
    if (bad != 0) {                       // this evaluation's, or an earlier asynchronous call's that nobody collected yet
        char b[160];
        snprintf(b, sizeof(b), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", bad);
        c->last_error = b;
        // The status words are sticky while asynchronous work is outstanding (nobody may clear them under a running chain).
        // Now that a failure has been reported: drain everything and clear them, so that evaluations queued from here on
        // start clean.  Evaluations that were ALREADY outstanding copied the words as they stood and report the failure too
        // (a failed wait poisons the ones queued before it returned; documented in gpcsd_hip.h).
        if (q_pipe_missed(c, bad, sl.piped)) {
            // a scheduling miss of the pipelined stage 5, not a failure: this evaluation again (unpipelined now), and the paired
            // prediction with it when it is still the one that owns the resident outputs
            int rc = loglik_parts_impl(c, &sl.hp.hp, out2, false);
            if (rc == 0 && sl.pred_seq >= 0 && sl.pred_seq == c->pred_seq && c->last_pred.have) {
                const gpcsd_ctx::PredKeep &k = c->last_pred;
                rc = predict_impl(c, &k.hp.hp, k.z.data(), k.nz, k.ts.data(), k.nts, k.type, k.lists, false);
            }
            return rc;
        }
        drain_after_failure(c);
        if (int *dst = reinterpret_cast<int *>(c->buf<double>("scal_status", gpcsd_ctx::RESULT_DOUBLES) + gpcsd_ctx::SCAL_N)) {
            GP_HIP(hipMemsetAsync(dst, 0, gpcsd_ctx::STATUS_N * sizeof(int), c->stream));
            GP_HIP(hipStreamSynchronize(c->stream));
            c->status_zeroed = true;
        }
        return bad > 0 ? bad : 1;
    }
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_loglik(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out) {
    if (!out) return fail(c, HipError{-3, "null output"});
    double p[2] = {0.0, 0.0};
    int rc = gpcsd_loglik_parts(c, hp, p);
    if (rc < 0) return rc;
    *out = -0.5 * (double)c->ntrials * p[0] - 0.5 * p[1];      // gpcsd1d.py:122,127-128
    return rc;
}

// Reflection symmetry of the prediction sites under the SAME reflection as the electrodes (then the cross-covariances
// commute with the pair of involutions and fold as well).  Cached on the site coordinates; ns == 0: none.
static const SymDev &site_symmetry(gpcsd_ctx *c, const double *z, int nz, int dim) {
    const size_t cnt = (size_t)nz * dim;
    if (c->sym_z_pts.size() == cnt && memcmp(c->sym_z_pts.data(), z, cnt * sizeof(double)) == 0) return c->sym_z;
    c->sym_z_pts.assign(z, z + cnt);
    if (c->geo_host.size() == cnt && memcmp(c->geo_host.data(), z, cnt * sizeof(double)) == 0) c->sym_z = c->sym_s;
    else c->sym_z = find_symmetry(c, "sym_z_tbl", z, nz, dim, c->sym_s_ctr, c->sym_s_refl);
    return c->sym_z;
}

static bool pred_unfold_chunked(gpcsd_ctx *c, PredUnfoldDesc pu, int which0, int nz, size_t out_elems, bool want_lists, hipStream_t s);

// predict_impl in the folded basis (see FoldMode).  Prediction sites and times must share the symmetry of the grids:
//   out_c = Fz^T [ diag_p( (Kc_pp^T U_p) ) Bm~ diag_q( V_q^T Kt*_c,qq ) ] Ft   with Bm~ = (diag(U)^T Y~ diag(V)) / D~ ,
// every flat GEMM split in its two parity blocks; the last pass unfolds sites and times while it transposes.
static int predict_fold(gpcsd_ctx *c, const gpcsd_hparams *hp, EigState &e, const FoldMode &fm, const double *Yf, const SymDev &sz,
                        const double *dz, int nz, const double *dts, int type, bool want_lists, bool async,
                        const std::function<void()> *after_spatial_join = nullptr,
                        const std::function<void()> *before_spatial_join = nullptr, const char *shared_x = nullptr,
                        bool prelude_side = false, const char *shared_w = nullptr) {
    const Geo g = resident_geo(c);
    const int nx = c->nx, nt = c->nt, R = c->ntrials, C = hp->n_temporal;
    const long RT = (long)R * nt;
    const int ns = fm.fs.ns, na = fm.fs.na, nts = fm.ft.ns, nta = fm.ft.na, nzs = sz.ns, nza = sz.na;
    hipStream_t s = c->stream;
    // shared_w (the paired call with X and the spatial eigenvectors shared): W~ = diag(U)^T X is the log-likelihood's own product
    double *W = c->buf<double>(shared_w ? shared_w : "proj_W", (size_t)nx * RT);
    double *Bm = c->buf<double>("pred_B", (size_t)nx * RT);
    const double *t = (const double *)c->bufs["time_t"].p;
    double *Kc = c->buf<double>("pred_Kcross", (size_t)nx * nz);
    const size_t kcf_sz = (size_t)ns * nzs + (size_t)na * nza;
    double *Kcf = c->buf<double>("pred_Kcross_fold", 2 * kcf_sz);
    double *S = c->buf<double>("pred_S", (size_t)nz * RT);
    // comp~ and Pcat keep every (parity, component) block of columns on a 128-byte boundary (block widths padded to a multiple
    // of 16 doubles): the final relayout pass reads comp~ in 16-column pieces per trial row, and unaligned blocks (250 columns)
    // made every piece straddle two cache lines -- 291 MB fetched for 154 MB of comp~ per cfg3 step
    const int ntsP = (nts + 15) & ~15, ntaP = (nta + 15) & ~15;
    // (+16: a row stride that is a power of two -- 1024 doubles at nt = 500 -- walks the same HBM channels row after row)
    const long ldcomp = (long)C * (ntsP + ntaP) + 16;
    double *comp = c->buf<double>("pred_comp", std::max((size_t)C * nz * RT, (size_t)nz * R * ldcomp));
    double *Kts = c->buf<double>("pred_Ktstar", (size_t)C * nt * nt);
    const size_t ktf_sz = (size_t)nts * nts + (size_t)nta * nta;
    double *Ktf = c->buf<double>("pred_Ktstar_fold", (size_t)C * ktf_sz);
    const size_t m1_sz = (size_t)nzs * ns + (size_t)nza * na;
    double *M1 = c->buf<double>("pred_M1", 2 * std::max(m1_sz, (size_t)nz * nx));
    const size_t pc_s = (size_t)nts * C * ntsP;                     // Pcat_sym: nts rows of C * ntsP columns; Pcat_anti follows
    double *Pc = c->buf<double>("pred_Pc", std::max(pc_s + (size_t)nta * C * ntaP, (size_t)C * nt * nt));
    const size_t out_elems = (size_t)nz * RT;
    ++c->fold_gemm_calls;
    // what needs neither decomposition runs first, beside both chains: the cross-covariances and the prediction-time Grams,
    // folded
    // (prelude_side -- the paired call: on stream5, behind the previous call's side products there; the main stream is still busy
    // with the previous prediction and has the log-likelihood's products to queue first)
    hipStream_t sp = prelude_side ? c->stream5 : s;
    if (prelude_side && c->m1_read_queued) GP_HIP(hipStreamWaitEvent(sp, c->ev_m1, 0));     // the last reader of Kc~ on the main stream
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        double *kf = Kcf + (size_t)(which - 1) * kcf_sz;
        if (which == 1) build_kphig(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, Kc, sp);       // gpcsd1d.py:273
        else build_kphi(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, 0.0, Kc, sp);              // gpcsd1d.py:275
        k_sym_fold_rect(c, Kc, nz, fm.sym_s, sz, kf, kf + (size_t)ns * nzs, sp);
    }
    for (int cc = 0; cc < C; ++cc) {
        temporal_cross_gram(c, hp, cc, dts, nt, t, nt, Kts + (size_t)cc * nt * nt, sp);
        k_sym_fold_rect(c, Kts + (size_t)cc * nt * nt, nt, fm.sym_t, fm.sym_t, Ktf + cc * ktf_sz,
                        Ktf + cc * ktf_sz + (size_t)nts * nts, sp);
    }
    if (prelude_side) GP_HIP(hipEventRecord(c->ev_prelude, sp));
    // then everything that needs only the spatial eigenvectors, beside the temporal eigensolver
    if (before_spatial_join) (*before_spatial_join)();
    // tridiagonal form (EigState::tri): X = Y~ Q needs the temporal tridiagonalisation + Q only and is queued in front of the wait
    // for the spatial chain, exactly as the log-likelihood's
    // (a call of its own: in front of the wait for the spatial chain; in the paired call: behind the log-likelihood's tail -- in
    // front of it the product delays the result the caller is waiting for by its 0.09 ms)
    // shared_x (the paired call with equal temporal hyper-parameters): X = Y~ Q is the log-likelihood's own product -- the two
    // replicas of the temporal problem are the same matrix, decomposed by the same deterministic launches into the same bits, so
    // the second product would only recompute `shared_x` (0.1 ms of MFMA time and 154 MB of traffic per cfg3 step)
    const char *xname = shared_x ? shared_x : "pred_X";
    if (e.tri && !after_spatial_join && !shared_x) loglik_tri_pre(c, e, fm, Yf, "pred_X", "gemm_pred_YQ", false);
    join_spatial(c, e);
    // gpcsd_loglik_predict_async: the log-likelihood's whole tail goes here, in front of everything of predict that needs a
    // decomposition -- it is what the caller waits for
    if (after_spatial_join) (*after_spatial_join)();
    if (e.tri && after_spatial_join && !shared_x) loglik_tri_pre(c, e, fm, Yf, "pred_X", "gemm_pred_YQ", false);
    if (e.tri && shared_x) {              // (the wait for Q the skipped product would have queued; the log-likelihood's tail has passed it)
        if (e.wait_q) GP_HIP(hipStreamWaitEvent(s, c->ev_q[c->tgen], 0));
        e.wait_q = false;
    }
    if (!shared_w) fold_proj_spatial(c, fm.fs, e.tri ? c->buf<double>(xname, (size_t)nx * RT) : Yf, W, RT, s);   // W~ = diag(U)^T Y~ (or of Y~ Q)
    if (prelude_side) GP_HIP(hipStreamWaitEvent(s, c->ev_prelude, 0));
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        const double *kf = Kcf + (size_t)(which - 1) * kcf_sz;
        for (int p = 0; p < 2; ++p) {
            const int np = p ? na : ns, nzp = p ? nza : nzs;
            if (np == 0 || nzp == 0) continue;
            GemmDesc gm;                  // M1_p[zq][x'] = sum_xq Kc~_pp[xq][zq] U_p[xq][x']
            gm.M = nzp; gm.N = np; gm.K = np;
            gm.A = kf + (p ? (size_t)ns * nzs : 0); gm.lda = nzp; gm.transA = true;
            gm.B = fm.fs.U + (p ? (size_t)ns * ns : 0); gm.ldb = np;
            gm.C = M1 + (size_t)(which - 1) * m1_sz + (p ? (size_t)nzs * ns : 0); gm.ldc = np;
            gm.prof_name = "gemm_pred_M1";
            gemm_f64(c, gm, s);
        }
    }
    GP_HIP(hipEventRecord(c->ev_m1, s));
    c->m1_read_queued = true;
    // the temporal basis of everything below: the eigenvectors V_p (blocks of fm.ft.U), or in the tridiagonal form the orthogonal
    // factors Q_p of the tridiagonalisation (replica e.tri_rep of the temporal classes)
    const double *Tb[2] = {fm.ft.U, fm.ft.U + (size_t)nts * nts};
    if (e.tri) {
        const char *const *tg = eigh_fold_tags(c, 1);
        for (int p = 0; p < 2; ++p) {
            const int np = p ? nta : nts;
            Tb[p] = np > 0 ? eigh_Q_view(c, tg[p], np, e.tri_count) + (size_t)e.tri_rep * np * np : nullptr;
        }
    } else {
        join_temporal(c, e, &fm, false);  // predict never reads sum(log D)
    }
    GemmDesc g2[2];                       // Bm~[:, p block] = (W~[:, p block] V_p) / D~
    for (int p = 0; p < 2; ++p) {
        const int np = p ? nta : nts, c0 = p ? nts : 0;
        g2[p].M = nx * R; g2[p].N = np; g2[p].K = np;
        g2[p].A = W + c0; g2[p].lda = nt;
        g2[p].B = Tb[p]; g2[p].ldb = np;
        g2[p].C = Bm + c0; g2[p].ldc = nt;
        g2[p].epi = EPI_DIV_D; g2[p].D = e.Dinv + c0; g2[p].rdiv = R; g2[p].ldd = nt;
        g2[p].prof_name = "gemm_pred_temporal_div";
    }
    // Pcat = V^T Kt*~ needs the temporal eigenvectors and the folded prediction-time Grams only: it runs on a stream of its
    // own beside the large Bm~ / S~ products of the main stream instead of in front of them (two small launches off the
    // serial tail) -- not on stream2, where it would sit between this call's temporal chain and the next call's
    GP_HIP(hipEventRecord(c->ev_aux, s));                            // Kt*~ and the temporal eigenvectors are complete here
    GP_HIP(hipStreamWaitEvent(c->stream5, c->ev_aux, 0));
    for (int p = 0; p < 2; ++p) {
        const int np = p ? nta : nts, npP = p ? ntaP : ntsP;
        if (np == 0) continue;
        GemmDesc gp;                      // Pcat_p[i'][cc*npP + b] = sum_j V_p[j][i'] Kt*~_cc,pp[j][b], all components batched
        gp.M = np; gp.N = np; gp.K = np;
        gp.A = Tb[p]; gp.lda = np; gp.transA = true;
        gp.B = Ktf + (p ? (size_t)nts * nts : 0); gp.ldb = np;
        gp.C = Pc + (p ? pc_s : 0); gp.ldc = (long)C * npP;
        gp.batch = C; gp.sA = 0; gp.sB = (long)ktf_sz; gp.sC = npP;
        gp.prof_name = "gemm_pred_Pc";
        gemm_f64(c, gp, c->stream5);
    }
    GP_HIP(hipEventRecord(c->ev_pc, c->stream5));
    c->tl("Pc end (s5)", c->stream5);
    if (e.tri) {
        // Bm~ = the solutions of the shifted tridiagonal systems (es[x'] m T_p + sig2 I) b = w, row by row of W~ = diag(U)^T Y~ Q
        const char *const *tg = eigh_fold_tags(c, 1);
        const double *d[2], *ee[2], *am[2];
        int np[2], c0[2];
        for (int p = 0; p < 2; ++p) {
            np[p] = p ? nta : nts;
            c0[p] = p ? nts : 0;
            const EigArenaView av = eigh_arena_view(c, tg[p], np[p], e.tri_count);
            const long o = (long)e.tri_rep * av.blk;
            d[p] = av.d + o; ee[p] = av.e + o; am[p] = av.amax + o;
        }
        if (c->t1_wait_pending) {
            GP_HIP(hipStreamWaitEvent(s, c->ev_t1, 0));
            c->t1_wait_pending = false;
        }
        k_tridiag_solve(c, W, Bm, fm.fs.w, d, ee, am, e.d_sig, nx, R, nt, np, c0, s);
        GP_HIP(hipEventRecord(c->ev_tri_done[c->tgen], s));     // the last reader of Q / the tridiagonal on this stream
        c->tri_reader_queued[c->tgen] = true;
    } else {
        gemm_pair(c, g2[0], g2[1], s);
    }
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        double *o_sum = c->buf<double>(which == 1 ? "pred_out_csd" : "pred_out_lfp", out_elems);
        double *o_list = want_lists ? c->buf<double>(which == 1 ? "pred_out_csd_list" : "pred_out_lfp_list", out_elems * C)
                                    : nullptr;
        GemmDesc g5[2], g6[2];
        for (int p = 0; p < 2; ++p) {     // S~[p rows] = M1_p Bm~[p rows]
            const int np = p ? na : ns, nzp = p ? nza : nzs;
            g5[p].M = nzp; g5[p].N = (int)RT; g5[p].K = np;
            g5[p].A = M1 + (size_t)(which - 1) * m1_sz + (p ? (size_t)nzs * ns : 0); g5[p].lda = np;
            g5[p].B = Bm + (p ? (size_t)ns * RT : 0); g5[p].ldb = RT;
            g5[p].C = S + (p ? (size_t)nzs * RT : 0); g5[p].ldc = RT;
            g5[p].prof_name = "gemm_pred_cross";
        }
        gemm_pair(c, g5[0], g5[1], s);
        if (which == 1 || !(type & 1)) GP_HIP(hipStreamWaitEvent(s, c->ev_pc, 0));     // Pcat (stream5) before its first use
        for (int p = 0; p < 2; ++p) {     // comp~[(zq, r)][p][cc][b] = sum_i' S~[(zq, r)][p block i'] Pcat_p[i'][cc*npP + b]
            const int np = p ? nta : nts, npP = p ? ntaP : ntsP, c0 = p ? nts : 0;
            // (the padding columns between two components are computed along -- whatever Pcat holds there only reaches comp~'s
            // own padding columns, which nobody reads; the last component's padding is left out)
            g6[p].M = nz * R; g6[p].N = (C - 1) * npP + np; g6[p].K = np;
            g6[p].A = S + c0; g6[p].lda = nt;
            g6[p].B = Pc + (p ? pc_s : 0); g6[p].ldb = (long)C * npP;
            g6[p].C = comp + (p ? (size_t)C * ntsP : 0); g6[p].ldc = ldcomp;
            g6[p].prof_name = "gemm_pred_tstar";
        }
        if (gemm_pred_unfold_supported(C, (long)nz * R, nt)) {
            // ... as ONE launch whose epilogue unfolds in site and time, turns (r, t) into (t, r) and sums the components:
            // comp~ is never written (gemm_f64.hip: gemm_pred_unfold_kernel)
            PredUnfoldDesc pu{};
            pu.S = S; pu.lds = nt;
            pu.Pc[0] = Pc; pu.Pc[1] = Pc + pc_s;
            pu.ldp[0] = (long)C * ntsP; pu.ldp[1] = (long)C * ntaP;
            pu.npP[0] = ntsP; pu.npP[1] = ntaP;
            pu.K[0] = nts; pu.K[1] = nta;
            pu.kcol0[0] = 0; pu.kcol0[1] = nts;
            pu.nb = nts; pu.nba = nta;
            pu.ncolS = (long)nzs * R; pu.ncolA = (long)nza * R; pu.anti_row0 = (long)nzs * R;
            pu.R = R; pu.nt = nt; pu.C = C;
            pu.sz = sz; pu.st = fm.sym_t;
            pu.list = o_list; pu.list_stride = (long)out_elems; pu.sum = o_sum;
            if (!pred_unfold_chunked(c, pu, which - 1, nz, out_elems, want_lists, s)) gemm_pred_unfold(c, pu, s);
        } else {
            gemm_pair(c, g6[0], g6[1], s);
            k_unfold_swap_sum(c, comp, C, o_list, (long)out_elems, o_sum, R, nt, sz, fm.sym_t, s, ntsP, ntaP, ldcomp);
        }
    }
    c->tl("predict end (main)", s);
    if (async && c->prof_mode != 1) {     // results stay on the device: return with the tail still in flight
        c->async_pending = true;
        c->status_zeroed = false;
        return 0;
    }
    return finish_call(c, e, nullptr, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// gpcsd_predict with host outputs (the class API's predict(), gpcsd1d.py:286-293: the arrays ARE the result): the copy of 231 MB
// at cfg3 is 4 ms on the host link against ~1 ms of device work, and it used to start when the last launch had finished.  The
// fused last product writes whole output rows [z][t][r] per site orbit, so it is launched in chunks of column tiles and a chunk's
// finished rows leave through the DMA engine (stream4) while the next chunk computes: the copy starts one chunk (~40 us) after
// the product instead of ~220 us.  Rows: orbit a writes z = rep_i[a] and rep_j[a] -- runs of consecutive rows are one copy each
// (a mirror-symmetric probe gives two runs per chunk).  Returns false when it does not apply (the caller launches the product whole).
// gpcsd_predict_chunked_copy(ctx, 0) / GPCSD_PRED_CHUNKED=0: A/B.
static bool pred_unfold_chunked(gpcsd_ctx *c, PredUnfoldDesc pu, int which0, int nz, size_t out_elems, bool want_lists, hipStream_t s) {
    gpcsd_ctx::PredSink &sk = c->pred_sink;
    if (!sk.active || !c->pred_chunked || !sk.sum[which0] || (want_lists && !sk.list[which0]) || c->prof_mode == 1) return false;
    const auto it = c->sym_host.find(pu.sz.rep_i);
    if (it == c->sym_host.end()) return false;
    const int nso = pu.sz.ns;                                         // site orbits
    if ((int)it->second.size() < 2 * nso || pu.ncolS != (long)nso * pu.R) return false;
    const int *rep_i = it->second.data(), *rep_j = rep_i + nso;
    const long tiles_all = (pu.ncolS + PRED_UNFOLD_BN - 1) / PRED_UNFOLD_BN;
    const size_t row_bytes = (size_t)pu.nt * pu.R * sizeof(double);
    const size_t total = (size_t)nz * row_bytes * (1 + (want_lists ? pu.C : 0));
    if (total < ((size_t)32 << 20) || tiles_all < 16) return false;   // small outputs: one launch, one copy
    // chunk boundaries in eighths of the tiles (GPCSD_PRED_CHUNKS="1,8": a first eighth, then the rest).  Every copy costs the DMA
    // engine ~10 us of set-up and a chunk is two runs of rows x (1 + C) arrays: six equal chunks (36 copies) were SLOWER than one
    // copy behind the whole product (5.60 against 5.41 ms per call at cfg3) -- what pays is starting the first copy early.
    static const std::vector<int> cuts = [] {
        std::vector<int> v;
        const char *e = getenv("GPCSD_PRED_CHUNKS");
        for (const char *p = e ? e : "1,8"; *p;) {
            v.push_back(atoi(p));
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        if (v.empty() || v.back() != 8) v.push_back(8);
        return v;
    }();
    const int nchunk = (int)cuts.size();
    hipStream_t sc = c->stream4;
    std::vector<char> have((size_t)nz, 0);
    int done_orbits = 0;
    for (int k = 0; k < nchunk; ++k) {
        pu.tile_c0 = k == 0 ? 0 : tiles_all * cuts[k - 1] / 8;
        pu.tile_c1 = tiles_all * cuts[k] / 8;
        gemm_pred_unfold(c, pu, s);
        // orbits all of whose R columns lie in front of the end of this chunk are complete
        const int upto = (k + 1 == nchunk) ? nso : (int)std::min<long>(nso, pu.tile_c1 * PRED_UNFOLD_BN / pu.R);
        if (upto <= done_orbits) continue;
        std::vector<int> rows;
        for (int a = done_orbits; a < upto; ++a) {
            for (int z : {rep_i[a], rep_j[a]})
                if (z >= 0 && z < nz && !have[z]) { have[z] = 1; rows.push_back(z); }
        }
        done_orbits = upto;
        std::sort(rows.begin(), rows.end());
        hipEvent_t ev = c->get_event();
        GP_HIP(hipEventRecord(ev, s));
        GP_HIP(hipStreamWaitEvent(sc, ev, 0));
        c->pred_sink_events.push_back(ev);                        // (back to the pool once gpcsd_predict has synchronised)
        for (size_t i = 0; i < rows.size();) {
            size_t j = i + 1;
            while (j < rows.size() && rows[j] == rows[j - 1] + 1) ++j;
            const size_t off = (size_t)rows[i] * (size_t)pu.nt * pu.R, cnt = (j - i) * row_bytes;
            GP_HIP(hipMemcpyAsync(sk.sum[which0] + off, pu.sum + off, cnt, hipMemcpyDeviceToHost, sc));
            if (want_lists)
                for (int cc = 0; cc < pu.C; ++cc)
                    GP_HIP(hipMemcpyAsync(sk.list[which0] + (size_t)cc * out_elems + off, pu.list + (size_t)cc * pu.list_stride + off, cnt,
                                          hipMemcpyDeviceToHost, sc));
            i = j;
        }
    }
    for (int z = 0; z < nz; ++z)
        if (!have[z]) return true;                                    // (cannot happen: every site belongs to an orbit) -- not marked done
    sk.done[which0] = true;
    ++c->pred_chunked_calls;
    return true;
}

// Posterior mean into ctx-owned device buffers, already in the reference's output layout (z, t, trial):
//   pred_out_csd / pred_out_lfp            (nz, ntstar, R)
//   pred_out_csd_list / pred_out_lfp_list  (C, nz, ntstar, R)     when want_lists
static int predict_impl(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                        int type, bool want_lists, bool async) {
    GP_REQUIRE(z && tstar && nz > 0 && ntstar > 0, -3, "predict: bad arguments");
    GP_REQUIRE(type >= 1 && type <= 3, -3, "predict: type must be CSD(1), LFP(2) or BOTH(3)");
    GP_REQUIRE(c->nt > 0 && ntstar == c->nt, -22,
               "predict: len(t)=%d must equal the training nt=%d (the reference's reshape raises ValueError, gpcsd1d.py:279)",
               ntstar, c->nt);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    // folded basis when the grids, the prediction sites and the prediction times all share the reflection symmetries
    const FoldMode fm0 = fold_mode(c, hp);         // the decision only: views are taken after the front half
    // a folded side needs its outputs on a grid with the same symmetry (t* = t; mirror-symmetric sites); an unfolded side
    // takes any sites / times
    const bool t_ok = fm0.on && ntstar == c->nt &&
                      (!fm0.ft.on || ((int)c->time_host.size() == c->nt &&
                                      memcmp(c->time_host.data(), tstar, (size_t)ntstar * sizeof(double)) == 0));
    if (t_ok) {
        SymDev sz = fm0.fs.on ? site_symmetry(c, z, nz, c->dim) : identity_sym(c, nz);
        // sites without the electrodes' symmetry (the reference's own use: four off-grid depths): the spatial side unfolded, the
        // temporal side folded as ever -- not the full-size path
        bool fold_s = true;
        if (!(sz.ns > 0 && sz.ns + sz.na == nz) && fm0.fs.on && fm0.ft.on) {
            fold_s = false;
            sz = identity_sym(c, nz);
        }
        if (sz.ns > 0 && sz.ns + sz.na == nz) {
            // the chains go first (they need no upload of this call), then the host-side uploads
            const bool ptri = fm0.ft.on && predict_tridiag_applies(fm0.ft.ns, fm0.ft.na, c->ntrials);
            c->q_pipe_want = ptri;                                          // (X = Y~ Q goes through loglik_tri_pre: stage 5 may apply)
            EigState ef;
            try {
                ef = front_half(c, hp, 0.0, false, /*join_s=*/false, /*want_tri=*/ptri, fold_s ? -1 : 1);  // no jitter in predict (gpcsd1d.py:258)
            } catch (...) {
                c->q_pipe_want = false;
                throw;
            }
            c->q_pipe_want = false;
            const FoldMode fm = fold_mode(c, hp, fold_s);
            const double *Yf = folded_lfp(c, fm);
            double *dzf = c->upload_cached<double>("pred_z", z, (size_t)nz * c->dim);
            double *dtf = c->upload_cached<double>("pred_tstar", tstar, ntstar);
            return predict_fold(c, hp, ef, fm, Yf, sz, dzf, nz, dtf, type, want_lists, async);
        }
    }
    EigState e = front_half(c, hp, 0.0);           // no jitter in predict (gpcsd1d.py:258)
    const Geo g = resident_geo(c);
    const int nx = c->nx, nt = c->nt, R = c->ntrials, C = hp->n_temporal;
    const long RT = (long)R * nt;
    hipStream_t s = c->stream;
    double *W = c->buf<double>("proj_W", (size_t)nx * RT);
    double *Bm = c->buf<double>("pred_B", (size_t)nx * RT);
    GemmDesc g1;                          // W = Qs^T Y
    g1.M = nx; g1.N = (int)RT; g1.K = nx;
    g1.A = e.Qs; g1.lda = nx; g1.transA = true;
    g1.B = c->d_lfp; g1.ldb = RT; g1.C = W; g1.ldc = RT;
    g1.prof_name = "gemm_proj_spatial";
    gemm_f64(c, g1, s);
    // invy = (Qs (x) Qt) vec(Bm) (gpcsd1d.py:262-265) is never formed: the cross-covariance contraction
    //   out_c = Kc^T Qs Bm Qt^T Kt*_c  is re-associated as  (Kc^T Qs) Bm (Qt^T Kt*_c),
    // i.e. two small (n^3) products M1, Pc and two flat GEMMs, instead of back-projecting to the original bases first
    // (saves 2 nx^2 nt + 2 nx nt^2 flops per trial; identical up to rounding).
    double *dz = c->upload_cached<double>("pred_z", z, (size_t)nz * g.dim);
    double *dts = c->upload_cached<double>("pred_tstar", tstar, ntstar);
    const double *t = (const double *)c->bufs["time_t"].p;
    double *Kc = c->buf<double>("pred_Kcross", (size_t)nx * nz);
    double *S = c->buf<double>("pred_S", (size_t)nz * RT);
    double *comp = c->buf<double>("pred_comp", (size_t)C * nz * RT);
    double *Kts = c->buf<double>("pred_Ktstar", (size_t)C * ntstar * nt);
    double *M1 = c->buf<double>("pred_M1", (size_t)2 * nz * nx);
    double *Pc = c->buf<double>("pred_Pc", (size_t)C * nt * nt);
    const size_t out_elems = (size_t)nz * RT;
    // Everything that needs only Qs is queued before the join, i.e. it runs beside the temporal eigensolver:
    // cross-covariances Kc, M1 = Kc^T Qs for the requested outputs, and the prediction-time temporal Grams.
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        if (which == 1) build_kphig(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, Kc, s);        // gpcsd1d.py:273
        else build_kphi(c, g, hp->R, hp->eps, hp->ell_s, dz, nz, 0.0, Kc, s);               // gpcsd1d.py:275
        GemmDesc gm;                      // M1[z][x'] = sum_x Kc[x][z] Qs[x][x']
        gm.M = nz; gm.N = nx; gm.K = nx;
        gm.A = Kc; gm.lda = nz; gm.transA = true; gm.B = e.Qs; gm.ldb = nx; gm.C = M1 + (size_t)(which - 1) * nz * nx; gm.ldc = nx;
        gm.prof_name = "gemm_pred_M1";
        gemm_f64(c, gm, s);
    }
    GP_HIP(hipEventRecord(c->ev_m1, s));          // (a later paired call rewrites the cross-covariances on stream5)
    c->m1_read_queued = true;
    for (int cc = 0; cc < C; ++cc) {
        // Ktstar_c = cov_c.compute_Kt(tstar): (ntstar, nt); its FIRST axis is contracted with the training
        // time index (reference quirk when tstar != t)      gpcsd1d.py:277-279
        temporal_cross_gram(c, hp, cc, dts, ntstar, t, nt, Kts + (size_t)cc * ntstar * nt, s);
    }
    join_temporal(c, e, nullptr, false);      // predict never reads sum(log D)
    GemmDesc g2;                          // Bm = (W Qt) / D
    g2.M = nx * R; g2.N = nt; g2.K = nt;
    g2.A = W; g2.lda = nt; g2.B = e.Qt; g2.ldb = nt; g2.C = Bm; g2.ldc = nt;
    g2.epi = EPI_DIV_D; g2.D = e.Dinv; g2.rdiv = R; g2.ldd = nt;
    g2.prof_name = "gemm_pred_temporal_div";
    gemm_f64(c, g2, s);
    for (int which = 1; which <= 2; ++which) {
        if (!(type & which)) continue;
        double *o_sum = c->buf<double>(which == 1 ? "pred_out_csd" : "pred_out_lfp", out_elems);
        double *o_list = want_lists ? c->buf<double>(which == 1 ? "pred_out_csd_list" : "pred_out_lfp_list", out_elems * C)
                                    : nullptr;
        GemmDesc g5;                      // S[z][(r,i')] = sum_x' M1[z][x'] Bm[x'][(r,i')]
        g5.M = nz; g5.N = (int)RT; g5.K = nx;
        g5.A = M1 + (size_t)(which - 1) * nz * nx; g5.lda = nx; g5.B = Bm; g5.ldb = RT; g5.C = S; g5.ldc = RT;
        g5.prof_name = "gemm_pred_cross";
        gemm_f64(c, g5, s);
        for (int cc = 0; cc < C; ++cc) {
            GemmDesc gp;                  // Pcat[i'][cc*nt + t'] = sum_j Qt[j][i'] Ktstar_cc[j][t']
            gp.M = nt; gp.N = nt; gp.K = ntstar;
            gp.A = e.Qt; gp.lda = nt; gp.transA = true; gp.B = Kts + (size_t)cc * ntstar * nt; gp.ldb = nt;
            gp.C = Pc + (size_t)cc * nt; gp.ldc = (long)C * nt;
            gp.prof_name = "gemm_pred_Pc";
            gemm_f64(c, gp, s);
        }
        // All temporal components in ONE flat GEMM: out[(z,r)][cc*nt + t'] = sum_i' S[(z,r)][i'] Pcat[i'][cc*nt + t'],
        // then one pass writes every component in the reference's (z, t, r) layout plus their sum (no read-modify-write
        // epilogue, one launch instead of C, a single relayout pass instead of C + 1).
        GemmDesc g6;
        g6.M = nz * R; g6.N = C * nt; g6.K = nt;
        g6.A = S; g6.lda = nt; g6.B = Pc; g6.ldb = (long)C * nt; g6.C = comp; g6.ldc = (long)C * nt;
        g6.prof_name = "gemm_pred_tstar";
        gemm_f64(c, g6, s);
        k_swap_last2_sum(c, comp, C, o_list, (long)out_elems, o_sum, nz, R, nt, s);     // (z,r,c,t) -> (c,z,t,r), sum over c
    }
    return finish_call(c, e, nullptr, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// loglik + predict as ONE queued call with the four decompositions batched two by two.
//
// Chains of small dependent launches do not overlap on this part (1.4x at best, however many queues), but
// replicas inside one chain are nearly free (gpcsd_eigh_batch: 8 problems in 1.09 ms against 0.93 ms for one).  So when a
// caller wants the log-likelihood at one hyper-parameter set and the prediction at another (the same set without jitter, in
// practice), the two temporal problems go through ONE chain as two replicas and the two spatial problems through another:
// two chains per pair of calls instead of four.  Every problem is still solved (nothing is reused between the two unless the
// decomposition cache is on and the temporal hyper-parameters coincide: then that side is solved once, as the cache would).
// Results: the bits of the two calls made separately.
struct PairFront {
    EigState e[2];
    FoldMode fm[2];
    bool share_s = false;       // one spatial decomposition served both sets (gpcsd_ctx::pair_share_s)
};

static bool same_temporal(const gpcsd_hparams *a, const gpcsd_hparams *b) {
    if (a->n_temporal != b->n_temporal) return false;
    for (int i = 0; i < a->n_temporal; ++i)
        if (a->kind[i] != b->kind[i] || a->ell_t[i] != b->ell_t[i] || a->sigma2_t[i] != b->sigma2_t[i]) return false;
    return true;
}

// Both sets decomposed, set b's results at replica b of the generation just started (folded-basis callers only).
// pred_tri: the prediction (set 1) takes the tridiagonal form too, so that nobody reads the temporal spectrum or eigenvectors: the
// staged temporal chain then stops behind stages 1 and 3.
// fold_s = false: the tails take the spatial side unfolded (merged eigenvectors; the eigensolver still folds it internally).
static void front_half_pair(gpcsd_ctx *c, const gpcsd_hparams *const hp[2], const double jitter[2], PairFront &out, bool pred_tri,
                            bool fold_s = true) {
    const Geo g = resident_geo(c);
    const int nx = c->nx, nt = c->nt;
    const long nxx = (long)nx * nx, ntt = (long)nt * nt;
    hipStream_t s = c->stream, s2 = c->stream2, s3 = c->stream3;
    const SymDev *sym_s = c->sym_s.ns > 0 ? &c->sym_s : nullptr, *sym_t = c->sym_t.ns > 0 ? &c->sym_t : nullptr;
    const double *t = (const double *)c->bufs["time_t"].p;
    // replicas of the temporal problem: ONE when the two sets have the same temporal hyper-parameters (every loglik -> predict
    // pair: the jitter is spatial) and either the decomposition cache or the pair's sharing of X says that equal sides are formed
    // once -- the second replica would be the same launches on the same matrix, the same bits (GPCSD_PAIR_ONE_KT=0: A/B)
    static const bool one_kt_off = getenv("GPCSD_PAIR_ONE_KT") && getenv("GPCSD_PAIR_ONE_KT")[0] == '0';
    const int nT = (same_temporal(hp[0], hp[1]) && (c->decomp_cache_on || (c->pair_share_x && !one_kt_off))) ? 1 : 2;
    double *scal = c->buf<double>("scal_status", gpcsd_ctx::RESULT_DOUBLES);
    int *status = reinterpret_cast<int *>(scal + gpcsd_ctx::SCAL_N);
    int *late = status + gpcsd_ctx::STATUS_LATE;           // stages 2 and 4 of a staged temporal chain report here
    const bool clear_now = !c->status_zeroed && !c->async_pending;
    if (clear_now) GP_HIP(hipMemsetAsync(status, 0, gpcsd_ctx::STATUS_N * sizeof(int), s));
    c->status_zeroed = false;
    begin_generation(c, 1, s2, clear_now);
    begin_generation(c, 0, s3, clear_now);
    // inputs and outputs (of the generations just started), two replicas each.  The inputs have names of their own: the
    // one-chain form below reads them on stream2, the separate calls' spatial chain writes "Ks" on stream3.
    double *Ks = c->buf<double>("Ks_pair", (size_t)nxx * 2), *Kt = c->buf<double>("Kt_pair", (size_t)ntt * 2);
    double *Qs = c->buf<double>(gen_name(c, 0, "Qs"), (size_t)nxx * 2), *es = c->buf<double>(gen_name(c, 0, "es"), (size_t)nx * 2);
    double *Qt = c->buf<double>(gen_name(c, 1, "Qt"), (size_t)ntt * 2), *et = c->buf<double>(gen_name(c, 1, "et"), (size_t)nt * 2);
    const FoldView vs = sym_s ? eigh_fold_view(c, 0, sym_s, nx, 2) : FoldView(), vt = sym_t ? eigh_fold_view(c, 1, sym_t, nt, 2) : FoldView();
    c->tl("call start (main)", s);
    // Gram matrices.  Temporal (stream2): replica b = Kt(hp[b]).  Spatial (stream3): replica b = Ks(hp[b]) + jitter[b] I; with
    // equal spatial hyper-parameters -- the usual pair -- the two differ by the diagonal shift only, so the matrix is
    // assembled once and copied (the same GEMM output plus the same diagonal add: the same bits).
    const bool tfill = temporal_fill_applies(c, sym_t, nt, false);       // (the paired call is refused for host temporal Grams)
    const bool staged = tfill && ll_tridiag_enabled(c) && eigh_stageable(sym_t, nt);
    bool st5 = false, pipe = false;      // stage 5's kernels instead of stage 3's, and pipelined: decided in part 1
    // part 1: the inputs and (staged) stage 1, or the whole chain; part 2 (staged only): stage 2, and stage 3 beside it
    auto run_T = [&](int part) {
        if (part == 1) {
            c->tl("T chain start (s2)", s2);
            c->tgen ^= 1;                    // the other generation of the temporal class arenas (gpcsd_ctx::tgen)
            staged_chain_guard(c, s2);
            clear_late_status(c, status, s2, staged && !pred_tri);
            if (tfill) temporal_fill(c, hp, nT, t, nt, *sym_t, status + 1, 2, s2);
            else for (int b = 0; b < nT; ++b) build_kt(c, hp[b], t, nt, t, nt, Kt + b * ntt, s2);
        }
        // the temporal chain is the critical path of the call: it is queued before the host spends its time on the launches of the
        // spatial Gram assembly (status words [1], [3]; one replica when the problem is shared -- decomposition cache on, equal
        // temporal hyper-parameters).  (All four problems in ONE chain was measured slower, 1.38 against 1.18 ms per cfg3 step: with
        // two chains the log-likelihood's spatial projection runs under the end of the temporal one.)
        {
            ProfScope ps(c, part == 2 ? "eigh_temporal_stage2" : "eigh_temporal", part == 2 ? 0.0 : 9.0 * (double)nt * nt * nt * nT, s2);
            if (staged && part == 1) {   // the log-likelihood's tail starts behind stages 1 + 3 (see front_half, EigState::tri)
                // (only when the prediction takes the tridiagonal form too: a stage 4 behind stage 5 would read T factors summed in
                // another order than stage 3's, and the pair would differ from its fenced calls in the last bits)
                st5 = pred_tri && q_stage5_applies(c, sym_t);
                pipe = st5 && q_pipe_applies(c, sym_t);
                if (pipe) GP_HIP(hipEventRecord(c->ev_t0, s2));
                c->pipe_req = st5 ? 1 : 0;
                eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, status + 1, s2, false, nT, 2, -1, 2, 1);
                c->pipe_req = 0;
                GP_HIP(hipEventRecord(c->ev_t1, s2));
                c->tl("T stage 1 end (s2)", s2);
                return;
            }
            if (staged) {
                // stage 2 (divide & conquer) on the chain's stream; beside it, on stream4, stage 3 (T factors, Q); stage 4
                // (back-transformation) behind both.  (Stage 3 on the main stream, in front of X: 1.14 against 1.10 ms -- the
                // main stream is rarely idle when stage 1 ends.)  With the prediction in the tridiagonal form as well: stage 3 alone.
                if (!pred_tri)
                    eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, late + 1, s2, false, nT, 2, -1, 2, 2);
                hipStream_t sq = c->stream4;
                if (pipe) {              // (stage 5 is queued by the log-likelihood's loglik_tri_pre: EigState::pipe_pending)
                    c->q_queued[c->tgen] = false;
                } else {
                    if (st5) {
                        queue_stage5_plain(c, Kt, nt, et, Qt, sym_t, status + 1, false, nT, 2);
                    } else {
                        GP_HIP(hipStreamWaitEvent(sq, c->ev_t1, 0));
                        GP_HIP(hipStreamWaitEvent(sq, c->ev_pc, 0));      // (stream5's readers of the Q about to be rewritten)
                        eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, status + 1, sq, false, nT, 2, -1, 2, 3);
                    }
                    GP_HIP(hipEventRecord(c->ev_q[c->tgen], sq));
                    c->tl("Q end", sq);
                    c->q_queued[c->tgen] = true;
                }
                c->q_gen = -1;           // (replicas: not what a separate call's cache looks for)
                if (!pred_tri) {
                    GP_HIP(hipStreamWaitEvent(s2, c->ev_q[c->tgen], 0));
                    eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, late + 1, s2, false, nT, 2, -1, 2, 4);
                }
            } else {
                eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, status + 1, s2, false, nT, 2, -1,
                                 tfill ? 2 : 0);
            }
        }
        GP_HIP(hipEventRecord(c->ev_join, s2));
        c->tl("T chain end (s2)", s2);
    };
    const bool same_ks = hp[0]->R == hp[1]->R && hp[0]->ell_s[0] == hp[1]->ell_s[0] &&
                         (g.dim == 1 || (hp[0]->eps == hp[1]->eps && hp[0]->ell_s[1] == hp[1]->ell_s[1]));
    // Equal spatial hyper-parameters: the two matrices are Ks + jitter[b] I -- the same eigenvectors, spectra that differ by the
    // shift.  ONE replica is decomposed (set 0's, with its jitter); set 1 reads its eigenvectors and a shifted copy of its spectrum.
    const bool share_s = c->pair_share_s && same_ks;
    const int nS = share_s ? 1 : 2;
    out.share_s = share_s;
    if (share_s) ++c->pair_shared_s_calls;
    auto run_S = [&]() {
        c->tl("S chain start (s3)", s3);
        const bool sfill = spatial_fill_applies(c, sym_s, nx);
        if (sfill) {
            // the fill folds Ks and adds each replica's jitter to the folded diagonals: one assembly, no copy, no diagonal pass
            if (same_ks) build_kphi(c, g, hp[0]->R, hp[0]->eps, hp[0]->ell_s, nullptr, 0, 0.0, Ks, s3, "ks_");
            else for (int b = 0; b < 2; ++b) build_kphi(c, g, hp[b]->R, hp[b]->eps, hp[b]->ell_s, nullptr, 0, 0.0, Ks + b * nxx, s3, "ks_");
            spatial_fill(c, Ks, nx, same_ks ? 0 : nxx, nS, jitter, *sym_s, status, 2, s3);
        } else if (share_s) {
            build_kphi(c, g, hp[0]->R, hp[0]->eps, hp[0]->ell_s, nullptr, 0, jitter[0], Ks, s3, "ks_");
        } else if (same_ks) {
            const int lo = jitter[0] == 0.0 ? 0 : 1, hi = 1 - lo;           // assemble the one without a shift (if any) first
            build_kphi(c, g, hp[lo]->R, hp[lo]->eps, hp[lo]->ell_s, nullptr, 0, 0.0, Ks + lo * nxx, s3, "ks_");
            GP_HIP(hipMemcpyAsync(Ks + hi * nxx, Ks + lo * nxx, (size_t)nxx * sizeof(double), hipMemcpyDeviceToDevice, s3));
            if (jitter[lo] != 0.0) k_add_diag(c, Ks + lo * nxx, nx, jitter[lo], s3);
            if (jitter[hi] != 0.0) k_add_diag(c, Ks + hi * nxx, nx, jitter[hi], s3);
        } else {
            for (int b = 0; b < 2; ++b) build_kphi(c, g, hp[b]->R, hp[b]->eps, hp[b]->ell_s, nullptr, 0, jitter[b], Ks + b * nxx, s3, "ks_");
        }
        // two replicas of the spatial problem on stream3 (status words [0], [2]) -- or the one both sets share
        {
            ProfScope ps(c, "eigh_spatial", 9.0 * (double)nx * nx * nx * nS, s3);
            eigh_pair_device(c, Ks, nx, es, Qs, sym_s, nullptr, 0, nullptr, nullptr, nullptr, status, s3, /*need_merged=*/!fold_s, nS, 2, -1,
                             sfill ? 1 : 0);
        }
        if (share_s) {                   // set 1's spectrum: set 0's shifted by the difference of the jitters, in replica 1's slots
            const double dj = jitter[1] - jitter[0];
            if (vs.on) k_shift_copy(c, vs.w, vs.w + vs.sw, nx, dj, s3);
            if (!vs.on || !fold_s) k_shift_copy(c, es, es + nx, nx, dj, s3);
        }
        GP_HIP(hipEventRecord(c->ev_sjoin, s3));
        c->tl("S chain end (s3)", s3);
    };
    // The temporal chain's long first kernels go first; staged, the host queues the spatial chain (~0.1 ms of launches) before
    // it comes back for the temporal chain's second stage -- stage 1 runs half a millisecond, the queue is never empty
    static const bool s_first = getenv("GPCSD_S_FIRST") && getenv("GPCSD_S_FIRST")[0] == '1';     // (A/B: the spatial chain queued first)
    if (s_first) {
        run_S();
        run_T(1);
    } else {
        run_T(1);
        run_S();
    }
    if (staged) run_T(2);
    c->decomp_gen[0] = c->decomp_gen[1] = -1;          // replicas are not what the separate calls' cache looks for
    const double *d_sig[2] = {c->upload_cached<double>("sig2n", hp[0]->sig2n, 1), c->upload_cached<double>("sig2n_pair", hp[1]->sig2n, 1)};
    for (int b = 0; b < 2; ++b) {
        const int bt = nT == 2 ? b : 0;
        EigState &e = out.e[b];
        const int bs = share_s ? 0 : b;      // the replica whose EIGENVECTORS set b reads (the spectra: always slot b)
        e.Qs = Qs + bs * nxx; e.es = es + (long)b * nx; e.Qt = Qt + bt * ntt; e.et = et + (long)bt * nt;
        e.D = c->buf<double>("D", (size_t)nx * nt);
        e.Dinv = c->buf<double>("Dinv", (size_t)nx * nt);
        e.scal = scal;
        e.status = status;
        e.pending = true;
        e.wait_temporal = e.wait_spatial = true;
        e.d_sig = d_sig[b];
        e.nsig = 1;
        if (staged && (b == 0 || pred_tri)) {      // set 0 is the log-likelihood's: replica 0 of the temporal classes
            e.tri = e.wait_q = true;
            e.tri_rep = bt;
            e.tri_count = nT;
            if (pipe && b == 0) {        // X with replica 0's Q: the log-likelihood's set
                e.pipe_pending = true;
                e.pa.Kt = Kt; e.pa.nt = nt; e.pa.et = et; e.pa.Qt = Qt; e.pa.sym_t = sym_t; e.pa.status = status + 1;
                e.pa.need_merged = false; e.pa.nT = nT; e.pa.stride = 2; e.pa.rep = 0; e.pa.q_gen = -1;
            }
        }
        FoldMode &fm = out.fm[b];
        fm = fold_mode(c, hp[b], fold_s);               // replica 0 of the generations just started ...
        if (fm.fs.on) { fm.fs.w += (long)b * vs.sw; fm.fs.U += (long)bs * vs.sU; }      // ... moved to replica b
        else { fm.fs.w += (long)b * nx; fm.fs.U += bs * nxx; }
        if (fm.ft.on) { fm.ft.w += (long)bt * vt.sw; fm.ft.U += (long)bt * vt.sU; }
        else { fm.ft.w += (long)bt * nt; fm.ft.U += bt * ntt; }
    }
}

// Collect the status words of an asynchronous predict (see gpcsd_ctx::async_pending): drains the streams.
static int drain_async(gpcsd_ctx *c) {
    if (!c->async_pending) return 0;
    c->async_pending = false;
    int *st = reinterpret_cast<int *>(c->buf<double>("scal_status", gpcsd_ctx::RESULT_DOUBLES) + gpcsd_ctx::SCAL_N);
    GP_HIP(hipStreamSynchronize(c->stream2));
    GP_HIP(hipStreamSynchronize(c->stream3));
    int rc = finish_status(c, st, gpcsd_ctx::STATUS_N);        // downloads + synchronises; the words are cleared by the next call's front half
    if (c->last_pred.have && q_pipe_missed(c, rc, c->last_pred.piped)) {      // (see q_pipe_missed: the queued prediction again)
        const gpcsd_ctx::PredKeep &k = c->last_pred;
        rc = predict_impl(c, &k.hp.hp, k.z.data(), k.nz, k.ts.data(), k.nts, k.type, k.lists, false);
    }
    return rc;
}

extern "C" int gpcsd_predict_resident(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar,
                                      int ntstar, int type, int want_lists) {
    GP_API_BEGIN(c)
    const bool piped = c->q_pipe;
    const int rc = predict_impl(c, hp, z, nz, tstar, ntstar, type, want_lists != 0, /*async=*/true);
    if (rc == 0) keep_prediction(c, piped, hp, z, nz, tstar, ntstar, type, want_lists != 0);
    return rc;
    GP_API_END(c)
}

// What a paired call decides before it queues anything: does the paired front half apply, and in which form.
struct PairPlan {
    bool pair = false, fold_s = true, pred_tri = false;
    FoldMode fm0;
    SymDev sz;
};
static PairPlan pair_plan(gpcsd_ctx *c, const gpcsd_hparams *hp_ll, const gpcsd_hparams *hp_pr, const double *z, int nz, const double *tstar,
                          int ntstar) {
    PairPlan P;
    // the paired front half serves the folded-basis tails only; anything else is the two calls one after the other
    bool pair = two_stream_front() && c->prof_mode != 1 && !uses_host_kt(hp_ll) && !uses_host_kt(hp_pr) &&
                hp_ll->n_sig2n == 1 && hp_pr->n_sig2n == 1;
    if (pair) {
        P.fm0 = fold_mode(c, hp_ll);
        pair = P.fm0.on && fold_mode(c, hp_pr).on &&
               (!P.fm0.ft.on || ((int)c->time_host.size() == c->nt &&
                                 memcmp(c->time_host.data(), tstar, (size_t)ntstar * sizeof(double)) == 0));
    }
    if (pair) {
        P.sz = P.fm0.fs.on ? site_symmetry(c, z, nz, c->dim) : identity_sym(c, nz);
        if (!(P.sz.ns > 0 && P.sz.ns + P.sz.na == nz) && P.fm0.fs.on && P.fm0.ft.on) {    // sites without the electrodes' symmetry: spatial
            P.fold_s = false;                                                                // side unfolded for BOTH sets (see predict_impl)
            P.sz = identity_sym(c, nz);
            P.fm0 = fold_mode(c, hp_ll, false);
        }
        pair = P.sz.ns > 0 && P.sz.ns + P.sz.na == nz;
    }
    P.pair = pair;
    // (decided here, where the fold sizes are known: does the prediction take the tridiagonal form too?)
    if (pair) P.pred_tri = P.fm0.ft.on && predict_tridiag_applies(P.fm0.ft.ns, P.fm0.ft.na, c->ntrials);
    return P;
}

// Everything the paired front half's launches depend on (a prefetched front half is only taken by a call with the same key)
static std::vector<unsigned char> pair_key(const gpcsd_ctx *c, const gpcsd_hparams *const hp[2], const double jit[2], const PairPlan &P) {
    std::vector<double> k;
    for (int b = 0; b < 2; ++b) {
        const gpcsd_hparams *h = hp[b];
        k.push_back(h->R); k.push_back(h->eps); k.push_back(h->ell_s[0]); k.push_back(h->ell_s[1]); k.push_back(jit[b]);
        k.push_back((double)h->n_temporal); k.push_back((double)h->n_sig2n); k.push_back(h->sig2n[0]);
        for (int i = 0; i < h->n_temporal; ++i) { k.push_back((double)h->kind[i]); k.push_back(h->ell_t[i]); k.push_back(h->sigma2_t[i]); }
    }
    const double cfg[] = {(double)P.pred_tri, (double)P.fold_s, (double)c->grid_epoch, (double)c->alloc_epoch, (double)c->ntrials, (double)c->nx,
                          (double)c->nt, (double)c->q_pipe, (double)c->pair_share_s, (double)c->tail_early_exit,
                          (double)c->ll_tridiag_mode, (double)c->gram_fp32, (double)c->decomp_cache_on, (double)c->fold_gemm_on};
    k.insert(k.end(), cfg, cfg + sizeof(cfg) / sizeof(cfg[0]));
    const unsigned char *p = reinterpret_cast<const unsigned char *>(k.data());
    return std::vector<unsigned char>(p, p + k.size() * sizeof(double));
}

struct PairPrefetch {
    std::vector<unsigned char> key;
    PairFront pf;
};
static void pair_prefetch_drop(gpcsd_ctx *c) {
    delete static_cast<PairPrefetch *>(c->pair_prefetch);
    c->pair_prefetch = nullptr;
}

// The paired front half with the promise that X = Y~ Q goes through loglik_tri_pre (stage 5 may apply)
static void front_half_pair_q(gpcsd_ctx *c, const gpcsd_hparams *const hps[2], const double jit[2], PairFront &pf, const PairPlan &P,
                              bool pipelined = true) {
    c->q_pipe_want = P.fm0.ft.on && pipelined;
    try {
        front_half_pair(c, hps, jit, pf, P.pred_tri, P.fold_s);
    } catch (...) {
        c->q_pipe_want = false;
        throw;
    }
    c->q_pipe_want = false;
}

// gpcsd_prefetch_pair: the caller knows the hyper-parameters of its NEXT paired call (a grid or a chain of proposals fixed in
// advance, the replicas of a lock-step batch, a benchmark loop): the two decomposition chains of that call are queued NOW, on their own
// streams, and start as soon as those streams are free -- typically while the current call's log-likelihood is still being
// formed, instead of after the host has collected it.  The next gpcsd_loglik_predict_async with the same arguments takes them
// over; any other call on the context drops them (they have then run for nothing).  Returns 1 when queued, 0 when the paired
// form does not apply to these arguments.
extern "C" int gpcsd_prefetch_pair(gpcsd_ctx *c, const gpcsd_hparams *hp_ll, const gpcsd_hparams *hp_pr, const double *z, int nz,
                                   const double *tstar, int ntstar) {
    GP_API_BEGIN(c)
    GP_REQUIRE(hp_ll && hp_pr && z && tstar && nz > 0 && ntstar > 0, -3, "prefetch_pair: bad arguments");
    GP_REQUIRE(c->d_lfp != nullptr && c->nt > 0 && ntstar == c->nt, -4, "prefetch_pair: resident data / time grid do not match");
    pair_prefetch_drop(c);
    const PairPlan P = pair_plan(c, hp_ll, hp_pr, z, nz, tstar, ntstar);
    if (!P.pair || c->time_nt != c->nt || resident_geo(c).nx != c->nx) return 0;
    check_hp(c, hp_ll, c->nx);
    check_hp(c, hp_pr, c->nx);
    const gpcsd_hparams *hps[2] = {hp_ll, hp_pr};
    const double jit[2] = {hp_ll->jitter, 0.0};
    PairPrefetch *pp = new PairPrefetch();
    try {
        // (not pipelined: the chain has a whole step's head start, T and Q follow it at once -- the same kernels, the same bits)
        front_half_pair_q(c, hps, jit, pp->pf, P, /*pipelined=*/false);
    } catch (...) {
        delete pp;
        throw;
    }
    pp->key = pair_key(c, hps, jit, P);
    c->pair_prefetch = pp;                 // (begin_generation dropped whatever was there; set after the front half for the same reason)
    ++c->pair_prefetch_queued;
    return 1;
    GP_API_END(c)
}

extern "C" int gpcsd_prefetch_stats(gpcsd_ctx *c, long *queued, long *taken) {
    GP_API_BEGIN(c)
    if (queued) *queued = c->pair_prefetch_queued;
    if (taken) *taken = c->pair_prefetch_taken;
    return 0;
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_predict_async(gpcsd_ctx *c, const gpcsd_hparams *hp_ll, const gpcsd_hparams *hp_pr, const double *z, int nz,
                                          const double *tstar, int ntstar, int type, int want_lists) {
    if (c && c->ll_count >= gpcsd_ctx::LL_SLOTS)
        return fail(c, HipError{-3, "loglik_predict_async: too many asynchronous evaluations outstanding (collect with "
                                    "gpcsd_loglik_parts_wait)"});
    GP_API_BEGIN(c)
    GP_REQUIRE(hp_ll && hp_pr, -3, "loglik_predict_async: null hparams");
    GP_REQUIRE(z && tstar && nz > 0 && ntstar > 0, -3, "predict: bad arguments");
    GP_REQUIRE(type >= 1 && type <= 3, -3, "predict: type must be CSD(1), LFP(2) or BOTH(3)");
    GP_REQUIRE(c->nt > 0 && ntstar == c->nt, -22,
               "predict: len(t)=%d must equal the training nt=%d (the reference's reshape raises ValueError, gpcsd1d.py:279)",
               ntstar, c->nt);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    const PairPlan P = pair_plan(c, hp_ll, hp_pr, z, nz, tstar, ntstar);
    const bool pair = P.pair;
    const bool fold_s = P.fold_s;
    const SymDev sz = P.sz;
    {   // what gpcsd_loglik_parts_wait needs to evaluate the pair again (q_pipe_missed)
        gpcsd_ctx::LlSlot &sl = c->ll_slot[(c->ll_head + c->ll_count) % gpcsd_ctx::LL_SLOTS];
        sl.piped = c->q_pipe;
        sl.hp.set(hp_ll);
        keep_prediction(c, sl.piped, hp_pr, z, nz, tstar, ntstar, type, want_lists != 0);
        sl.pred_seq = c->pred_seq;
    }
    if (!pair) {
        const int rc = loglik_parts_impl(c, hp_ll, nullptr, true);
        if (rc != 0) return rc;
        return predict_impl(c, hp_pr, z, nz, tstar, ntstar, type, want_lists != 0, true);
    }
    GP_REQUIRE(c->time_nt == c->nt, -4, "time grid has %d points but lfp has nt=%d", c->time_nt, c->nt);
    GP_REQUIRE(resident_geo(c).nx == c->nx, -4, "geometry has %d electrodes but lfp has nx=%d", resident_geo(c).nx, c->nx);
    check_hp(c, hp_ll, c->nx);
    check_hp(c, hp_pr, c->nx);
    const gpcsd_hparams *hps[2] = {hp_ll, hp_pr};
    const double jit[2] = {hp_ll->jitter, 0.0};          // no jitter in predict (gpcsd1d.py:258)
    PairFront pf;
    const bool pred_tri = P.pred_tri;
    (void)pred_tri;
    (void)fold_s;
    // the front half: prefetched by the previous call's gpcsd_prefetch_pair (same arguments, nothing in between), or queued now
    bool taken = false;
    if (PairPrefetch *pp = static_cast<PairPrefetch *>(c->pair_prefetch)) {
        if (pp->key == pair_key(c, hps, jit, P)) {
            pf = pp->pf;
            // (device copies of the noise variances: the prefetch's uploads are the same values, the pointers the same buffers)
            taken = true;
            ++c->pair_prefetch_taken;
        }
        pair_prefetch_drop(c);
    }
    if (!taken) front_half_pair_q(c, hps, jit, pf, P);
    const double *Yf = folded_lfp(c, pf.fm[1]);
    const long up0 = c->upload_count;
    double *dzf = c->upload_cached<double>("pred_z", z, (size_t)nz * c->dim);
    double *dtf = c->upload_cached<double>("pred_tstar", tstar, ntstar);
    // (sites or times uploaded just now, on the main stream: the builders that read them stay there)
    // GPCSD_PRELUDE_SIDE: 0 never, 1 (default) when stage 5 is pipelined, 2 always, 3 also for announced pairs
    static const int side_mode = getenv("GPCSD_PRELUDE_SIDE") ? atoi(getenv("GPCSD_PRELUDE_SIDE")) : 1;
    const bool prelude_side = side_mode != 0 && c->upload_count == up0 &&
                              (pf.e[0].pipe_pending || side_mode == 2 || (side_mode == 3 && taken));
    const std::function<void()> ll_pre = [&]() {
        if (pf.e[0].tri) loglik_tri_pre(c, pf.e[0], pf.fm[0], Yf);
    };
    const std::function<void()> ll_tail = [&]() {
        double *Wll = c->buf<double>("proj_W_ll", (size_t)c->nx * c->ntrials * c->nt);
        bool batched = true;
        bool wrote = false;
        if (pf.e[0].tri) wrote = loglik_tri_tail(c, pf.e[0], pf.fm[0], Yf, Wll, next_ll_slot(c));
        else batched = loglik_fold_tail(c, pf.e[0], pf.fm[0], Yf, Wll);
        (void)finish_loglik_async(c, pf.e[0], !batched, wrote);
    };
    // equal temporal hyper-parameters (every loglik -> predict pair of a fit or a bench step: the jitter is spatial): the prediction
    // reads the log-likelihood's X = Y~ Q instead of forming its own from the bit-identical second replica
    // (gpcsd_pair_share_x(ctx, 0, ..) / GPCSD_PAIR_SHARE_X=0: A/B)
    const bool share_x = c->pair_share_x && pf.e[0].tri && pf.e[1].tri && ll_order() == 0 && same_temporal(hp_ll, hp_pr);
    if (share_x) ++c->pair_shared_x_calls;
    // one spatial decomposition (PairFront::share_s): the two projected data sets W~ = diag(U)^T (Y~ Q or Y~) are then the same
    // product whenever both sets take the same form -- tridiagonal with X shared, or both with the temporal eigenvectors
    const bool share_w = pf.share_s && (share_x || (!pf.e[0].tri && !pf.e[1].tri));
    // a pair that queued its own front half is (in a loop) followed by one that does the same while this one's solve runs: the
    // narrow form of the solve leaves room beside it (gram.hip: k_tridiag_solve)
    struct SolvePass {
        gpcsd_ctx *c;
        SolvePass(gpcsd_ctx *cc, int v) : c(cc) { c->solve_pass = v; }
        ~SolvePass() { c->solve_pass = 0; }
    } solve_pass(c, taken ? 64 : 32);
    return predict_fold(c, hp_pr, pf.e[1], pf.fm[1], Yf, sz, dzf, nz, dtf, type, want_lists != 0, true, &ll_tail, &ll_pre,
                        share_x ? "ll_X" : nullptr, prelude_side, share_w ? "proj_W_ll" : nullptr);
    GP_API_END(c)
}

extern "C" int gpcsd_fetch(gpcsd_ctx *c, const char *name, double *host, long count) {
    GP_API_BEGIN(c)
    GP_REQUIRE(name && host && count > 0, -3, "fetch: bad arguments");
    auto it = c->bufs.find(name);
    GP_REQUIRE(it != c->bufs.end() && it->second.p, -2, "fetch: no device buffer named '%s'", name);
    GP_REQUIRE((size_t)count * sizeof(double) <= it->second.bytes, -3, "fetch: '%s' holds %zu bytes, asked for %ld doubles", name,
               it->second.bytes, count);
    // a numerical failure of a preceding asynchronous predict surfaces here -- collected BEFORE the copy: a prediction whose
    // pipelined stage missed its tail is evaluated again by drain_async (q_pipe_missed), and the caller gets that result
    const int rc = drain_async(c);
    c->download(host, it->second.p, (size_t)count * sizeof(double));
    c->sync();
    return rc;
    GP_API_END(c)
}

// The device address of a named result buffer (what gpcsd_fetch copies from), for callers that keep working on the GPU: a
// prediction left in HBM by gpcsd_predict_resident gathered over ranks with RCCL, handed to another library through
// __cuda_array_interface__ / DLPack.  Collects a queued prediction's deferred status first (as gpcsd_fetch does) and drains the
// context's streams: the buffer is complete when the call returns, and stays valid until the next call that writes it.
extern "C" int gpcsd_device_buffer(gpcsd_ctx *c, const char *name, unsigned long long *dev_ptr, unsigned long long *bytes) {
    GP_API_BEGIN(c)
    GP_REQUIRE(name && dev_ptr && bytes, -3, "device_buffer: bad arguments");
    auto it = c->bufs.find(name);
    GP_REQUIRE(it != c->bufs.end() && it->second.p, -2, "device_buffer: no device buffer named '%s'", name);
    const int rc = drain_async(c);
    c->sync();
    *dev_ptr = (unsigned long long)(uintptr_t)it->second.p;
    *bytes = (unsigned long long)it->second.bytes;
    return rc;
    GP_API_END(c)
}

extern "C" int gpcsd_predict(gpcsd_ctx *c, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                             int type, double *csd_list, double *csd, double *lfp_list, double *lfp) {
    GP_API_BEGIN(c)
    const bool want_lists = (csd_list != nullptr) || (lfp_list != nullptr);
    // the caller's arrays are known to the tail: a folded prediction copies its outputs out chunk by chunk under its last product
    gpcsd_ctx::PredSink &sk = c->pred_sink;
    sk = gpcsd_ctx::PredSink();
    // (... into page-locked arrays only -- the class API's come from its pinned pool.  Pageable arrays of another caller are
    // filled through the context's bounce blocks after the product: the runtime never gets to register the caller's pages,
    // ctx.hpp copy_in / copy_out)
    sk.active = true;
    for (const double *p : {csd, csd_list, lfp, lfp_list})
        if (p && !gpcsd_ctx::host_is_pinned(p)) sk.active = false;
    sk.sum[0] = csd; sk.list[0] = csd_list; sk.sum[1] = lfp; sk.list[1] = lfp_list;
    int rc;
    try {
        const bool piped = c->q_pipe;
        rc = predict_impl(c, hp, z, nz, tstar, ntstar, type, want_lists, false);
        if (q_pipe_missed(c, rc, piped)) {         // (a scheduling miss of the pipelined stage 5: again, unpipelined)
            (void)hipStreamSynchronize(c->stream4);
            for (hipEvent_t ev : c->pred_sink_events) c->event_pool.push_back(ev);
            c->pred_sink_events.clear();
            sk.done[0] = sk.done[1] = false;
            rc = predict_impl(c, hp, z, nz, tstar, ntstar, type, want_lists, false);
        }
    } catch (...) {
        sk.active = false;
        (void)hipStreamSynchronize(c->stream4);
        throw;
    }
    sk.active = false;
    if (rc < 0) {
        (void)hipStreamSynchronize(c->stream4);
        return rc;
    }
    const size_t out_elems = (size_t)nz * ntstar * c->ntrials;
    const int C = hp->n_temporal;
    if ((type & 1) && !sk.done[0]) {
        if (csd) c->download(csd, c->bufs["pred_out_csd"].p, out_elems * sizeof(double));
        if (csd_list) c->download(csd_list, c->bufs["pred_out_csd_list"].p, out_elems * C * sizeof(double));
    }
    if ((type & 2) && !sk.done[1]) {
        if (lfp) c->download(lfp, c->bufs["pred_out_lfp"].p, out_elems * sizeof(double));
        if (lfp_list) c->download(lfp_list, c->bufs["pred_out_lfp_list"].p, out_elems * C * sizeof(double));
    }
    if (sk.done[0] || sk.done[1]) GP_HIP(hipStreamSynchronize(c->stream4));
    c->sync();
    for (hipEvent_t ev : c->pred_sink_events) c->event_pool.push_back(ev);
    c->pred_sink_events.clear();
    return rc;
    GP_API_END(c)
}

extern "C" int gpcsd_sample_prior(gpcsd_ctx *c, const gpcsd_hparams *hp, int which, const double *normals, int ntrials,
                                  double *out) {
    GP_API_BEGIN(c)
    const Geo g = resident_geo(c);
    GP_REQUIRE(normals && out && ntrials > 0, -3, "sample_prior: bad arguments");
    GP_REQUIRE(which == GPCSD_PRED_CSD || which == GPCSD_PRED_LFP, -3, "sample_prior: which must be CSD(1) or LFP(2)");
    GP_REQUIRE(c->time_nt > 0, -4, "time grid not set");
    check_hp(c, hp, g.nx);
    const int nx = g.nx, nt = c->time_nt, R = ntrials;
    const long RT = (long)R * nt;
    hipStream_t s = c->stream;
    double *Ks = c->buf<double>("Ks", (size_t)nx * nx);
    double *Kt = c->buf<double>("Kt", (size_t)nt * nt);
    int *st = c->buf<int>("status", 4);
    GP_HIP(hipMemsetAsync(st, 0, 4 * sizeof(int), s));
    const double *t = (const double *)c->bufs["time_t"].p;
    if (which == GPCSD_PRED_CSD) {
        build_ks_csd(c, g, hp->ell_s, Ks, s);                                      // gpcsd1d.py:298
        k_add_diag(c, Ks, nx, hp->jitter, s);
    } else {
        build_kphi(c, g, hp->R, hp->eps, hp->ell_s, nullptr, 0, hp->jitter, Ks, s);   // gpcsd2d.py:346-347
    }
    if (uses_host_kt(hp)) {
        GP_REQUIRE(c->host_kt_nt == nt && (int)c->host_kt.size() == nt * nt, -3, "host temporal Gram does not match nt=%d", nt);
        c->copy_in(Kt, c->host_kt.data(), (size_t)nt * nt * sizeof(double), s);
    } else {
        build_kt(c, hp, t, nt, t, nt, Kt, s);
    }
    potrf_device(c, Kt, nt, st, s);                                                 // Lt
    potrf_device(c, Ks, nx, st, s);                                                 // Ls
    double *stage = c->upload<double>("sp_stage", normals, (size_t)nx * RT);
    double *Z = c->buf<double>("sp_Z", (size_t)nx * RT);
    double *T1 = c->buf<double>("sp_T1", (size_t)nx * RT);
    k_swap_last2(c, stage, Z, nx, nt, R, s);                                        // (x,t,r) -> (x,r,t)
    GemmDesc g1;                          // T1[x'][(r,t)] = sum_x Ls[x'][x] Z[x][(r,t)]
    g1.M = nx; g1.N = (int)RT; g1.K = nx;
    g1.A = Ks; g1.lda = nx; g1.B = Z; g1.ldb = RT; g1.C = T1; g1.ldc = RT;
    g1.prof_name = "gemm_sample_spatial";
    gemm_f64(c, g1, s);
    GemmDesc g2;                          // out[(x',r)][t'] = sum_t T1[(x',r)][t] Lt[t'][t]
    g2.M = nx * R; g2.N = nt; g2.K = nt;
    g2.A = T1; g2.lda = nt; g2.B = Kt; g2.ldb = nt; g2.transB = true; g2.C = Z; g2.ldc = nt;
    g2.prof_name = "gemm_sample_temporal";
    gemm_f64(c, g2, s);
    k_swap_last2(c, Z, stage, nx, R, nt, s);                                        // (x,r,t) -> (x,t,r)
    c->download(out, stage, (size_t)nx * RT * sizeof(double));
    return finish_status(c, st);
    GP_API_END(c)
}
