// Part of capi.hip (included there: one translation unit, so the file-local helpers of capi.hip are in scope) --
// log-likelihood + analytic gradient, one or B hyper-parameter sets per chain of launches.

// ---- log-likelihood + analytic gradient for B hyper-parameter sets in ONE chain of launches ---------------------------------
// fit() restarts are independent optimiser chains (gpcsd1d.py:193-220) whose evaluations are latency-bound: ~100 dependent
// launches in which the longest kernel occupies one workgroup per eigenproblem.  B sets evaluated together share every launch:
// the Gram builders and derivative kernels take the set index as a grid dimension (scalars from a device table of
// hyper-parameters), the eigensolver runs B replicas of each problem class, every GEMM gets an outer batch level.  Each set
// executes exactly the arithmetic of an evaluation on its own (same kernels, same tile configurations, same reduction
// order), so its results do not depend on B.
static HpDev hp_image(const gpcsd_hparams *hp) {
    HpDev h{};
    h.R = hp->R; h.eps = hp->eps; h.ell_s[0] = hp->ell_s[0]; h.ell_s[1] = hp->ell_s[1];
    h.ncomp = hp->n_temporal;
    for (int i = 0; i < hp->n_temporal; ++i) {
        h.kind[i] = hp->kind[i];
        h.ell_t[i] = hp->ell_t[i];
        h.sigma2_t[i] = hp->sigma2_t[i];
    }
    h.sig2n = hp->sig2n[0];
    h.jitter = hp->jitter;
    return h;
}

// out2: (B, 2) = (sum log D, quad) per set; grad: (B, ngrad); status: (B) -- 0 ok, > 0 numerical failure of that set alone.
static int loglik_grad_impl(gpcsd_ctx *c, const gpcsd_hparams *hps, int B, double *out2, double *grad, int ngrad, int *status) {
    GP_REQUIRE(out2 && grad && hps && B >= 1, -3, "loglik_grad: null argument");
    // every argument check comes before any work is queued (the front half launches on two streams)
    const Geo g = resident_geo(c);
    GP_REQUIRE(c->d_lfp != nullptr, -4, "lfp not set (call gpcsd_set_lfp)");
    GP_REQUIRE(c->time_nt == c->nt, -4, "time grid has %d points but lfp has nt=%d", c->time_nt, c->nt);
    GP_REQUIRE(g.nx == c->nx, -4, "geometry has %d electrodes but lfp has nx=%d", g.nx, c->nx);
    const int nx = c->nx, nt = c->nt, R = c->ntrials, C = hps[0].n_temporal, G = g.G();
    const int nsig = hps[0].n_sig2n;
    for (int b = 0; b < B; ++b) {
        check_hp(c, &hps[b], nx);
        // user-defined temporal covariances: the caller supplies d Kt / d theta_k (gpcsd_set_host_temporal_dgram), one set at a time
        GP_REQUIRE(!uses_host_kt(&hps[b]) || (B == 1 && c->host_kt_on && c->host_kt_nt == nt && c->host_dkt_n == 2 * C &&
                                              c->host_dkt.size() == (size_t)2 * C * nt * nt), -3,
                   "loglik_grad: user-defined temporal covariances need their Gram matrix and the %d derivative matrices "
                   "d Kt / d (ell_c, sigma2_c) (gpcsd_set_host_temporal_gram + gpcsd_set_host_temporal_dgram), one set per call", 2 * C);
        GP_REQUIRE(hps[b].n_temporal == C && hps[b].n_sig2n == nsig, -3,
                   "loglik_grad_batch: every hyper-parameter set must have the same number of temporal components and noise entries");
        for (int i = 0; i < C; ++i)
            GP_REQUIRE(hps[b].kind[i] == hps[0].kind[i], -3, "loglik_grad_batch: temporal kernel kinds differ between sets");
    }
    // scalar sig2n: one trailing entry; per-electrode list (indexed by eigen-row like the reference's D): nx entries
    GP_REQUIRE(nsig == 1 || nsig == nx, -3, "loglik_grad: sig2n must be a scalar or a list of nx=%d values (got %d)", nx, nsig);
    const int nhead = 1 + g.dim + 2 * C;
    GP_REQUIRE(ngrad == nhead + nsig, -3, "loglik_grad: ngrad=%d, expected %d", ngrad, nhead + nsig);
    const long RT = (long)R * nt, nxx = (long)nx * nx, ntt = (long)nt * nt, nD = (long)nx * nt, nxRT = (long)nx * RT;
    const long nxG = (long)nx * G, GG = (long)G * G;
    hipStream_t s = c->stream, s2 = c->stream2;

    // ---- device table of the hyper-parameter sets
    std::vector<HpDev> himg(B);
    for (int b = 0; b < B; ++b) himg[b] = hp_image(&hps[b]);
    const HpDev *tab = c->upload_cached<HpDev>("b_hp_tab", himg.data(), B);
    // per-electrode noise lists (fit_gpcsd_baseline.py:85-89): set b's nx variances at d_siglist + b * nx
    const double *d_siglist = nullptr;
    if (nsig > 1) {
        std::vector<double> lists((size_t)nsig * B);
        for (int b = 0; b < B; ++b) memcpy(lists.data() + (size_t)b * nsig, hps[b].sig2n, (size_t)nsig * sizeof(double));
        d_siglist = c->upload_cached<double>("b_sig2n_lists", lists.data(), lists.size());
    }

    double *Ks = c->buf<double>("b_Ks", nxx * B), *Kt = c->buf<double>("b_Kt", ntt * B);
    double *Qs = c->buf<double>("b_Qs", nxx * B), *Qt = c->buf<double>("b_Qt", ntt * B);
    double *es = c->buf<double>("b_es", (size_t)nx * B), *et = c->buf<double>("b_et", (size_t)nt * B);
    double *D = c->buf<double>("b_D", nD * B), *Dinv = c->buf<double>("b_Dinv", nD * B);
    constexpr int NS = 8;                                     // scalars per set: sumlog, quad, sum B^2, sum 1/D
    // everything the host reads back lives in ONE block -- [scalars NS B][gradient slots 64 B][status words 2 B] -- so that it
    // comes back in one copy into a page-locked block (three copies into pageable vectors were three staged round trips, ~70 us)
    const size_t res_doubles = (size_t)(NS + 64 + 1) * B;
    double *resblk = c->buf<double>("b_result", res_doubles);
    double *scal = resblk;
    int *st = reinterpret_cast<int *>(resblk + (size_t)(NS + 64) * B);        // [0, B): spatial chains, [B, 2B): temporal chains
    // 2D: the GL grid is a tensor grid and Kgl = K1 (x) K2 (covariances.py:216).  Forward AND backward then run on the per-axis
    // factors (kron below); 1D keeps the flat Kgl.  GPCSD_GRAD_KRON=0: the flat form in 2D as well (A/B, cross-check).
    static const bool kron_off = getenv("GPCSD_GRAD_KRON") && getenv("GPCSD_GRAD_KRON")[0] == '0';
    const bool kron = g.dim == 2 && !kron_off;
    double *A = c->buf<double>("b_ks_A", nxG * B), *T = c->buf<double>("b_ks_T", nxG * B);
    double *Kgl = kron ? nullptr : c->buf<double>("b_ks_Kgl", GG * B);
    const int n1 = g.ngl1, n2 = g.dim == 2 ? g.ngl2 : 0;
    double *K1 = nullptr, *K2 = nullptr, *dK1 = nullptr, *dK2 = nullptr, *Uk = nullptr, *U2 = nullptr, *Tl1 = nullptr, *Tl2 = nullptr;
    if (kron) {
        K1 = c->buf<double>("b_ks_K1", (size_t)n1 * n1 * B); dK1 = c->buf<double>("b_ks_dK1", (size_t)n1 * n1 * B);
        K2 = c->buf<double>("b_ks_K2", (size_t)n2 * n2 * B); dK2 = c->buf<double>("b_ks_dK2", (size_t)n2 * n2 * B);
        Uk = c->buf<double>("b_ks_U", nxG * B); U2 = c->buf<double>("b_ks_U2", nxG * B);
        Tl1 = c->buf<double>("b_ks_Tl1", nxG * B); Tl2 = c->buf<double>("b_ks_Tl2", nxG * B);
    }
    double *W = c->buf<double>("b_W", nxRT * B), *Bm = c->buf<double>("b_Bm", nxRT * B);
    double *gdev = resblk + (size_t)NS * B;
    const double *t = (const double *)c->bufs["time_t"].p;
    const bool host_kt = uses_host_kt(&hps[0]);
    // (a caller-supplied Gram need not commute with the reflection of the time grid: that side is not folded, cf. front_half)
    const SymDev *sym_s = c->sym_s.ns > 0 ? &c->sym_s : nullptr, *sym_t = (c->sym_t.ns > 0 && !host_kt) ? &c->sym_t : nullptr;
    // Folded basis (see FoldMode): with a scalar noise variance the whole evaluation runs on the half-size eigenvector blocks
    // of the symmetry-folded eigensolver -- projections, the Ghat_s / Ghat_t sums and the back-rotations are each two
    // half-size products.  The cross-parity blocks of Ghat are never needed: dKs and dKt commute with the reflections, so
    // <G, dK> only sees the parity-diagonal blocks.  Half the GEMM flops of the full-size path below.
    // ---- front half: temporal chain on stream2 (queued first: the critical path), spatial chain on the main stream.  Both
    // read the hyper-parameter table uploaded above and report into the status words cleared here: they start behind the
    // main stream's current position (this call returns values, so nothing of it outlives it anyway).
    GP_HIP(hipMemsetAsync(st, 0, (size_t)2 * B * sizeof(int), s));
    begin_generation(c, 1, s2, true);
    begin_generation(c, 0, s, true);
    const FoldMode fm = fold_mode(c, &hps[0]);
    const bool fold = fm.on;
    const double *Yf = fold ? folded_lfp(c, fm) : nullptr;
    // the temporal chain's input as one launch straight from t and the hyper-parameter table (folded, scaled blocks in the class
    // arenas: capi.hip temporal_fill) instead of Gram -> fold -> absmax -> scale, as in the fused calls
    const bool tfill = temporal_fill_applies(c, sym_t, nt, host_kt);
    // what the fills may announce as positive semi-definite (EigArenaView::psd): the signs are the host's to check
    bool variances_nonneg = true, jitters_nonneg = true;
    for (int b = 0; b < B; ++b) {
        jitters_nonneg = jitters_nonneg && hps[b].jitter >= 0.0;
        for (int cc = 0; cc < hps[b].n_temporal; ++cc) variances_nonneg = variances_nonneg && hps[b].sigma2_t[cc] >= 0.0;
    }
    staged_chain_guard(c, s2);            // (a queued staged chain's side-stream readers of the class arenas)
    if (tfill) {
        const char *const *tg = eigh_fold_tags(c, 1);
        const EigArenaView as = eigh_arena_view(c, tg[0], sym_t->ns, B), aa = eigh_arena_view(c, tg[1], sym_t->na, B);
        k_temporal_fold_fill_tab(c, tab, B, t, nt, *sym_t, as, aa, st + B, 1, s2, variances_nonneg);
    } else if (host_kt) {
        c->copy_in(Kt, c->host_kt.data(), (size_t)ntt * sizeof(double), s2);
    } else {
        k_temporal_gram(c, C, nullptr, nullptr, nullptr, t, nt, t, nt, Kt, s2, tab, B, ntt);
    }
    {
        ProfScope ps(c, "eigh_temporal", 9.0 * (double)nt * nt * nt * B, s2);
        eigh_pair_device(c, nullptr, 0, nullptr, nullptr, nullptr, Kt, nt, et, Qt, sym_t, st + B, s2, !fold, B, 1, -1, tfill ? 2 : 0);
    }
    GP_HIP(hipEventRecord(c->ev_join, s2));
    // Ks_b = A_b Kgl_b A_b^T + jitter_b I                     covariances.py:74-96 / :204-232
    if (g.dim == 1) {
        k_fwd_weights_1d(c, g.x, nx, g.gx1, g.gw1, g.ngl1, 0.0, A, s, tab, B, nxG);
        k_se_1d(c, g.gx1, G, g.gx1, G, 0.0, Kgl, s, tab, B, GG);
    } else {
        k_fwd_weights_2d(c, g.x, nx, g.gx1, g.gw1, g.ngl1, g.gx2, g.gw2, g.ngl2, 0.0, 0.0, A, s, tab, B, nxG);
        if (!kron) k_se_2d(c, g.gx1, g.gx2, G, g.ngl2, g.gx1, g.gx2, G, g.ngl2, 0.0, 0.0, Kgl, s, tab, B, GG);
    }
    // out[x][(h1,h2)] = sum_g1 F1[g1][h1] V[x][(g1,h2)], one small product per electrode and set (F1 = K1 or dK1)
    auto kron_axis1 = [&](const double *F1, const double *V, double *out, const char *name) {
        GemmDesc v;
        v.M = n1; v.N = n2; v.K = n1;
        v.A = F1; v.lda = n1; v.transA = true; v.B = V; v.ldb = n2; v.C = out; v.ldc = n2;
        v.batch = nx; v.sA = 0; v.sB = G; v.sC = G;
        v.batch2 = B; v.sA2 = (long)n1 * n1; v.sB2 = nxG; v.sC2 = nxG;
        v.prof_name = name;
        gemm_f64(c, v, s);
    };
    // out[(x,g1)][h2] = sum_g2 A[(x,g1)][g2] F2[g2][h2] (F2 = K2 or dK2)
    auto kron_axis2 = [&](const double *F2, double *out, const char *name) {
        GemmDesc u;
        u.M = nx * n1; u.N = n2; u.K = n2;
        u.A = A; u.lda = n2; u.B = F2; u.ldb = n2; u.C = out; u.ldc = n2;
        u.batch2 = B; u.sA2 = nxG; u.sB2 = (long)n2 * n2; u.sC2 = nxG;
        u.prof_name = name;
        gemm_f64(c, u, s);
    };
    {
        if (kron) {
            // T = A (K1 (x) K2) as two small products (build_kphi does the same for the fused calls: 74 MF instead of 1.1 GF at
            // 384 x 20 x 60, and Kgl's 1200^2 exponentials are never formed)
            k_se_axis_tab(c, g.gx1, n1, 0, tab, B, K1, dK1, s);
            k_se_axis_tab(c, g.gx2, n2, 1, tab, B, K2, dK2, s);
            kron_axis2(K2, Uk, "gemm_Ks_AK2");
            kron_axis1(K1, Uk, T, "gemm_Ks_K1U");
        } else {
            GemmDesc d1;                                   // T = A Kgl
            d1.M = nx; d1.N = G; d1.K = G;
            d1.A = A; d1.lda = G; d1.B = Kgl; d1.ldb = G; d1.C = T; d1.ldc = G;
            d1.batch2 = B; d1.sA2 = nxG; d1.sB2 = GG; d1.sC2 = nxG;
            d1.prof_name = "gemm_Ks_AKgl";
            gemm_f64(c, d1, s);
        }
        GemmDesc d2;                                   // Ks = T A^T
        d2.M = nx; d2.N = nx; d2.K = G;
        d2.A = T; d2.lda = G; d2.B = A; d2.ldb = G; d2.transB = true; d2.C = Ks; d2.ldc = nx;
        d2.batch2 = B; d2.sA2 = nxG; d2.sB2 = nxG; d2.sC2 = nxx;
        d2.prof_name = "gemm_Ks_TAt";
        gemm_f64(c, d2, s);
    }
    {
        // the spatial chain's input the same way (psd fold fill: fold + jitter on the folded diagonals + scale in one launch)
        const bool sfill = spatial_fill_applies(c, sym_s, nx);
        if (sfill) {
            const char *const *tg = eigh_fold_tags(c, 0);
            const EigArenaView as = eigh_arena_view(c, tg[0], sym_s->ns, B), aa = eigh_arena_view(c, tg[1], sym_s->na, B);
            k_psd_fold_fill(c, Ks, nx, nxx, B, nullptr, *sym_s, as, aa, st, 1, s, tab, jitters_nonneg);
        } else {
            k_add_diag(c, Ks, nx, 0.0, s, tab, B, nxx);
        }
        ProfScope ps(c, "eigh_spatial", 9.0 * (double)nx * nx * nx * B, s);
        eigh_pair_device(c, Ks, nx, es, Qs, sym_s, nullptr, 0, nullptr, nullptr, nullptr, st, s, !fold, B, 1, -1, sfill ? 1 : 0);
    }
    if (kron) {
        // the backward pass's hyper-parameter-only factors, queued here where the main stream would otherwise wait for the chains:
        // Tl1 = A (dK1 (x) K2) = dK1^T (A K2),  Tl2 = A (K1 (x) dK2) = K1^T (A dK2)
        kron_axis1(dK1, Uk, Tl1, "gemm_grad_dK1U");
        kron_axis2(dK2, U2, "gemm_grad_AdK2");
        kron_axis1(K1, U2, Tl2, "gemm_grad_K1U2");
    }
    double *av = c->buf<double>("b_grad_a", (size_t)nx * B), *bv = c->buf<double>("b_grad_b", (size_t)nt * B);
    double *Gs = c->buf<double>("b_grad_Gs", nxx * B), *Gt = c->buf<double>("b_grad_Gt", ntt * B);
    const long nmx = (long)std::max(nx, nt) * std::max(nx, nt);
    double *T1 = c->buf<double>("b_grad_T1", (size_t)nmx * B);
    // row chunk of the Ghat_t sums (GPCSD_GRAD_CH: A/B).  A chunk is one K range of a 64 x 64 tile.  Measured: 256-row chunks make the
    // product itself faster for ONE set at 384 x 500 x 50 (592 tiles of 32 dependent K steps are too few to hide the load latency:
    // 94 -> 61 us per parity) but the evaluation no faster (the product runs beside the spatial branch), and every batch slower (twice
    // the partial matrices written and read back: 8 sets 4.41 against 4.17 ms; cfg5 at 32 sets 7.8 k against 8.5 k evaluations/s).
    // 512 it stays; the choice never depends on the batch: a set is summed the same way alone and in a batch.
    static const int CH = getenv("GPCSD_GRAD_CH") ? std::max(64, atoi(getenv("GPCSD_GRAD_CH"))) : 512;
    static const int GS_CFG = getenv("GPCSD_GRAD_GS_CFG") ? atoi(getenv("GPCSD_GRAD_GS_CFG")) : 3;
    static const int GT_CFG = getenv("GPCSD_GRAD_GT_CFG") ? atoi(getenv("GPCSD_GRAD_GT_CFG")) : 3;
    // tile configuration of the mid-size products of the tail (Gs A, Gs T: nx x G x nx; the rotations U Ghat U^T from 128 rows): the
    // automatic choice looks at ONE set's tile count (it must not depend on the batch) and takes the 32 x 32 latency tile, which at
    // a batch of 8 sets is 13 % of the GPU time at a sixth of the MFMA rate.  Default 3 (64 x 64, BK 16): 8 sets 4.30 -> 4.05 ms, one set unchanged.  GPCSD_GRAD_MID_CFG=0: automatic (A/B).
    static const int MID_CFG = getenv("GPCSD_GRAD_MID_CFG") ? atoi(getenv("GPCSD_GRAD_MID_CFG")) : 3;
    hipStream_t sT = s;                   // the stream of the gradient's temporal half (folded path: stream2, see below)
    bool quad_in_two = false;             // the quadratic form came out as two partial sums (parity blocks of unequal shape)
    if (fold) {
        ++c->fold_gemm_calls;
        // a side that is not folded takes part as one "symmetric" block of full size (identity fold, U = Q, w = eigenvalues)
        struct Side {
            int n, ns, na;
            const double *U, *w;
            long sU, sw;
            SymDev sym;
        } S_, T_;
        auto side = [&](int slot, const FoldView &fv1, const SymDev &sym, int n, const double *Q, const double *ev) {
            Side sd;
            sd.n = n;
            if (fv1.on) {
                const FoldView fv = eigh_fold_view(c, slot, slot ? &c->sym_t : &c->sym_s, n, B);
                sd.ns = fv.ns; sd.na = fv.na; sd.U = fv.U; sd.w = fv.w; sd.sU = fv.sU; sd.sw = fv.sw;
            } else {
                sd.ns = n; sd.na = 0; sd.U = Q; sd.w = ev; sd.sU = (long)n * n; sd.sw = n;
            }
            sd.sym = sym;
            return sd;
        };
        S_ = side(0, fm.fs, fm.sym_s, nx, Qs, es);
        T_ = side(1, fm.ft, fm.sym_t, nt, Qt, et);
        const long sUs = (long)S_.ns * S_.ns + (long)S_.na * S_.na, sUt = (long)T_.ns * T_.ns + (long)T_.na * T_.na;
        // W~_b = diag(U_b)^T Y~ : the data is folded once per geometry and shared by all sets
        for (int p = 0; p < 2; ++p) {
            const int np = p ? S_.na : S_.ns;
            const long r0 = p ? S_.ns : 0;
            if (np == 0) continue;
            GemmDesc gw;
            gw.M = np; gw.N = (int)RT; gw.K = np;
            gw.A = S_.U + (p ? (long)S_.ns * S_.ns : 0); gw.lda = np; gw.transA = true;
            gw.B = Yf + r0 * RT; gw.ldb = RT; gw.C = W + r0 * RT; gw.ldc = RT;
            gw.batch2 = B; gw.sA2 = S_.sU; gw.sB2 = 0; gw.sC2 = nxRT;
            gw.prof_name = "gemm_proj_spatial";
            gemm_f64(c, gw, s);
        }
        GP_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
        // D~_b = ws_b (x) wt_b + sig2n_b in fold order, sum log D_b -> scal[b][0]
        k_build_D(c, S_.w, nx, T_.w, nt, nullptr, 1, D, Dinv, scal, s, tab, B, NS);
        {   // alpha~ = W~ V (per temporal parity block);  B~ = alpha~ / D~, B~ wt, B~ ws;  sums of alpha~ B~ and B~^2
            GemmDesc gq[2];
            for (int q = 0; q < 2; ++q) {
                const int nq = q ? T_.na : T_.ns, c0 = q ? T_.ns : 0;
                gq[q].M = nx * R; gq[q].N = nq; gq[q].K = nq;
                gq[q].A = W + c0; gq[q].lda = nt; gq[q].B = T_.U + (q ? (long)T_.ns * T_.ns : 0); gq[q].ldb = nq;
                // (no B~ wt / B~ ws copies: the two products below scale B~ along their contracted index as they load it --
                // GemmDesc::kscale -- the same rounded products, without 2 x 0.6 GB written and read back per 32-set batch)
                gq[q].C = Bm + c0; gq[q].ldc = nt; gq[q].C2 = nullptr; gq[q].C3 = nullptr;
                gq[q].epi = EPI_GRAD; gq[q].D = Dinv + c0; gq[q].rdiv = R; gq[q].ldd = nt;
                gq[q].colscale = T_.w + c0; gq[q].rowscale = S_.w;
                gq[q].quad_out = scal + 1 + 3 * q;        // scal[b][1], [2] (first block or both), scal[b][4], [5] (second block)
                gq[q].batch2 = B; gq[q].sA2 = nxRT; gq[q].sB2 = T_.sU; gq[q].sC2 = nxRT; gq[q].sD2 = nD; gq[q].sColscale2 = nt;
                gq[q].sRowscale2 = nx; gq[q].sQuad2 = NS;
                gq[q].prof_name = "gemm_grad_temporal";
            }
            if (T_.na > 0 && T_.na == T_.ns) {           // equal parity blocks: one launch, one sum over both
                gq[0].batch = 2;
                gq[0].sA = gq[1].A - gq[0].A; gq[0].sB = gq[1].B - gq[0].B; gq[0].sC = gq[1].C - gq[0].C;
                gq[0].sD = gq[1].D - gq[0].D; gq[0].sColscale = gq[1].colscale - gq[0].colscale;
                gemm_f64(c, gq[0], s);
            } else {
                gemm_f64(c, gq[0], s);
                if (T_.na > 0) {
                    gemm_f64(c, gq[1], s);
                    quad_in_two = true;
                }
            }
        }
        k_D_sums(c, D, S_.w, T_.w, nx, nt, av, bv, scal + 3, s, B, NS);   // a, b in fold order; scal[b][3] = sum 1/D
        // From here the spatial and the temporal half of the gradient are independent (both read B~, a, b): the temporal one --
        // Ghat_t, its rotation back, <Gt, dKt> -- runs on stream2, idle since its chain ended, beside the spatial one on the main
        // stream; each branch's small launches (reductions, the rotations' 8 us products) hide under the other's large products.
        // GPCSD_GRAD_BRANCHES=0: one after the other on the main stream (A/B; same kernels, same bits).
        static const bool branches_off = getenv("GPCSD_GRAD_BRANCHES") && getenv("GPCSD_GRAD_BRANCHES")[0] == '0';
        if (!branches_off && c->prof_mode != 1) {
            sT = s2;
            hipEvent_t ev = c->get_event();
            GP_HIP(hipEventRecord(ev, s));
            GP_HIP(hipStreamWaitEvent(sT, ev, 0));
            c->event_pool.push_back(ev);
        }
        // Ghat_t~ parity blocks: 1/2 sum_{(x,r)} (B~ ws)[:, q]^T B~[:, q] - R/2 diag(b[q block])   (row chunks, then a fixed-order sum)
        const long rows = (long)nx * R;
        double *wsr = c->buf<double>("b_grad_ws_rows", (size_t)rows * B);        // ws spread over the (x', r) rows
        k_repeat_rows(c, S_.w, nx, nx, R, B, wsr, sT);
        const int nfull = (int)(rows / CH), rem = (int)(rows % CH), nchunk = nfull + (rem > 0 ? 1 : 0);
        const long sCt = (long)nchunk * sUt;
        double *Ct = c->buf<double>("b_grad_Ct", (size_t)sCt * B);
        double *Ght = c->buf<double>("b_grad_Ght", (size_t)sUt * B);
        for (int q = 0; q < 2; ++q) {
            const int nq = q ? T_.na : T_.ns, c0 = q ? T_.ns : 0;
            const long o_in = q ? (long)nchunk * T_.ns * T_.ns : 0, o_out = q ? (long)T_.ns * T_.ns : 0, nqq = (long)nq * nq;
            if (nq == 0) continue;
            if (nfull > 0) {
                GemmDesc gt;
                gt.M = nq; gt.N = nq; gt.K = CH;
                gt.A = Bm + c0; gt.lda = nt; gt.transA = true; gt.B = Bm + c0; gt.ldb = nt; gt.C = Ct + o_in; gt.ldc = nq;
                gt.kscale = wsr; gt.sKscale = CH; gt.sKscale2 = rows;         // (B~ ws)^T B~: the factor runs along the contracted row
                gt.batch = nfull; gt.sA = (long)CH * nt; gt.sB = (long)CH * nt; gt.sC = nqq;
                gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
                if (nq >= 64) gt.cfg = GT_CFG;
                gt.lower = true;
                gt.prof_name = "gemm_grad_Gt";
                gemm_f64(c, gt, sT);
            }
            if (rem > 0) {
                GemmDesc gt;
                gt.M = nq; gt.N = nq; gt.K = rem;
                if (nq >= 64) gt.cfg = GT_CFG;
                gt.lower = true;
                gt.A = Bm + (long)nfull * CH * nt + c0; gt.lda = nt; gt.transA = true; gt.B = Bm + (long)nfull * CH * nt + c0; gt.ldb = nt;
                gt.kscale = wsr + (long)nfull * CH; gt.sKscale2 = rows;
                gt.C = Ct + o_in + (long)nfull * nqq; gt.ldc = nq;
                gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
                gt.prof_name = "gemm_grad_Gt";
                gemm_f64(c, gt, sT);
            }
            k_batch_reduce(c, Ct + o_in, nchunk, nqq, nq, 0.5, bv + c0, -0.5 * R, Ght + o_out, sT, B, sCt, nt, sUt);
        }
        // Ghat_s~ parity blocks: 1/2 sum_r (B~ wt)[p rows] B~[p rows]^T - R/2 diag(a[p rows])
        const long sCs = (long)R * sUs;
        double *Cs = c->buf<double>("b_grad_Cs", (size_t)sCs * B);
        double *Ghs = c->buf<double>("b_grad_Ghs", (size_t)sUs * B);
        for (int p = 0; p < 2; ++p) {
            const int np = p ? S_.na : S_.ns;
            const long r0 = p ? S_.ns : 0, o_in = p ? (long)R * S_.ns * S_.ns : 0, o_out = p ? (long)S_.ns * S_.ns : 0;
            if (np == 0) continue;
            GemmDesc gs;
            gs.M = np; gs.N = np; gs.K = nt;
            gs.A = Bm + r0 * RT; gs.lda = RT; gs.B = Bm + r0 * RT; gs.ldb = RT; gs.transB = true; gs.C = Cs + o_in; gs.ldc = np;
            gs.kscale = T_.w; gs.sKscale = 0; gs.sKscale2 = nt;           // (B~ wt) B~^T: the factor runs along the contracted t'
            gs.batch = R; gs.sA = nt; gs.sB = nt; gs.sC = (long)np * np;
            gs.batch2 = B; gs.sA2 = nxRT; gs.sB2 = nxRT; gs.sC2 = sCs;
            // the tile configuration must not depend on B (a set has to run the same tiles alone or in a batch): these
            // half-size products have few tiles per set, which the automatic choice would read as "latency-bound"
            if (np >= 64) gs.cfg = GS_CFG;
            gs.lower = true;                                              // symmetric: k_batch_reduce mirrors the lower triangle
            gs.prof_name = "gemm_grad_Gs";
            gemm_f64(c, gs, s);
            k_batch_reduce(c, Cs + o_in, R, (long)np * np, np, 0.5, av + r0, -0.5 * R, Ghs + o_out, s, B, sCs, nx, sUs);
        }
        // back to the original bases, block by block: G~_pp = U_p Ghat_pp U_p^T, then G = F^T diag(G~_ss, G~_aa) F
        double *Gsf = c->buf<double>("b_grad_Gsf", (size_t)sUs * B), *Gtf = c->buf<double>("b_grad_Gtf", (size_t)sUt * B);
        double *T1t = c->buf<double>("b_grad_T1t", (size_t)nmx * B);          // (the temporal branch's own scratch)
        auto sandwich_blocks = [&](const Side &sd, const double *H, long sH, double *outf, double *tmp, hipStream_t sq) {
            for (int p = 0; p < 2; ++p) {
                const int np = p ? sd.na : sd.ns;
                const long o = p ? (long)sd.ns * sd.ns : 0;
                if (np == 0) continue;
                GemmDesc a;
                a.M = np; a.N = np; a.K = np; a.A = sd.U + o; a.lda = np; a.B = H + o; a.ldb = np; a.C = tmp; a.ldc = np;
                a.batch2 = B; a.sA2 = sd.sU; a.sB2 = sH; a.sC2 = nmx;
                if (np >= 128) a.cfg = MID_CFG;
                a.prof_name = "gemm_grad_sandwich";
                gemm_f64(c, a, sq);
                GemmDesc bq;
                bq.M = np; bq.N = np; bq.K = np; bq.A = tmp; bq.lda = np; bq.B = sd.U + o; bq.ldb = np; bq.transB = true;
                bq.C = outf + o; bq.ldc = np;
                bq.batch2 = B; bq.sA2 = nmx; bq.sB2 = sd.sU; bq.sC2 = sH;
                if (np >= 128) bq.cfg = MID_CFG;
                bq.prof_name = "gemm_grad_sandwich";
                gemm_f64(c, bq, sq);
            }
        };
        sandwich_blocks(T_, Ght, sUt, Gtf, T1t, sT);
        k_sym_unfold_mat(c, Gtf, sUt, T_.sym, nt, Gt, sT, B);
        sandwich_blocks(S_, Ghs, sUs, Gsf, T1, s);
        k_sym_unfold_mat(c, Gsf, sUs, S_.sym, nx, Gs, s, B);
    } else {
        GemmDesc g1;                          // W_b = Qs_b^T Y          (gpcsd1d.py:125 inner dot; the data is shared)
        g1.M = nx; g1.N = (int)RT; g1.K = nx;
        g1.A = Qs; g1.lda = nx; g1.transA = true; g1.B = c->d_lfp; g1.ldb = RT; g1.C = W; g1.ldc = RT;
        g1.batch2 = B; g1.sA2 = nxx; g1.sB2 = 0; g1.sC2 = nxRT;
        g1.prof_name = "gemm_proj_spatial";
        gemm_f64(c, g1, s);
        GP_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
        // D_b = es_b (x) et_b + sig2n_b, sum log D_b -> scal[b][0]
        k_build_D(c, es, nx, et, nt, d_siglist, nsig, D, Dinv, scal, s, tab, B, NS);
        GemmDesc g2;                          // alpha = W Qt;  B = alpha / D, B*et, B*es;  sum alpha*B, sum B^2
        g2.M = nx * R; g2.N = nt; g2.K = nt;
        g2.A = W; g2.lda = nt; g2.B = Qt; g2.ldb = nt; g2.C = Bm; g2.ldc = nt; g2.C2 = nullptr; g2.C3 = nullptr;   // (see the folded path)
        g2.epi = EPI_GRAD; g2.D = Dinv; g2.rdiv = R; g2.ldd = nt; g2.colscale = et; g2.rowscale = es;
        g2.quad_out = scal + 1;               // scal[b][1] = quad, scal[b][2] = sum B^2
        g2.batch2 = B; g2.sA2 = nxRT; g2.sB2 = ntt; g2.sC2 = nxRT; g2.sD2 = nD; g2.sColscale2 = nt; g2.sRowscale2 = nx; g2.sQuad2 = NS;
        g2.prof_name = "gemm_grad_temporal";
        gemm_f64(c, g2, s);
        k_D_sums(c, D, es, et, nx, nt, av, bv, scal + 3, s, B, NS);           // scal[b][3] = sum 1/D

        // Ghat_s = 1/2 sum_r (B_r et) B_r^T - R/2 diag(a)      (one GEMM per trial, batched; then a fixed-order sum)
        const long sCs = (long)R * nxx;
        double *Cs = c->buf<double>("b_grad_Cs", (size_t)sCs * B);
        GemmDesc gs;
        gs.M = nx; gs.N = nx; gs.K = nt;
        gs.A = Bm; gs.lda = RT; gs.B = Bm; gs.ldb = RT; gs.transB = true; gs.C = Cs; gs.ldc = nx;
        gs.kscale = et; gs.sKscale = 0; gs.sKscale2 = nt;
        gs.batch = R; gs.sA = nt; gs.sB = nt; gs.sC = nxx;
        gs.batch2 = B; gs.sA2 = nxRT; gs.sB2 = nxRT; gs.sC2 = sCs;
        gs.lower = true;                                                  // symmetric: k_batch_reduce mirrors the lower triangle
        gs.prof_name = "gemm_grad_Gs";
        gemm_f64(c, gs, s);
        double *Ghs = c->buf<double>("b_grad_Ghs", nxx * B);
        k_batch_reduce(c, Cs, R, nxx, nx, 0.5, av, -0.5 * R, Ghs, s, B, sCs);
        if (nsig > 1) {
            // noise tied to the eigen-index: eigenvector-rotation term, S = sum_r B_r B_r^T (see grad.hip)
            GemmDesc g3 = gs;
            g3.A = Bm;
            g3.kscale = nullptr;
            g3.prof_name = "gemm_grad_BBt";
            gemm_f64(c, g3, s);
            double *Ssum = c->buf<double>("grad_Ssum", (size_t)nxx * B);
            double *zero = c->buf<double>("grad_zero", nx);
            k_fill(c, zero, nx, 0.0, s);
            k_batch_reduce(c, Cs, R, nxx, nx, 1.0, zero, 0.0, Ssum, s, B, sCs, /*s_dvec=*/0, nxx);
            k_siglist_eigvec_term(c, Ghs, Ssum, es, d_siglist, nx, 0.0, s, B);
        }
        // Ghat_t = 1/2 sum_{(x,r)} (B es)^T B - R/2 diag(b)    (row chunks of 512, batched; remainder separately)
        const long rows = (long)nx * R;
        double *wsr = c->buf<double>("b_grad_ws_rows", (size_t)rows * B);        // es spread over the (x, r) rows
        k_repeat_rows(c, es, nx, nx, R, B, wsr, s);
        const int nfull = (int)(rows / CH), rem = (int)(rows % CH);
        const long sCt = (long)(nfull + 1) * ntt;
        double *Ct = c->buf<double>("b_grad_Ct", (size_t)sCt * B);
        if (nfull > 0) {
            GemmDesc gt;
            gt.M = nt; gt.N = nt; gt.K = CH;
            gt.A = Bm; gt.lda = nt; gt.transA = true; gt.B = Bm; gt.ldb = nt; gt.C = Ct; gt.ldc = nt;
            gt.kscale = wsr; gt.sKscale = CH; gt.sKscale2 = rows;
            gt.batch = nfull; gt.sA = (long)CH * nt; gt.sB = (long)CH * nt; gt.sC = ntt;
            gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
            gt.lower = true;
            gt.prof_name = "gemm_grad_Gt";
            gemm_f64(c, gt, s);
        }
        if (rem > 0) {
            GemmDesc gt;
            gt.lower = true;
            gt.M = nt; gt.N = nt; gt.K = rem;
            gt.A = Bm + (long)nfull * CH * nt; gt.lda = nt; gt.transA = true; gt.B = Bm + (long)nfull * CH * nt; gt.ldb = nt;
            gt.kscale = wsr + (long)nfull * CH; gt.sKscale2 = rows;
            gt.C = Ct + (long)nfull * ntt; gt.ldc = nt;
            gt.batch2 = B; gt.sA2 = nxRT; gt.sB2 = nxRT; gt.sC2 = sCt;
            gt.prof_name = "gemm_grad_Gt";
            gemm_f64(c, gt, s);
        }
        double *Ght = c->buf<double>("b_grad_Ght", ntt * B);
        k_batch_reduce(c, Ct, nfull + (rem > 0 ? 1 : 0), ntt, nt, 0.5, bv, -0.5 * R, Ght, s, B, sCt);
        // back to the original bases: Gs = Qs Ghat_s Qs^T, Gt = Qt Ghat_t Qt^T
        auto sandwich = [&](const double *Q, const double *H, int n, double *out) {
            const long nn = (long)n * n;
            GemmDesc a;
            a.M = n; a.N = n; a.K = n; a.A = Q; a.lda = n; a.B = H; a.ldb = n; a.C = T1; a.ldc = n;
            a.batch2 = B; a.sA2 = nn; a.sB2 = nn; a.sC2 = nmx;
            a.prof_name = "gemm_grad_sandwich";
            gemm_f64(c, a, s);
            GemmDesc bq;
            bq.M = n; bq.N = n; bq.K = n; bq.A = T1; bq.lda = n; bq.B = Q; bq.ldb = n; bq.transB = true; bq.C = out; bq.ldc = n;
            bq.batch2 = B; bq.sA2 = nmx; bq.sB2 = nn; bq.sC2 = nn;
            bq.prof_name = "gemm_grad_sandwich";
            gemm_f64(c, bq, s);
        };
        sandwich(Qs, Ghs, nx, Gs);
        sandwich(Qt, Ght, nt, Gt);
    }
    // natural-parameter order: [R, ell_s (dim), (ell_t, sigma2_t) per component, sig2n]; 64 slots per set
    if (host_kt) {                        // <Gt, d Kt / d theta_k> with the caller's derivative matrices
        double *dK = c->upload<double>("b_host_dkt", c->host_dkt.data(), (size_t)2 * C * ntt);
        if (sT != s) {                    // (the upload is on the main stream)
            hipEvent_t ev = c->get_event();
            GP_HIP(hipEventRecord(ev, s));
            GP_HIP(hipStreamWaitEvent(sT, ev, 0));
            c->event_pool.push_back(ev);
        }
        k_frob_inner(c, Gt, dK, ntt, 2 * C, gdev + 1 + g.dim, sT);
    } else {
        k_temporal_grad(c, &hps[0], Gt, t, nt, gdev + 1 + g.dim, sT, tab, B, 64);
    }
    hipEvent_t ev_tbranch = nullptr;
    if (sT != s) {                        // the temporal branch ends here; the main stream takes it in before the results are copied
        ev_tbranch = c->get_event();
        GP_HIP(hipEventRecord(ev_tbranch, sT));
    }
    double *P = c->buf<double>("b_grad_P", nxG * B);
    GemmDesc gp;                          // P = Gs A
    gp.M = nx; gp.N = G; gp.K = nx; gp.A = Gs; gp.lda = nx; gp.B = A; gp.ldb = G; gp.C = P; gp.ldc = G;
    gp.batch2 = B; gp.sA2 = nxx; gp.sB2 = nxG; gp.sC2 = nxG;
    if (nx >= 128) gp.cfg = MID_CFG;
    gp.prof_name = "gemm_grad_GsA";
    gemm_f64(c, gp, s);
    if (kron) {                           // <A^T Gs A, dKgl/dell_k> = <Gs A, Tl_k>   (grad.hip: k_frob_pair)
        k_frob_pair(c, P, Tl1, Tl2, nxG, gdev + 1, s, B, 64);
    } else {
        double *Mg = c->buf<double>("b_grad_M", GG * B);
        GemmDesc gm;                      // M = A^T P
        gm.M = G; gm.N = G; gm.K = nx; gm.A = A; gm.lda = G; gm.transA = true; gm.B = P; gm.ldb = G; gm.C = Mg; gm.ldc = G;
        gm.batch2 = B; gm.sA2 = nxG; gm.sB2 = nxG; gm.sC2 = GG;
        gm.prof_name = "gemm_grad_AtP";
        gemm_f64(c, gm, s);
        k_kgl_grad(c, Mg, Kgl, g.gx1, g.gx2, G, g.dim == 2 ? g.ngl2 : 0, 0.0, 0.0, gdev + 1, s, tab, B, 64);
    }
    GemmDesc gr;                          // S = Gs T  (T = A Kgl from the forward pass)
    gr.M = nx; gr.N = G; gr.K = nx; gr.A = Gs; gr.lda = nx; gr.B = T; gr.ldb = G; gr.C = P; gr.ldc = G;
    gr.batch2 = B; gr.sA2 = nxx; gr.sB2 = nxG; gr.sC2 = nxG;
    if (nx >= 128) gr.cfg = MID_CFG;
    gr.prof_name = "gemm_grad_GsT";
    gemm_f64(c, gr, s);
    k_fwdR_grad(c, P, g.x, nx, g.gx1, g.gw1, g.gx2, g.gw2, G, g.dim == 2 ? g.ngl2 : 0, 0.0, 0.0, gdev, s, tab, B, 64);
    std::vector<double> hb2, hinv;
    if (nsig > 1) {                       // d/d sig2n_x = -R/2 sum_i 1/D_xi + 1/2 sum_{r,i} B_{(x,r),i}^2
        double *b2row = c->buf<double>("grad_b2row", (size_t)nx * B);               // (Bm is [set][x][r][t]: nx * B rows of R nt)
        k_rowgroup_sumsq(c, Bm, nx * B, RT, b2row, s);
        hb2.resize((size_t)nx * B);
        hinv.resize((size_t)nx * B);
        c->download(hb2.data(), b2row, hb2.size() * sizeof(double));
        c->download(hinv.data(), c->bufs["grad_s1row"].p, hinv.size() * sizeof(double));   // written by k_D_sums, [set][x]
    }
    if (ev_tbranch) {
        GP_HIP(hipStreamWaitEvent(s, ev_tbranch, 0));
        c->event_pool.push_back(ev_tbranch);
    }
    double *hres = c->pinned<double>("b_result_host", res_doubles);
    c->download(hres, resblk, res_doubles * sizeof(double));
    GP_HIP(hipStreamSynchronize(s2));
    c->sync();
    const double *hs = hres, *hg = hres + (size_t)NS * B;
    const int *hst = reinterpret_cast<const int *>(hres + (size_t)(NS + 64) * B);
    if (c->prof_mode == 1) c->prof_collect();
    int worst = 0;
    for (int b = 0; b < B; ++b) {
        out2[2 * b] = hs[(size_t)NS * b];
        out2[2 * b + 1] = hs[(size_t)NS * b + 1] + (quad_in_two ? hs[(size_t)NS * b + 4] : 0.0);
        double *gb = grad + (size_t)b * ngrad;
        for (int k = 0; k < nhead; ++k) gb[k] = hg[(size_t)64 * b + k];
        if (nsig == 1)
            gb[nhead] = -0.5 * R * hs[(size_t)NS * b + 3] + 0.5 * (hs[(size_t)NS * b + 2] + (quad_in_two ? hs[(size_t)NS * b + 5] : 0.0));
        else
            for (int x = 0; x < nx; ++x) gb[nhead + x] = -0.5 * R * hinv[(size_t)b * nx + x] + 0.5 * hb2[(size_t)b * nx + x];
        int stb = hst[b] != 0 ? hst[b] : hst[B + b];
        if (stb < 0) stb = 1;
        if (status) status[b] = stb;
        worst = std::max(worst, stb);
    }
    if (worst != 0) {
        char msg[160];
        snprintf(msg, sizeof(msg), "numerical failure (status %d): eigensolver did not converge or matrix not positive definite", worst);
        c->last_error = msg;
    }
    return worst;
}

extern "C" int gpcsd_loglik_grad(gpcsd_ctx *c, const gpcsd_hparams *hp, double *out2, double *grad, int ngrad) {
    GP_API_BEGIN(c)
    GP_REQUIRE(hp != nullptr, -3, "loglik_grad: null hparams");
    if (int rc = drain_async(c)) return rc;      // (this path keeps its own status words)
    return loglik_grad_impl(c, hp, 1, out2, grad, ngrad, nullptr);
    GP_API_END(c)
}

extern "C" int gpcsd_loglik_grad_batch(gpcsd_ctx *c, const gpcsd_hparams *hps, int nsets, double *out2, double *grad, int ngrad,
                                       int *status) {
    GP_API_BEGIN(c)
    GP_REQUIRE(hps && nsets >= 1 && status, -3, "loglik_grad_batch: bad arguments");
    if (int rc = drain_async(c)) return rc;
    (void)loglik_grad_impl(c, hps, nsets, out2, grad, ngrad, status);   // per-set failures are reported in status[], not as rc
    return 0;
    GP_API_END(c)
}
