// Part of capi.hip (included there: one translation unit) --
// multi-device entry points of the C ABI (gpcsd_dist_create / _loglik / _predict / _fit_restarts): ONE process
// driving one context per device.  The Python layer shards with one process per GPU over torch.distributed (RCCL; dist.py);
// a C caller without a launcher gets the same partition here.  No device-to-device exchange exists on this path (SURVEY
// 8(e)): trials are independent, every device recomputes the deterministic decompositions, and what is combined is one double
// per device (the partial quadratic term) -- summed on the host, in device order, so the result does not depend on timing.

#include <memory>
#include <thread>

struct gpcsd_dist {
    std::vector<gpcsd_ctx *> ctx;
    int nx = 0, nt = 0, ntrials = 0;            // shape of the data handed to gpcsd_dist_set_lfp
    bool replicated = false;                    // every device holds all trials (restart sharding) instead of a block of them
    std::vector<int> first, count;              // block of trials of device i (trial sharding)
    std::string last_error;
};

static int dist_fail(gpcsd_dist *d, int rc, int dev) {
    if (d && dev >= 0 && dev < (int)d->ctx.size() && d->ctx[dev])
        d->last_error = "device " + std::to_string(dev) + ": " + gpcsd_last_error(d->ctx[dev]);
    return rc;
}

// run fn(i) for every device on its own host thread (the synchronous calls of the ABI return values, so devices only work side
// by side when their calls are issued side by side); returns the first non-zero rc in device order
template <typename F>
static int dist_parallel(gpcsd_dist *d, F fn) {
    const int n = (int)d->ctx.size();
    std::vector<int> rc(n, 0);
    if (n == 1) {
        rc[0] = fn(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i]() { rc[i] = fn(i); });
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rc[i] != 0) return dist_fail(d, rc[i], i);
    return 0;
}

extern "C" int gpcsd_dist_create(int ndev, const int *devices, gpcsd_dist **out) {
    if (!out || ndev < 1 || ndev > 64) return -3;
    *out = nullptr;
    std::unique_ptr<gpcsd_dist> d(new gpcsd_dist());
    for (int i = 0; i < ndev; ++i) {
        gpcsd_ctx *c = nullptr;
        const int rc = gpcsd_ctx_create(devices ? devices[i] : i, &c);
        if (rc != 0) {
            for (gpcsd_ctx *p : d->ctx) (void)gpcsd_ctx_destroy(p);
            return rc;
        }
        d->ctx.push_back(c);
    }
    *out = d.release();
    return 0;
}

extern "C" int gpcsd_dist_destroy(gpcsd_dist *d) {
    if (!d) return -1;
    for (gpcsd_ctx *p : d->ctx) (void)gpcsd_ctx_destroy(p);
    delete d;
    return 0;
}

extern "C" int gpcsd_dist_size(gpcsd_dist *d) { return d ? (int)d->ctx.size() : -1; }

extern "C" int gpcsd_dist_ctx(gpcsd_dist *d, int i, gpcsd_ctx **ctx) {
    if (!d || !ctx || i < 0 || i >= (int)d->ctx.size()) return -3;
    *ctx = d->ctx[i];
    return 0;
}

extern "C" const char *gpcsd_dist_last_error(gpcsd_dist *d) { return d ? d->last_error.c_str() : ""; }

extern "C" int gpcsd_dist_set_geometry_1d(gpcsd_dist *d, const double *x, int nx, const double *gl_x, const double *gl_w, int ngl) {
    if (!d) return -1;
    return dist_parallel(d, [&](int i) { return gpcsd_set_geometry_1d(d->ctx[i], x, nx, gl_x, gl_w, ngl); });
}

extern "C" int gpcsd_dist_set_geometry_2d(gpcsd_dist *d, const double *xy, int nx, const double *gl_x1, const double *gl_w1, int ngl1,
                                          const double *gl_x2, const double *gl_w2, int ngl2) {
    if (!d) return -1;
    return dist_parallel(d, [&](int i) { return gpcsd_set_geometry_2d(d->ctx[i], xy, nx, gl_x1, gl_w1, ngl1, gl_x2, gl_w2, ngl2); });
}

extern "C" int gpcsd_dist_set_time(gpcsd_dist *d, const double *t, int nt) {
    if (!d) return -1;
    return dist_parallel(d, [&](int i) { return gpcsd_set_time(d->ctx[i], t, nt); });
}

// lfp (nx, nt, ntrials) C-order.  replicate = 0: device i receives the block of trials gpcsd_shard_block assigns to rank i
// (trial sharding: loglik / predict / gradient add over devices); replicate = 1: every device receives all trials (restart
// sharding: gpcsd_dist_loglik_grad_batch deals hyper-parameter sets to devices).
extern "C" int gpcsd_dist_set_lfp(gpcsd_dist *d, const double *lfp, int nx, int nt, int ntrials, int replicate) {
    if (!d || !lfp || nx < 1 || nt < 1 || ntrials < 1) return -3;
    const int n = (int)d->ctx.size();
    // validate and partition into locals first: a rejected or failed call must not leave the handle describing data the
    // contexts do not hold (gpcsd_dist_loglik / _predict combine and scatter by first / count / ntrials)
    const bool repl = replicate != 0;
    std::vector<int> first(n, 0), count(n, ntrials);
    if (!repl) {
        if (ntrials < n) {
            d->last_error = "fewer trials than devices";
            return -3;
        }
        for (int i = 0; i < n; ++i) (void)gpcsd_shard_block(ntrials, i, n, &first[i], &count[i]);
    }
    const int rc = dist_parallel(d, [&](int i) {
        if (repl || n == 1) return gpcsd_set_lfp(d->ctx[i], lfp, nx, nt, ntrials);
        // the trial index is innermost: a block of trials is a strided slice, packed here
        const int f = first[i], cnt = count[i];
        std::vector<double> blk((size_t)nx * nt * cnt);
        for (long r = 0; r < (long)nx * nt; ++r) memcpy(&blk[(size_t)r * cnt], lfp + (size_t)r * ntrials + f, (size_t)cnt * sizeof(double));
        return gpcsd_set_lfp(d->ctx[i], blk.data(), nx, nt, cnt);
    });
    if (rc != 0) {                       // some contexts may hold the new data, some the old: the handle holds none
        d->nx = d->nt = d->ntrials = 0;
        d->first.assign(n, 0);
        d->count.assign(n, 0);
        return rc;
    }
    d->nx = nx; d->nt = nt; d->ntrials = ntrials; d->replicated = repl;
    d->first = first;
    d->count = count;
    return 0;
}

// GPCSD1D/2D.loglik over all trials (gpcsd1d.py:113-128): every device queues its partial evaluation, then the partial quadratic
// terms are collected and summed in device order
extern "C" int gpcsd_dist_loglik(gpcsd_dist *d, const gpcsd_hparams *hp, double *out) {
    if (!d || !hp || !out) return -3;
    if (d->ntrials <= 0) {
        d->last_error = "no data on the devices (gpcsd_dist_set_lfp has not succeeded)";
        return -4;
    }
    if (d->replicated) return dist_fail(d, gpcsd_loglik(d->ctx[0], hp, out), 0);
    const int n = (int)d->ctx.size();
    for (int i = 0; i < n; ++i) {                         // asynchronous: the devices work side by side without host threads
        const int rc = gpcsd_loglik_parts_async(d->ctx[i], hp);
        if (rc != 0) {
            for (int j = 0; j < i; ++j) {
                double tmp[2];
                (void)gpcsd_loglik_parts_wait(d->ctx[j], tmp);
            }
            return dist_fail(d, rc, i);
        }
    }
    double sumlog = 0.0, quad = 0.0;
    int rc_first = 0, bad = -1;
    for (int i = 0; i < n; ++i) {
        double p[2] = {0.0, 0.0};
        const int rc = gpcsd_loglik_parts_wait(d->ctx[i], p);
        if (rc != 0 && rc_first == 0) {
            rc_first = rc;
            bad = i;
        }
        if (i == 0) sumlog = p[0];                        // identical on every device (deterministic replicas)
        quad += p[1];
    }
    if (rc_first != 0) return dist_fail(d, rc_first, bad);
    return gpcsd_combine_loglik(d->ntrials, sumlog, quad, out);
}

// Local pieces and gradient over all trials: out2[0] = sum log D, out2[1] = sum of the partial quadratic terms, grad = the sum
// of the devices' gradients of L_loc (see gpcsd_loglik_grad: both are sums over trials plus a term linear in the trial count)
extern "C" int gpcsd_dist_loglik_grad(gpcsd_dist *d, const gpcsd_hparams *hp, double *out2, double *grad, int ngrad) {
    if (!d || !hp || !out2 || !grad || ngrad < 1) return -3;
    if (d->ntrials <= 0) {
        d->last_error = "no data on the devices (gpcsd_dist_set_lfp has not succeeded)";
        return -4;
    }
    if (d->replicated) return dist_fail(d, gpcsd_loglik_grad(d->ctx[0], hp, out2, grad, ngrad), 0);
    const int n = (int)d->ctx.size();
    std::vector<double> o2((size_t)2 * n), g((size_t)ngrad * n);
    const int rc = dist_parallel(d, [&](int i) { return gpcsd_loglik_grad(d->ctx[i], hp, &o2[2 * i], &g[(size_t)ngrad * i], ngrad); });
    if (rc != 0) return rc;
    out2[0] = o2[0];
    out2[1] = 0.0;
    for (int k = 0; k < ngrad; ++k) grad[k] = 0.0;
    for (int i = 0; i < n; ++i) {
        out2[1] += o2[2 * i + 1];
        for (int k = 0; k < ngrad; ++k) grad[k] += g[(size_t)ngrad * i + k];
    }
    return 0;
}

// The restarts of fit (gpcsd1d.py:193-220): nsets hyper-parameter sets dealt to the devices (set k -> device k mod ndev, every
// device holds all trials: gpcsd_dist_set_lfp(..., replicate = 1)), each device evaluating its sets in one chain of launches.
// out2 (nsets, 2), grad (nsets, ngrad), status (nsets) as gpcsd_loglik_grad_batch.
extern "C" int gpcsd_dist_loglik_grad_batch(gpcsd_dist *d, const gpcsd_hparams *hps, int nsets, double *out2, double *grad, int ngrad,
                                            int *status) {
    if (!d || !hps || nsets < 1 || !out2 || !grad || !status || ngrad < 1) return -3;
    if (d->ntrials <= 0) {
        d->last_error = "no data on the devices (gpcsd_dist_set_lfp has not succeeded)";
        return -4;
    }
    if (!d->replicated) {
        d->last_error = "gpcsd_dist_loglik_grad_batch needs every device to hold all trials (gpcsd_dist_set_lfp with replicate = 1)";
        return -3;
    }
    const int n = (int)d->ctx.size();
    return dist_parallel(d, [&](int i) {
        std::vector<gpcsd_hparams> mine;
        std::vector<int> idx;
        for (int k = i; k < nsets; k += n) {
            mine.push_back(hps[k]);
            idx.push_back(k);
        }
        if (mine.empty()) return 0;
        const int m = (int)mine.size();
        std::vector<double> o2((size_t)2 * m), g((size_t)ngrad * m);
        std::vector<int> st(m, 0);
        const int rc = gpcsd_loglik_grad_batch(d->ctx[i], mine.data(), m, o2.data(), g.data(), ngrad, st.data());
        if (rc != 0) return rc;
        for (int j = 0; j < m; ++j) {
            out2[2 * idx[j]] = o2[2 * j];
            out2[2 * idx[j] + 1] = o2[2 * j + 1];
            memcpy(grad + (size_t)ngrad * idx[j], &g[(size_t)ngrad * j], (size_t)ngrad * sizeof(double));
            status[idx[j]] = st[j];
        }
        return 0;
    });
}

// GPCSD{1,2}D.predict over all trials (gpcsd1d.py:248-293): every device predicts its block; the outputs -- (n_temporal, nz,
// ntstar, ntrials) lists and (nz, ntstar, ntrials) sums, trial index innermost -- are assembled from the blocks.  Any may be NULL.
extern "C" int gpcsd_dist_predict(gpcsd_dist *d, const gpcsd_hparams *hp, const double *z, int nz, const double *tstar, int ntstar,
                                  int type, double *csd_list, double *csd, double *lfp_list, double *lfp) {
    if (!d || !hp || !z || !tstar || nz < 1 || ntstar < 1) return -3;
    if (d->ntrials <= 0) {
        d->last_error = "no data on the devices (gpcsd_dist_set_lfp has not succeeded)";
        return -4;
    }
    const int n = (int)d->ctx.size();
    if (d->replicated || n == 1)
        return dist_fail(d, gpcsd_predict(d->ctx[0], hp, z, nz, tstar, ntstar, type, csd_list, csd, lfp_list, lfp), 0);
    const int C = hp->n_temporal, R = d->ntrials;
    return dist_parallel(d, [&](int i) {
        const int f = d->first[i], cnt = d->count[i];
        const size_t rows = (size_t)nz * ntstar;
        std::vector<double> b_cl, b_c, b_ll, b_l;
        if (csd_list) b_cl.resize(rows * C * cnt);
        if (csd) b_c.resize(rows * cnt);
        if (lfp_list) b_ll.resize(rows * C * cnt);
        if (lfp) b_l.resize(rows * cnt);
        const int rc = gpcsd_predict(d->ctx[i], hp, z, nz, tstar, ntstar, type, csd_list ? b_cl.data() : nullptr,
                                     csd ? b_c.data() : nullptr, lfp_list ? b_ll.data() : nullptr, lfp ? b_l.data() : nullptr);
        if (rc != 0) return rc;
        auto scatter = [&](const std::vector<double> &blk, double *dst, size_t nrows) {       // rows of cnt trials -> rows of R trials
            for (size_t r = 0; r < nrows; ++r) memcpy(dst + r * R + f, &blk[r * cnt], (size_t)cnt * sizeof(double));
        };
        if ((type & 1) && csd_list) scatter(b_cl, csd_list, rows * C);
        if ((type & 1) && csd) scatter(b_c, csd, rows);
        if ((type & 2) && lfp_list) scatter(b_ll, lfp_list, rows * C);
        if ((type & 2) && lfp) scatter(b_l, lfp, rows);
        return 0;
    });
}
