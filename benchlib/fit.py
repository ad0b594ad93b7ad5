"""bench.py's fit benchmark (objective + analytic gradient evaluations in lock-step batches: cfg3fit, cfg5, npx69fit, aud24) and the dense
Cholesky on its own."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .profiles import pmc_step_traffic
from .workloads import FP64_MFMA_SPEC_TFLOPS, SETTLE_S, build_model, oracle_setup, synth_data


def potrf_bench(n=12000, reps=3):
    """The dense Cholesky path (north_star: "(Ks (x) Kt + sig2 I) Cholesky factor, log-det and triangular solves") at the one size of
    BASELINE's configurations where the dense matrix fits one GPU -- cfg2's N = 24 x 500 = 12 000 (1.15 GB): the blocked factorisation
    on a device-resident SPD matrix (gpcsd_potrf_bench: HIP events on the library's stream around the factorisation alone), n^3 / 3
    flops against the fp64 MFMA peak, with the event-scope split of one profiled factorisation."""
    from gpcsd_amd import _hip
    ctx = _hip.default_context()
    _quiesce_host()
    # median of single factorisations: each takes ~600 launches that the host has to keep ahead of; one host stall (a BLAS worker
    # pool of the CPU baseline still spinning: a driver-style run once read 28.5 ms where the event scopes of the same process
    # said 20.7) would otherwise sit in the mean
    g0 = ctx.potrf_gate_timeouts()
    runs = sorted(ctx.potrf_bench(n, reps=1) for _ in range(max(3, reps)))
    ms, tf = runs[len(runs) // 2]
    gate_timeouts = ctx.potrf_gate_timeouts() - g0         # (gates that gave up waiting: they steer the order of execution only)
    ctx.prof_reset()
    ctx.prof_enable(1)
    ctx.potrf_bench(n, reps=1)
    ctx.prof_enable(0)
    prof = {k: v for k, v in ctx.prof_all().items() if k.startswith("potrf") and v["count"]}
    # (the profiled call factors twice: one untimed repetition + one)
    split = {k: {"ms": v["ms"] / 2.0, "launches": v["count"] // 2,
                 "tflops": ((v["flops"] / 2.0) / (v["ms"] / 2.0 * 1e-3) / 1e12) if (v["ms"] and v["flops"]) else None} for k, v in prof.items()}
    tu = split.get("potrf_syrk", {})
    out = {"metric": "gpcsd_dense_cholesky_factorisations_per_sec", "value": 1e3 / ms, "unit": "factorisations/s", "n": n, "ms": ms,
           "dtype": "f64", "flops": n ** 3 / 3.0, "tflops": tf, "frac_of_fp64_mfma_peak": tf / FP64_MFMA_SPEC_TFLOPS,
           "trailing_update": {"kernel": "gemm_f64_kernel<EPI_SUB, lower> [potrf_syrk]: rank-256 update A22 -= L21 L21^T, tiles on or "
                                         "below the diagonal", "ms": tu.get("ms"), "launches": tu.get("launches"),
                               "tflops": tu.get("tflops"), "frac": (tu.get("tflops") or 0.0) / FP64_MFMA_SPEC_TFLOPS},
           "scopes": split, "diag128_phases_us": ctx.potrf_diag_probe(), "gate_timeouts": gate_timeouts,
           "config": {"workload": "blocked Cholesky of a %d x %d SPD matrix resident in HBM (N of BASELINE cfg2: 24 x 500)" % (n, n)}}
    out["headline"] = {"ms": ms, "frac": out["frac_of_fp64_mfma_peak"], "trailing_update_frac": out["trailing_update"]["frac"]}
    return out


def _quiesce_host(seconds=0.25):
    """The CPU legs of this script (oracle parity checks, the CPU baseline's thread sweep) leave OpenBLAS workers spinning for
    tens of milliseconds after their last call, on the very CPUs the launch thread is bound to; a GPU measurement that starts
    in that window sees 20-30 ms host stalls (cfg2's sub-result once read 1.09 ms per step for 0.86).  Wait them out."""
    time.sleep(seconds)


def run_fit_bench(args, w, rank, world, local_rank, backend, compact=False, cpu_legs=None):
    """BASELINE cfg5: GPCSD1D hyper-parameter fit, 24 x 500 x 200 trials resident on every GPU, 32 restarts sharded over the
    GPUs.  The unit of work is one objective + analytic-gradient evaluation of one restart (what L-BFGS-B asks for); a step
    evaluates one lock-step batch of B restarts in one chain of launches (gpcsd_loglik_grad_batch).  Reported: evaluations/s
    through the batched call, the same through one-at-a-time calls (the round-1 path), and a truncated real fit()
    (SciPy L-BFGS-B chains in lock-step) with restarts/s."""
    import torch
    from gpcsd_amd.dist import TrialSharding
    total_restarts = int(w.get("restarts", 32))
    mine = [k for k in range(total_restarts) if k % world == rank]
    B = args.fit_batch or min(32, len(mine))                        # all of this rank's restarts advance in one lock-step batch
    m = build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    m.set_device(local_rank)
    lfp = synth_data(w, m, w["trials_per_gpu"], seed=1000)          # every rank holds the same trials
    data_sigma2 = [float(tc.params["sigma2"]["value"]) for tc in m.temporal_cov_list]   # (2D: rescaled by 1 / mean diag Ks)
    m.update_lfp(lfp, w["t"])
    if os.environ.get("GPCSD_GRAM_PRECISION") == "32":              # BASELINE cfg5 names "fp32 kernel build + fp64 factor"
        m.gram_precision = 32
    ctx = m._sync_device()
    sharding = TrialSharding() if (world > 1 or os.environ.get("GPCSD_BENCH_FORCE_DIST") == "1") else None
    if sharding is not None:
        m.shard_restarts(sharding)
    # restart k starts from the k-th draw of the default priors (np.random.seed(k), sampled on the host)
    starts = []
    if "starts_around_truth" in w:
        tp_true = m._current_tparams()
        lo, hi = (np.array([b[i] for b in m._bounds()], dtype=float) for i in (0, 1))
        for k in range(total_restarts):
            s0 = tp_true + w["starts_around_truth"] * np.random.RandomState(k).standard_normal(tp_true.size)
            starts.append(np.minimum(np.maximum(s0, lo + 1e-6), hi - 1e-6))
    else:
        for k in range(total_restarts):
            np.random.seed(k)
            starts.append(m._sample_start(False))
    ng = 1 + m.dim + 2 * len(m.temporal_cov_list) + int(np.size(m.sig2n["value"]))

    def hp_of(tp):
        m._set_from_tparams(tp, False)
        return m._hparams(m.JITTER)
    sets = [hp_of(starts[k]) for k in mine[:B]]
    hps = [h for h, _ in sets]

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if sharding is not None:
            import torch.distributed as td
            td.barrier()
    # G lock-step groups side by side, each with its own context (what fit(batch=B, workers=G) does): a step is one batched
    # evaluation of EVERY group, i.e. G * B objective+gradient evaluations
    G = max(1, args.fit_groups)
    ctxs, group_hps = [ctx], [hps]
    for gi in range(1, G):
        mg = m._clone_for_worker()
        mg.set_device(local_rank)
        cg = mg._sync_device()
        ks = mine[gi * B:(gi + 1) * B] or mine[:B]
        gh = []
        for k in ks:
            mg._set_from_tparams(starts[k], False)
            gh.append(mg._hparams(mg.JITTER))
        ctxs.append(cg)
        group_hps.append([h for h, _ in gh])
        sets.extend(gh)                                          # keep the sig2n arrays alive

    def run_groups(nsteps):
        if G == 1:
            for _ in range(nsteps):
                ctx.loglik_grad_batch(hps, ng)
            return
        import threading
        ths = [threading.Thread(target=lambda c=c, h=h: [c.loglik_grad_batch(h, ng) for _ in range(nsteps)])
               for c, h in zip(ctxs, group_hps)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    t_first_eval = time.perf_counter()
    run_groups(max(3, min(args.setup_steps, 20)))
    while world == 1 and time.perf_counter() - t_first_eval < SETTLE_S:      # (see run_step_bench: a model's first tenth of a second)
        run_groups(1)
    run_groups(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_groups(args.steps)
    for cg in ctxs:
        cg.synchronize()
    fence()
    elapsed = time.perf_counter() - t0
    if args.only_value:                    # the command the rocprofv3 passes under profiles/ are taken over: nothing after the loop
        if rank != 0:
            return None
        return {"metric": "gpcsd_fit_loglik_grad_evals_per_sec", "value": G * B * world * args.steps / elapsed, "unit": "evals/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": w["label"], "n_elec": w["nx"], "n_t": w["nt"], "trials_per_gpu": w["trials_per_gpu"],
                           "lockstep_batch": B}, "only_value": "setup + warm-up + timed loop only (the command profiled under profiles/)"}
    # one group alone, for reference
    fence()
    t0g = time.perf_counter()
    for _ in range(args.steps):
        sumlog, quad, grad, st = ctx.loglik_grad_batch(hps, ng)
    ctx.synchronize()
    one_group_s = (time.perf_counter() - t0g) / args.steps
    # one at a time (what a chain on its own costs; fit(workers=1) of round 1)
    nseq = max(8, min(args.steps, 40))
    for _ in range(3):
        ctx.loglik_grad(hps[0], ng)
    ctx.synchronize()
    t1 = time.perf_counter()
    for i in range(nseq):
        ctx.loglik_grad(hps[i % len(hps)], ng)
    ctx.synchronize()
    seq_s = (time.perf_counter() - t1) / nseq
    # the same evaluations in lock-step batches of 1 / 4 / 8 sets (what a rank of an N-GPU fit holds), and the fenced
    # log-likelihood alone beside them: the gradient's price over the value's
    by_batch, ll_fenced_ms = None, None
    if w.get("starts_around_truth") and world == 1:
        by_batch = {}
        for bb in (1, 2, 4, 8):
            if bb > len(hps):
                break
            for _ in range(3):
                ctx.loglik_grad_batch(hps[:bb], ng)
            ctx.synchronize()
            tb = time.perf_counter()
            nb_ = max(10, args.steps // 2)
            for _ in range(nb_):
                ctx.loglik_grad_batch(hps[:bb], ng)
            ctx.synchronize()
            dtb = (time.perf_counter() - tb) / nb_
            by_batch[str(bb)] = {"ms_per_batched_call": 1e3 * dtb, "evals_per_sec": bb / dtb}
        ctx.decomposition_cache(False)                  # (every call decomposes both sides, as an optimiser's evaluations do)
        for _ in range(3):
            ctx.loglik_parts(hps[0])
        ctx.synchronize()
        tl = time.perf_counter()
        for _ in range(20):
            ctx.loglik_parts(hps[0])
        ll_fenced_ms = 1e3 * (time.perf_counter() - tl) / 20
        ctx.decomposition_cache(True)
    if sharding is not None:
        import torch.distributed as td
        tt = torch.tensor([elapsed, seq_s], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, seq_s = (float(v) for v in tt.cpu())
    # profiled pass of the batched step
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(3):
        ctx.loglik_grad_batch(hps, ng)
    ctx.prof_enable(False)
    prof = ctx.prof_all()
    # a truncated real fit: lock-step SciPy chains, all of this rank's restarts
    opts = {"maxiter": args.fit_maxiter, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}
    def timed_fit(driver):
        m.fit_driver = driver
        m.fit(n_restarts=total_restarts, options=opts, starts=starts, batch=B, workers=G)          # warm (allocations, graphs)
        tf = time.perf_counter()
        m.fit(n_restarts=total_restarts, options=opts, starts=starts, batch=B, workers=G)
        dt = time.perf_counter() - tf
        nb_, npts_ = getattr(m, "fit_batches_", (0, 0))
        return {"driver": getattr(m, "fit_driver_used_", driver), "restarts": total_restarts, "maxiter": args.fit_maxiter, "seconds": dt,
                "restarts_per_sec": total_restarts / dt, "evals": int(npts_) * world, "batched_calls": int(nb_),
                "evals_per_sec": npts_ * world / dt, "best_nll": float(np.min(m.fit_nll_values_)),
                "nll_values": [float(v) for v in np.asarray(m.fit_nll_values_)]}
    # the round-2 driver (unmodified minimize() calls on threads that rendezvous per evaluation) beside the default one
    # (one driver stepping SciPy's L-BFGS-B states through its reverse-communication interface): same optima, bit for bit
    fit_threads = timed_fit("threads")
    fit_main = timed_fit("auto")
    fit_s = fit_main["seconds"]
    nb, npts = fit_main["batched_calls"], fit_main["evals"] // max(world, 1)
    if rank != 0:
        return None
    n_eval = G * B * world * args.steps
    # parity spot check beside the numbers: the HIP objective and analytic gradient at restart 0's start against the oracle
    # objective and its central differences (checker code; 2 p + 1 oracle evaluations)
    cpu_leg = None
    if world == 1:
        kinds = [k for k, _, _ in w["temporal"]]
        snames = ("ell",) if w["dim"] == 1 else ("ell1", "ell2")
        n_sig = int(np.size(m.sig2n["value"]))
        # at the hyper-parameters the data were drawn from (a well-scaled point: central differences of a prior-drawn start,
        # where the objective is ~1e7 and dominated by one term, only measure the differences' own rounding)
        m.R["value"] = w["R"]
        m.sig2n["value"] = w["sig2n"] if n_sig == 1 else np.array(w["sig2n_list"], dtype=float)
        for nm, v in zip(snames, w["ell_s"]):
            m.spatial_cov.params[nm]["value"] = v
        for tc, (_, ell, _s2), s2 in zip(m.temporal_cov_list, w["temporal"], data_sigma2):
            tc.params["ell"]["value"], tc.params["sigma2"]["value"] = ell, s2
        tp0 = m._current_tparams()
        f_gpu, g_gpu = m._objective_and_grad(tp0, False)          # the GPU half now; the oracle half is a CPU leg

        def cpu_leg():
            O, geom, hpo, _hpo0 = oracle_setup(w, m)

            def cpu_obj(tp):
                hh = O.hparams_from_tparams(tp, w["dim"], kinds, n_sig, eps=w["eps"], jitter=m.JITTER)
                lp = m.R["prior"].lpdf(hh["R"])
                if n_sig == 1:
                    lp += m.sig2n["prior"].lpdf(hh["sig2n"])
                else:
                    lp += sum(pr.lpdf(v) for pr, v in zip(m.sig2n["prior"], np.atleast_1d(hh["sig2n"])))
                for nm, v in zip(snames, hh["ell_s"]):
                    lp += m.spatial_cov.params[nm]["prior"].lpdf(v)
                for tc, (_, ell, s2) in zip(m.temporal_cov_list, hh["temporal"]):
                    lp += tc.params["ell"]["prior"].lpdf(ell) + tc.params["sigma2"]["prior"].lpdf(s2)
                return -(O.loglik(geom, hh, lfp) + lp)
            f_cpu = cpu_obj(tp0)
            g_cpu = np.zeros_like(tp0)
            for i in range(tp0.size):
                e = np.zeros_like(tp0)
                e[i] = 1e-5
                g_cpu[i] = (cpu_obj(tp0 + e) - cpu_obj(tp0 - e)) / 2e-5
            res["parity"] = {"objective_rel_err_vs_oracle": abs(f_gpu - f_cpu) / abs(f_cpu),
                             "gradient_max_err_over_max_component_vs_oracle_fd":
                                 float(np.max(np.abs(g_gpu - g_cpu)) / np.max(np.abs(g_cpu)))}
            # ... and against the oracle's closed-form gradient (O.loglik_and_grad: pinned by central differences in the CPU
            # suite), timed as the CPU baseline of this workload: one objective + gradient evaluation on the host cores
            from threadpoolctl import threadpool_limits
            hh0 = O.hparams_from_tparams(tp0, w["dim"], kinds, n_sig, eps=w["eps"], jitter=m.JITTER)
            dlp = np.zeros_like(tp0)                                 # d log-prior / d tp (priors.py is host arithmetic on both sides)
            slots = [m.R] + [m.spatial_cov.params[nm] for nm in snames]
            for tc in m.temporal_cov_list:
                slots += [tc.params["ell"], tc.params["sigma2"]]
            nat = [hh0["R"]] + list(hh0["ell_s"]) + [v for (_, ell, s2) in hh0["temporal"] for v in (ell, s2)]
            for i, (sl, v) in enumerate(zip(slots, nat)):
                dlp[i] = sl["prior"].dlpdf(v) * v
            sv = np.atleast_1d(hh0["sig2n"])
            prs = [m.sig2n["prior"]] if n_sig == 1 else list(m.sig2n["prior"])
            for j, (pr, v) in enumerate(zip(prs, sv)):
                dlp[len(nat) + j] = pr.dlpdf(v) * v
            nthreads = min(16, os.cpu_count() or 1)
            with threadpool_limits(limits=nthreads):
                O.loglik_and_grad(geom, lfp, tp0, kinds, n_sig, eps=w["eps"], jitter=m.JITTER)          # warm
                ts = []
                while len(ts) < 3 or (sum(ts) < 10.0 and len(ts) < 20):
                    tc0 = time.perf_counter()
                    ll_cf, g_cf = O.loglik_and_grad(geom, lfp, tp0, kinds, n_sig, eps=w["eps"], jitter=m.JITTER)
                    ts.append(time.perf_counter() - tc0)
            g_cf = -(g_cf + dlp)
            res["parity"]["gradient_worst_component_rel_err_vs_oracle_closed_form"] = float(
                np.max(np.abs(g_gpu - g_cf) / np.maximum(np.abs(g_cf), 1e-9 * np.max(np.abs(g_cf)))))
            res["cpu_baseline"] = {"value": 1.0 / float(np.median(ts)), "unit": "evals/s", "cores": nthreads, "kind": "port",
                                   "blas_threads": nthreads, "host_cpus": os.cpu_count(),
                                   "sample": "oracle objective + closed-form gradient (O.loglik_and_grad) on the bench's own %d "
                                             "trials, %d repetitions, median, %d BLAS threads" % (lfp.shape[2], len(ts), nthreads)}
    # the script's next step (fit_gpcsd_baseline.py:103-105): predict at the electrodes -- and at 100 depths -- with the fitted model;
    # here at the hyper-parameters the data were drawn from, results left in HBM, every call fenced (rank-local)
    pred = None
    if "z100" in w and world == 1:
        from gpcsd_amd import _hip as _h
        pred = {}
        hp0, _k0 = m._hparams(0.0)
        for key, zz in (("predict_trials_per_sec", w["x"]), ("predict100_trials_per_sec", w["z100"])):
            for _ in range(3):
                ctx.predict_resident(hp0, zz, w["t"], _h.PRED_CSD, want_lists=True)
            ctx.synchronize()
            tpz = time.perf_counter()
            for _ in range(20):
                ctx.predict_resident(hp0, zz, w["t"], _h.PRED_CSD, want_lists=True)
                ctx.synchronize()
            pred[key] = w["trials_per_gpu"] * 20 / (time.perf_counter() - tpz)
    gemm_flops = sum(v["flops"] for k, v in prof.items() if k.startswith("gemm_")) / 3.0
    tail = prof.get("sytrd_rtail")
    eig_flops = 4.0 * tail["flops"] / 3.0 if tail else 0.0           # tridiagonalisation + 3x for the back-transformation
    step_s = elapsed / args.steps
    res = {
        "metric": "gpcsd_fit_loglik_grad_evals_per_sec",
        "value": n_eval / elapsed, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_steps": max(3, min(args.setup_steps, 20)),
        "ms_per_step": 1e3 * step_s, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if m.gram_precision == 64 else "f32 Gram build + f64", "data": "synthetic",
        "config": {"workload": w["label"], "n_elec": w["nx"], "n_t": w["nt"], "trials_per_gpu": w["trials_per_gpu"],
                   "restarts_total": total_restarts, "restarts_per_gpu": len(mine), "lockstep_batch": B, "lockstep_groups": G,
                   "parallelism": "restart-sharded x%d, %d lock-step groups of %d restarts per GPU" % (world, G, B)},
        "evals_per_sec_one_at_a_time_per_gpu": 1.0 / seq_s,
        "evals_per_sec_one_group_per_gpu": B / one_group_s,
        "batched_over_sequential": (B / one_group_s) / (1.0 / seq_s),
        "parity": None,
        "all_groups_over_sequential": (G * B / step_s) / (1.0 / seq_s),
        "fit": dict(fit_main, evals_per_sec_through_scipy=fit_main["evals_per_sec"],
                    real_fit_over_synthetic_evals_per_sec=fit_main["evals_per_sec"] / (n_eval / elapsed),
                    same_optima_as_threads_driver=fit_main["nll_values"] == fit_threads["nll_values"]),
        "fit_threads_driver": {k: v for k, v in fit_threads.items() if k != "nll_values"},
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_SPEC_TFLOPS,
                     "achieved": G * (gemm_flops + eig_flops) / step_s / 1e12,
                     "frac": G * (gemm_flops + eig_flops) / step_s / 1e12 / FP64_MFMA_SPEC_TFLOPS,
                     "scope": "one step = %d lock-step group(s) x %d objective+gradient evaluations, each group one chain of launches; "
                              "flops actually launched (profiled on one group)" % (G, B),
                     "executed_gflop_per_step": G * (gemm_flops + eig_flops) / 1e9, "traffic": None,
                     "dominant_kernel": None if not tail else {
                         "kernel": "sytrd_rtail_kernel", "avg_launch_ms": tail["ms"] / tail["count"],
                         "launches_per_step": tail["count"] / 3.0, "workgroups_per_launch": "%d (one per half problem and set)" % (2 * B),
                         "share_of_step_wall": tail["ms"] / 3.0 / (1e3 * step_s)},
                     "per_kernel_ms_per_step": {k: v["ms"] / 3.0 for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:16]}},
    }
    res["config"]["fit_evals_per_sec"], res["config"]["fit_restarts_per_sec"] = fit_main["evals_per_sec"], fit_main["restarts_per_sec"]
    if by_batch is not None:
        res["evals_by_lockstep_batch"] = by_batch
        res["single_eval_ms"] = by_batch["1"]["ms_per_batched_call"]
        res["fenced_loglik_ms"] = ll_fenced_ms
        res["single_eval_over_fenced_loglik"] = by_batch["1"]["ms_per_batched_call"] / ll_fenced_ms
        res["config"].update(single_eval_ms=res["single_eval_ms"], fenced_loglik_ms=ll_fenced_ms,
                             single_eval_over_fenced_loglik=res["single_eval_over_fenced_loglik"],
                             batch4_evals_per_sec=by_batch.get("4", {}).get("evals_per_sec"))
    # HBM traffic per batched step from the committed rocprofv3 --pmc passes over `bench.py --workload <this> --only-value`
    traffic, traffic_src = pmc_step_traffic(args.workload) if (world == 1 and args.fit_batch is None) else (None, None)
    res["roofline"]["traffic"] = traffic
    if traffic_src:
        res["roofline"]["traffic_source"] = traffic_src
        # algorithmic bytes of one evaluation: the trials read once; a batch reads them once per set
        res["roofline"]["algorithmic_bytes_per_step"] = 8.0 * w["nx"] * w["nt"] * w["trials_per_gpu"] * B
        res["roofline"]["traffic_over_algorithmic"] = traffic / res["roofline"]["algorithmic_bytes_per_step"]
    if pred:
        res.update(pred)
        res["config"].update(pred)
    if cpu_leg is not None:
        if cpu_legs is None:
            cpu_leg()
        else:
            cpu_legs.append(cpu_leg)
    return res
