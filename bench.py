#!/usr/bin/env python3
"""bench.py -- GPCSD log-marginal-likelihood + posterior-predict throughput on MI355X.

Workload (BASELINE.json configs[2]/[3]): GPCSD2D, 384-channel Neuropixels checkerboard x 500 time
points, 50 synthetic trials PER GPU (cfg3 at N=1; cfg4 = 400 trials at N=8: weak scaling), float64, ngl 20x60,
SE + Matern-1/2 temporal kernels.  One step = one loglik() evaluation over the resident trials + one
predict(z = electrodes, t, type="csd") of every resident trial, inputs resident in HBM, outputs left in HBM
(no PCIe in the timed region; the PCIe-inclusive rate is reported separately as `pcie_inclusive_trials_per_sec`).

    python bench.py --gpus 1 --steps 100 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the fp64 MFMA GEMM) from HIP-event timings taken
on the library's own stream in a separate profiled pass; `cpu_baseline` times the NumPy oracle (a port, not the
reference, which cannot travel to the GPU box) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

FP64_MFMA_SPEC_TFLOPS = 78.6      # AMD public MI355X fp64 matrix spec (v_mfma_f64_16x16x4_f64); not in the local guides
HBM_PEAK_GBS = 8000.0


def neuropixels_xy(nchan):
    c = np.arange(nchan)
    return np.stack([np.array([16.0, 48.0, 0.0, 32.0])[c % 4], np.floor(c / 2) * 20.0], axis=1)


def workload(name):
    if name == "cfg3":
        return dict(dim=2, nx=384, nt=500, x=neuropixels_xy(384), t=0.4 * np.arange(500.0)[:, None], ngl1=20, ngl2=60,
                    R=100.0, eps=80.0, ell_s=(40.0, 150.0), temporal=[(0, 20.0, 0.5), (1, 5.0, 0.7)], sig2n=0.05,
                    trials_per_gpu=50, label="GPCSD2D 384ch x 500t x 50 trials/GPU, fp64, ngl 20x60 (BASELINE cfg3/cfg4)")
    if name == "cfg2":
        return dict(dim=1, nx=24, nt=500, x=np.linspace(0, 2300, 24)[:, None], t=np.arange(500.0)[:, None], ngl=100,
                    R=100.0, eps=0.0, ell_s=(200.0,), temporal=[(0, 20.0, 0.5), (1, 5.0, 0.7)], sig2n=0.05,
                    trials_per_gpu=200, label="GPCSD1D 24 x 500t x 200 trials/GPU, fp64, ngl 100 (BASELINE cfg2)")
    raise SystemExit("unknown workload %r" % name)


def build_model(w, lfp):
    from gpcsd_amd.gpcsd1d import GPCSD1D
    from gpcsd_amd.gpcsd2d import GPCSD2D
    from gpcsd_amd.covariances import GPCSDTemporalCovSE, GPCSDTemporalCovMatern
    np.random.seed(0)
    tcl = []
    for kind, ell, s2 in w["temporal"]:
        tc = GPCSDTemporalCovSE(w["t"]) if kind == 0 else GPCSDTemporalCovMatern(w["t"])
        tc.params["ell"]["value"], tc.params["sigma2"]["value"] = ell, s2
        tcl.append(tc)
    if w["dim"] == 1:
        m = GPCSD1D(lfp, w["x"], w["t"], a=0.0, b=2300.0, ngl=w["ngl"], temporal_cov_list=tcl)
        m.spatial_cov.params["ell"]["value"] = w["ell_s"][0]
    else:
        m = GPCSD2D(lfp, w["x"], w["t"], ngl1=w["ngl1"], ngl2=w["ngl2"], temporal_cov_list=tcl, eps=w["eps"])
        m.spatial_cov.params["ell1"]["value"], m.spatial_cov.params["ell2"]["value"] = w["ell_s"]
    m.R["value"] = w["R"]
    m.sig2n["value"] = w["sig2n"]
    return m


def synth_data(w, m, ntrials, seed):
    """Draw trials from the model itself: Y = Qs sqrt(es+) Z (Qt sqrt(et+))^T + sqrt(sig2n) E.
    In 2D the temporal variances are first rescaled by 1/mean(diag Ks) so Ks (x) Kt is O(1)."""
    from gpcsd_amd import _hip
    ctx = _hip.default_context()
    if w["dim"] == 2:
        Ks = m.spatial_cov.compKphi_2d(w["R"], w["eps"])
        md = float(np.mean(np.diag(Ks)))
        for tc, (_, _, s2) in zip(m.temporal_cov_list, w["temporal"]):
            tc.params["sigma2"]["value"] = s2 / md
    else:
        Ks = m.spatial_cov.compKphi_1d(w["R"])
    Kt = sum(tc.compute_Kt() for tc in m.temporal_cov_list)
    es, Qs = ctx.eigh(Ks)
    et, Qt = ctx.eigh(Kt)
    # eigenvector signs are solver-dependent: fix them (largest |component| positive) so the synthetic data set does
    # not change when the eigensolver does
    for Q in (Qs, Qt):
        Q *= np.sign(Q[np.argmax(np.abs(Q), axis=0), np.arange(Q.shape[1])])[None, :]
    Ls = Qs * np.sqrt(np.maximum(es, 0.0))[None, :]
    Lt = Qt * np.sqrt(np.maximum(et, 0.0))[None, :]
    rs = np.random.RandomState(seed)
    Z = rs.standard_normal((ntrials, w["nx"], w["nt"]))
    E = rs.standard_normal((ntrials, w["nx"], w["nt"]))
    Y = np.matmul(np.matmul(Ls, Z), Lt.T) + np.sqrt(w["sig2n"]) * E
    return np.ascontiguousarray(np.moveaxis(Y, 0, 2))           # (nx, nt, R) like the reference


def algorithmic_flops(w, R, nz, C):
    nx, nt = w["nx"], w["nt"]
    G = w["ngl"] if w["dim"] == 1 else w["ngl1"] * w["ngl2"]
    f_spatial = 2.0 * nx * G * G + 2.0 * nx * nx * G
    f_eig = 9.0 * (nx ** 3 + nt ** 3)
    f_proj = 2.0 * nx * nx * nt + 2.0 * nx * nt * nt
    loglik = f_spatial + f_eig + R * f_proj
    pred_trial = 2.0 * f_proj + 2.0 * nz * nx * nt + C * 2.0 * nz * nt * nt
    predict = f_spatial + f_eig + 2.0 * nx * G * nz + R * pred_trial
    return loglik, predict, pred_trial


def cpu_baseline(w, m, lfp_sample):
    """Oracle (NumPy/LAPACK port) timed on the host cores, bounded sample; checker code, never the product."""
    from oracle import gpcsd_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    temporal = [(tc.kind, tc.params["ell"]["value"], tc.params["sigma2"]["value"]) for tc in m.temporal_cov_list]
    if w["dim"] == 1:
        geom = O.Geometry1D(w["x"], w["t"], a=0.0, b=2300.0, ngl=w["ngl"])
        jit = 1e-8
    else:
        geom = O.Geometry2D(w["x"], w["t"], ngl1=w["ngl1"], ngl2=w["ngl2"])
        jit = 1e-7
    hp = O.make_hparams(w["R"], w["ell_s"], temporal, w["sig2n"], eps=w["eps"], jitter=jit)
    hp0 = dict(hp)
    hp0["jitter"] = 0.0
    Rs = lfp_sample.shape[2]
    O.loglik(geom, hp, lfp_sample[:, :, :1])                     # warm BLAS
    reps, t_ll, t_pr = 0, 0.0, 0.0
    t_start = time.perf_counter()
    ll = None
    while reps < 3 or (time.perf_counter() - t_start < 10.0 and reps < 20):
        t0 = time.perf_counter()
        ll = O.loglik(geom, hp, lfp_sample)
        t1 = time.perf_counter()
        O.predict(geom, hp0, lfp_sample, w["x"], w["t"], type="csd")
        t2 = time.perf_counter()
        t_ll += t1 - t0
        t_pr += t2 - t1
        reps += 1
    # the reference projects trial by trial on strided slices lfp[:, :, r] (gpcsd2d.py:147-148); the oracle above uses
    # contiguous trials and one batched matmul ("fair" flavour).  Price the reference's access pattern too.
    nxs = lfp_sample.shape[0]
    Ks = O.spatial_kphi(geom, hp) + hp["jitter"] * np.eye(nxs)
    Kt = O.temporal_sum(hp["temporal"], geom.t)
    Qs, Qt, _ = O.eig_D(Ks, Kt, hp["sig2n"])
    ts = time.perf_counter()
    for r in range(min(2, Rs)):
        np.dot(np.dot(Qs.T, lfp_sample[:, :, r]), Qt)
    strided_ms = (time.perf_counter() - ts) * 1e3 / min(2, Rs)
    Yc = np.ascontiguousarray(np.moveaxis(lfp_sample, 2, 0))
    ts = time.perf_counter()
    for r in range(min(2, Rs)):
        np.dot(np.dot(Qs.T, Yc[r]), Qt)
    contiguous_ms = (time.perf_counter() - ts) * 1e3 / min(2, Rs)
    return {"value": Rs * reps / (t_ll + t_pr), "unit": "trials/s", "cores": int(cores), "kind": "port",
            "projection_ms_per_trial_reference_layout": strided_ms, "projection_ms_per_trial_contiguous": contiguous_ms,
            "sample": "%d trials x %d reps of oracle loglik+predict(csd) at the bench geometry (NumPy %s, BLAS threads=%d)"
                      % (Rs, reps, np.__version__, cores),
            "loglik_evals_per_sec": reps / t_ll, "predict_trials_per_sec": Rs * reps / t_pr}, ll


# Epilogue template index of each profiled GEMM role (last template argument of gemm_f64_kernel in rocprofv3's names)
GEMM_EPI_OF = {"gemm_pred_tstar": 0,   # all temporal components in one plain-store GEMM (largest EPI 0 launch of the step)
               "gemm_pred_temporal_div": 1, "gemm_proj_temporal_quad": 2}
PMC_PROFILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")


def pmc_traffic(prof_name):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate passes over this same bench command; tools/pmc_summary.py).  None if no profile is committed."""
    epi = GEMM_EPI_OF.get(prof_name)
    if epi is None or not os.path.exists(PMC_PROFILE):
        return None, None
    with open(PMC_PROFILE) as fh:
        rows = json.load(fh)["rows"]
    def epi_of(kernel):                      # gemm_f64_kernel<WM, WN, FM, FN, BK, TA, TB, EPI, NBUF>
        args = kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")
        return int(args[7]) if len(args) >= 8 else -1
    cand = [r for r in rows if r["kernel"].startswith("gemm_f64_kernel<") and epi_of(r["kernel"]) == epi
            and "hbm_traffic_bytes_per_launch" in r]
    if not cand:
        return None, None
    r = max(cand, key=lambda r: r["hbm_traffic_bytes_per_launch"])
    return r["hbm_traffic_bytes_per_launch"], {"read": r["hbm_read_bytes_per_launch"], "write": r["hbm_write_bytes_per_launch"],
                                              "source": "profiles/" + os.path.basename(PMC_PROFILE), "kernel": r["kernel"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--trials-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-trials", type=int, default=8)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): GPCSD_BENCH_BACKEND=gloo + GPCSD_DEVICE=0 run N ranks on one card
    backend = os.environ.get("GPCSD_BENCH_BACKEND", "nccl")
    if "GPCSD_DEVICE" in os.environ:
        local_rank = int(os.environ["GPCSD_DEVICE"])
    import torch
    torch.cuda.set_device(local_rank)
    sharding = None
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world

    from gpcsd_amd import _hip
    from gpcsd_amd.dist import TrialSharding
    w = workload(args.workload)
    R_local = args.trials_per_gpu or w["trials_per_gpu"]
    if world > 1:
        sharding = TrialSharding()

    # ---- synthetic resident data (each rank draws its own block of trials) ----
    m = build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    m.set_device(local_rank)
    lfp = synth_data(w, m, R_local, seed=1000 + rank)
    m.update_lfp(lfp, w["t"])
    ctx = m._sync_device()
    z = w["x"]
    C = len(m.temporal_cov_list)
    R_total = R_local * n_gpus

    def one_step():
        # hyper-parameters originate on rank 0 (tiny broadcast), every rank re-assembles Ks/Kt/eig deterministically
        hp, keep = m._hparams(m.JITTER)
        hp0, keep0 = m._hparams(0.0)
        if sharding is not None:
            vec = np.array([hp.R, hp.eps, hp.ell_s[0], hp.ell_s[1]] + [hp.ell_t[i] for i in range(C)]
                           + [hp.sigma2_t[i] for i in range(C)] + [float(keep[0])])
            vec = sharding.broadcast(vec, src=0)
            kinds = [hp.kind[i] for i in range(C)]
            temporal = [(kinds[i], vec[4 + i], vec[4 + C + i]) for i in range(C)]
            hp, keep = ctx.make_hparams(vec[0], vec[1], vec[2:4], temporal, vec[4 + 2 * C], m.JITTER)
            hp0, keep0 = ctx.make_hparams(vec[0], vec[1], vec[2:4], temporal, vec[4 + 2 * C], 0.0)
        ta = time.perf_counter()
        sumlog, quad = ctx.loglik_parts(hp)
        pending = sharding.allreduce_sum_async(np.array([quad])) if sharding is not None else None
        tb = time.perf_counter()
        ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        tc = time.perf_counter()
        if pending is not None:                  # the global log-likelihood is consumed after predict was queued
            quad = float(pending()[0])
        ll = -0.5 * R_total * sumlog - 0.5 * quad
        return ll, tb - ta, tc - tb

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as td
            td.barrier()

    # Setup, not warm-up: a fresh process on a cold box needs a few evaluations before it is in steady state (first call
    # eager + allocations, second captured into hipGraphs, third replayed; GPU clocks and host caches ramp over the first
    # tenths of a second -- a 10-step timed region measured 4.6 ms/step as the first command on a fresh box against 2.63
    # afterwards).  A fixed number of untimed evaluations (~0.4 s; the same count on every rank, each step carries
    # collectives), then the W warm-up steps the contract asks for, then K timed.
    for _ in range(150):
        one_step()
    for _ in range(args.warmup):
        ll, _, _ = one_step()
    fence()
    t0 = time.perf_counter()
    t_ll = t_pr = 0.0
    for _ in range(args.steps):
        ll, a, b = one_step()
        t_ll += a
        t_pr += b
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as td
        tt = torch.tensor([elapsed, t_ll, t_pr], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, t_ll, t_pr = (float(v) for v in tt.cpu())

    # ---- PCIe-inclusive variant (outputs copied to host arrays), rank-local, a few steps ----
    hp0, keep0 = m._hparams(0.0)
    t1 = time.perf_counter()
    n_pcie = 2
    for _ in range(n_pcie):
        ctx.predict(hp0, z, w["t"], _hip.PRED_CSD, (z.shape[0], w["nt"], R_local))
    pcie_predict = R_local * n_pcie / (time.perf_counter() - t1)

    # ---- roofline: profiled pass (HIP events on the library stream, per named kernel) ----
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(3):
        one_step()
    ctx.prof_enable(False)
    prof = ctx.prof_all()
    gemms = {k: v for k, v in prof.items() if k.startswith("gemm_") and v["count"] > 0}
    roof = None
    if gemms and rank == 0:
        name = max(gemms, key=lambda k: gemms[k]["ms"])
        g = gemms[name]
        avg_ms = g["ms"] / g["count"]
        ach = (g["flops"] / g["count"]) / (avg_ms * 1e-3) / 1e12
        mfma_meas = ctx.mfma_f64_peak()
        all_gemm_tf = sum(v["flops"] for v in gemms.values()) / (sum(v["ms"] for v in gemms.values()) * 1e-3) / 1e12
        traffic, traffic_detail = pmc_traffic(name)
        roof = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_SPEC_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_SPEC_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_detail": traffic_detail, "kernel": "gemm_f64_kernel [" + name + "]",
                "avg_launch_ms": avg_ms, "flops_per_launch": g["flops"] / g["count"],
                "measured_mfma_f64_peak_tflops": mfma_meas, "all_gemm_tflops": all_gemm_tf,
                "per_kernel_ms_per_step": {k: v["ms"] / 3.0 for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}

    if rank != 0:
        return
    f_ll, f_pred, f_pred_trial = algorithmic_flops(w, R_local, z.shape[0], C)
    out = {
        "metric": "gpcsd_loglik_plus_predict_trials_per_sec",
        "value": R_total * args.steps / elapsed,
        "unit": "trials/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["label"], "n_elec": w["nx"], "n_t": w["nt"], "trials_per_gpu": R_local,
                   "total_trials": R_total, "predict": "z=electrodes, t*=t, type=csd, %d temporal components" % C,
                   "parallelism": "trial-sharded x%d" % n_gpus},
        "loglik_evals_per_sec": args.steps / t_ll,
        "loglik_trial_evals_per_sec": R_total * args.steps / t_ll,
        "predict_trials_per_sec": R_total * args.steps / t_pr,
        "pcie_inclusive_predict_trials_per_sec_per_gpu": pcie_predict,
        # flops of the reference's own (dense, Kronecker-structured) algorithm for this step; the library executes about
        # half of the GEMM part (folded basis) and a quarter of the eigensolver part (symmetry folding), so the second
        # figure is a rate in units of the reference's work, not an MFMA utilisation -- that is roofline.frac
        "reference_algorithm_gflop_per_step_per_gpu": (f_ll + f_pred) / 1e9,
        "reference_algorithm_tflops_equivalent_per_gpu": (f_ll + f_pred) / (elapsed / args.steps) / 1e12,
        "loglik": float(ll),
    }
    if roof:
        out["roofline"] = roof
    if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported at N=1 only
        cb, ll_cpu = cpu_baseline(w, m, lfp[:, :, :args.cpu_sample_trials])
        out["cpu_baseline"] = cb
        # parity spot check beside the numbers: GPU loglik on the same sample vs the oracle
        m2 = build_model(w, lfp[:, :, :args.cpu_sample_trials].copy())
        for tc, tc0 in zip(m2.temporal_cov_list, m.temporal_cov_list):
            tc.params["sigma2"]["value"] = tc0.params["sigma2"]["value"]
        m2.set_device(local_rank)
        out["parity_rel_err_loglik_vs_oracle"] = abs(float(m2.loglik()) - ll_cpu) / abs(ll_cpu)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
