#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Run from the repo root (only where /root/reference exists; never on the GPU box):

    PYTHONPATH=tests/golden/_shim:/root/reference/src python tests/golden/generate_goldens.py

The reference is imported, never copied: the fixtures hold inputs and the
reference's outputs only.  `autograd` (HIPS) is not installed and there is no
network, so `tests/golden/_shim/autograd` maps `autograd.numpy` to NumPy
(forward-only; `grad` raises) -- SURVEY.md section 8c.  NumPy/SciPy versions used
are recorded in every fixture.
"""
import os
import sys

import numpy as np
import scipy
import scipy.integrate

if not hasattr(scipy.integrate, "trapz"):          # removed in SciPy >= 1.14; reference calls it
    scipy.integrate.trapz = scipy.integrate.trapezoid

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402

from gpcsd.gpcsd1d import GPCSD1D  # noqa: E402
from gpcsd.gpcsd2d import GPCSD2D  # noqa: E402
from gpcsd import covariances as rcov  # noqa: E402
from gpcsd import forward_models as rfm  # noqa: E402
from gpcsd import utility_functions as ruf  # noqa: E402
from gpcsd import priors as rpr  # noqa: E402

META = dict(numpy=np.__version__, scipy=scipy.__version__)


def build_ref_model(c, lfp):
    np.random.seed(0)
    tcl = []
    for kind, ell, s2 in c["temporal"]:
        tc = rcov.GPCSDTemporalCovSE(c["t"]) if kind == C.SE else rcov.GPCSDTemporalCovMatern(c["t"])
        tc.params["ell"]["value"] = ell
        tc.params["sigma2"]["value"] = s2 if s2 is not None else 1.0
        tcl.append(tc)
    if c["dim"] == 1:
        m = GPCSD1D(lfp, c["x"], c["t"], a=c["a"], b=c["b"], ngl=c["ngl"], temporal_cov_list=tcl)
        m.spatial_cov.params["ell"]["value"] = c["ell_s"][0]
    else:
        m = GPCSD2D(lfp, c["x"], c["t"], ngl1=c["ngl1"], ngl2=c["ngl2"], temporal_cov_list=tcl, eps=c["eps"])
        m.spatial_cov.params["ell1"]["value"] = c["ell_s"][0]
        m.spatial_cov.params["ell2"]["value"] = c["ell_s"][1]
    m.R["value"] = c["R"]
    m.sig2n["value"] = c["sig2n"]
    if "temporal_sigma2_rel" in c:          # sigma2_c = rel_c / mean(diag(Ks))
        Ks = m.spatial_cov.compKphi_2d(R=c["R"], eps=c["eps"])
        md = np.mean(np.diag(Ks))
        for tc, rel in zip(tcl, c["temporal_sigma2_rel"]):
            tc.params["sigma2"]["value"] = rel / md
    return m


def gen_models():
    for name, c in C.model_cases().items():
        lfp = C.case_lfp(c)
        m = build_ref_model(c, lfp)
        out = dict(META)
        out["lfp_checksum"] = np.array([lfp.sum(), np.abs(lfp).sum()])
        out["temporal_sigma2"] = np.array([tc.params["sigma2"]["value"] for tc in m.temporal_cov_list])
        if c["dim"] == 1:
            Ks = m.spatial_cov.compKphi_1d(c["R"])
            jitter = 1e-8
        else:
            Ks = m.spatial_cov.compKphi_2d(R=c["R"], eps=c["eps"])
            jitter = 1e-7
        Kt = sum(tc.compute_Kt() for tc in m.temporal_cov_list)
        out["jitter"] = jitter
        if Ks.shape[0] <= 96:
            out["Ks"] = Ks
        out["Ks_diag"] = np.diag(Ks).copy()
        out["Ks_row0"] = Ks[0].copy()
        if Kt.shape[0] <= 120:
            out["Kt"] = Kt
        out["Kt_row0"] = Kt[0].copy()
        Qs, Qt, D = ruf.comp_eig_D(Ks + jitter * np.eye(Ks.shape[0]), Kt, c["sig2n"])
        out["evals_s"] = np.linalg.eigvalsh(Ks + jitter * np.eye(Ks.shape[0]))
        out["evals_t"] = np.linalg.eigvalsh(Kt)
        if D.size <= 20000:
            out["Dvec"] = D
        out["loglik"] = np.array(float(m.loglik()))
        if c.get("predict_light", False):
            # big case: keep the fixture small -- summed predictions only, lfp for the first trial only
            m.predict(c["x"], c["t"], type="both")
            out["csd_pred"] = m.csd_pred
            out["lfp_pred_trial0"] = m.lfp_pred[:, :, :1]
        elif not c.get("loglik_only", False):
            z = c["x"]
            m.predict(z, c["t"], type="both")
            out["csd_pred"] = m.csd_pred
            out["lfp_pred"] = m.lfp_pred
            for i, a in enumerate(m.csd_pred_list):
                out["csd_pred_%d" % i] = a
            for i, a in enumerate(m.lfp_pred_list):
                out["lfp_pred_%d" % i] = a
            # different prediction sites, csd only
            if c["dim"] == 1:
                z2 = np.linspace(c["a"], c["b"], 31)[:, None]
            else:
                z2 = C.grid_xy(3, 7, c["x"][:, 0].min(), c["x"][:, 0].max(), c["x"][:, 1].min(), c["x"][:, 1].max())
            m.predict(z2, c["t"], type="csd")
            out["z2"] = z2
            out["csd_pred_z2"] = m.csd_pred
            # t* != t with equal length (axis quirk)
            tq = c["t"] + 0.37 * (c["t"][1] - c["t"][0])
            m.predict(z2, tq, type="lfp")
            out["tq"] = tq
            out["lfp_pred_z2_tq"] = m.lfp_pred
        np.savez_compressed(os.path.join(HERE, "model_%s.npz" % name), **out)
        print("wrote", name, "loglik", out["loglik"])


def gen_ops():
    out = dict(META)
    # G1 forward weights
    r = np.concatenate([[0.0], np.linspace(-2500, 2500, 41), [1e-9, 1e6]])[:, None] * np.ones((1, 3))
    out["bf1_r"] = r
    out["bf1_R"] = np.array([100.0, 37.5])
    out["bf1_out0"] = rfm.b_fwd_1d(r, 100.0)
    out["bf1_out1"] = rfm.b_fwd_1d(r, 37.5)
    d1 = np.linspace(-50, 50, 23)[:, None] * np.ones((1, 19))
    d2 = np.ones((23, 1)) * np.linspace(-3000, 3000, 19)[None, :]
    d1[11, 9] = 0.0
    d2[11, 9] = 0.0
    out["bf2_d1"], out["bf2_d2"] = d1, d2
    out["bf2_out_R100_e80"] = rfm.b_fwd_2d(d1, d2, 100.0, 80.0)
    out["bf2_out_R30_e5"] = rfm.b_fwd_2d(d1, d2, 30.0, 5.0)
    w = np.abs(np.linspace(0, 4000, 77))[None, :]
    out["bf2_w"] = w
    out["bf2_out_w"] = rfm.b_fwd_2d(None, None, 100.0, 80.0, w=w)
    # G2 GL rules after the affine map
    for n in (20, 30, 60, 100, 120):
        sc = rcov.GPCSD1DSpatialCov(np.linspace(0, 1, 4)[:, None], -200.0, 2600.0, n)
        out["gl_x_%d" % n], out["gl_w_%d" % n] = sc.gl_x, sc.gl_w
    # G4 temporal Grams, t/tprime given and defaulted
    np.random.seed(1)
    t = np.linspace(0, 30, 61)[:, None]
    tp = np.linspace(-3, 41, 47)[:, None]
    se = rcov.GPCSDTemporalCovSE(t)
    se.params["ell"]["value"], se.params["sigma2"]["value"] = 4.5, 1.7
    ma = rcov.GPCSDTemporalCovMatern(t)
    ma.params["ell"]["value"], ma.params["sigma2"]["value"] = 2.5, 0.6
    out["kt_t"], out["kt_tp"] = t, tp
    out["kt_se_default"] = se.compute_Kt()
    out["kt_se_t_tp"] = se.compute_Kt(tp, t)
    out["kt_se_tp_only"] = se.compute_Kt(tprime=tp)
    out["kt_ma_default"] = ma.compute_Kt()
    out["kt_ma_t_tp"] = ma.compute_Kt(tp, t)
    # G3 spatial operators 1D
    x = np.linspace(0, 2300, 24)[:, None]
    z = np.linspace(-100, 2400, 40)[:, None]
    xp = np.linspace(50, 2250, 11)[:, None]
    for tag, (a, b, ngl) in {"a": (0.0, 2300.0, 100), "b": (-200.0, 2600.0, 30)}.items():
        sc = rcov.GPCSD1DSpatialCovSE(x, a=a, b=b, ngl=ngl)
        sc.params["ell"]["value"] = 200.0
        out["s1%s_Ks" % tag] = sc.compute_Ks()
        out["s1%s_Kphi" % tag] = sc.compKphi_1d(100.0)
        out["s1%s_Kphi_xp" % tag] = sc.compKphi_1d(100.0, xp=xp)
        out["s1%s_Kphig" % tag] = sc.compKphig_1d(z, 100.0)
    out["s1_x"], out["s1_z"], out["s1_xp"] = x, z, xp
    # G3 spatial operators 2D
    x2 = C.grid_xy(4, 12, 0.0, 48.0, 0.0, 440.0)
    z2 = C.grid_xy(5, 9, -5.0, 53.0, 10.0, 400.0)
    xp2 = C.neuropixels_xy(20)
    sc = rcov.GPCSD2DSpatialCovSE(x2, a1=0.0, b1=48.0, a2=0.0, b2=440.0, ngl1=10, ngl2=24)
    sc.params["ell1"]["value"], sc.params["ell2"]["value"] = 30.0, 100.0
    out["s2_x"], out["s2_z"], out["s2_xp"] = x2, z2, xp2
    out["s2_gl_x_grid"] = sc.gl_x_grid
    out["s2_gl_w_prod"] = sc.gl_w_prod
    out["s2_Ks"] = sc.compute_Ks()
    out["s2_Kphi"] = sc.compKphi_2d(60.0, 20.0)
    out["s2_Kphi_xp"] = sc.compKphi_2d(60.0, 20.0, xp=xp2)
    out["s2_Kphig"] = sc.compKphig_2d(z2, 60.0, 20.0)
    # reset_x keeps the GL grid, swaps electrodes
    sc.reset_x(xp2)
    out["s2_Kphi_after_reset"] = sc.compKphi_2d(60.0, 20.0)
    # G5 comp_eig_D on a well-conditioned pair + list sigma
    rs = np.random.RandomState(5)
    M = rs.standard_normal((9, 9))
    Ks = M @ M.T + 9 * np.eye(9)
    M = rs.standard_normal((14, 14))
    Kt = M @ M.T + np.diag(np.arange(14.0))
    out["eig_Ks"], out["eig_Kt"] = Ks, Kt
    Qs, Qt, D = ruf.comp_eig_D(Ks, Kt, 0.3)
    out["eig_D_scalar"] = D
    out["eig_es"], out["eig_et"] = np.linalg.eigvalsh(Ks), np.linalg.eigvalsh(Kt)
    sl = np.linspace(0.1, 0.9, 9)
    out["eig_siglist"] = sl
    out["eig_D_list"] = ruf.comp_eig_D(Ks, Kt, sl)[2]
    # mykron / expand_grid / reduce_grid / normalize / sort_grid
    A = rs.standard_normal((3, 4))
    B = rs.standard_normal((5, 2))
    out["kron_A"], out["kron_B"], out["kron_out"] = A, B, ruf.mykron(A, B)
    out["expand_grid_out"] = ruf.expand_grid(np.array([1.0, 2.0, 3.5]), np.array([-1.0, 0.5]))
    g = rs.permutation(C.grid_xy(3, 4, 0, 2, 0, 3))
    out["grid_perm"] = g
    out["sort_grid_out"] = ruf.sort_grid(g)
    r1, r2 = ruf.reduce_grid(g)
    out["reduce_grid_1"], out["reduce_grid_2"] = r1, r2
    arr = rs.standard_normal((6, 7, 3))
    out["normalize_in"], out["normalize_out"] = arr, ruf.normalize(arr)
    # G8 priors
    ig = rpr.GPCSDInvGammaPrior()
    ig.set_params(1.2, 80.0)
    out["ig_alpha_beta"] = np.array([ig.alpha, ig.beta])
    xs = np.array([-1.0, 0.0, 0.3, 5.0, 100.0])
    out["prior_x"] = xs
    out["ig_lpdf"] = np.array([ig.lpdf(v) for v in xs])
    hn = rpr.GPCSDHalfNormalPrior(0.1)
    out["hn_lpdf"] = np.array([hn.lpdf(v) for v in xs])
    # N3 trapezoid forward simulators (next-row operators, pinned now while the reference is importable)
    xd = np.linspace(0, 2300, 50)[:, None]
    zz = np.linspace(0, 2300, 24)[:, None]
    arr = rs.standard_normal((50, 6))
    out["fm1_arr"], out["fm1_x"], out["fm1_z"] = arr, xd, zz
    out["fm1_out"] = rfm.fwd_model_1d(arr, xd, zz, 150.0)
    x1 = np.linspace(0, 48, 5)[:, None]
    x2v = np.linspace(0, 400, 11)[:, None]
    arr2 = rs.standard_normal((5, 11, 3))
    zz2 = C.grid_xy(2, 6, 0, 48, 0, 400)
    out["fm2_arr"], out["fm2_x1"], out["fm2_x2"], out["fm2_z"] = arr2, x1, x2v, zz2
    out["fm2_out"] = rfm.fwd_model_2d(arr2, x1, x2v, zz2, 60.0, 20.0)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("wrote ops")


def gen_sample_prior():
    out = dict(META)
    cs = C.model_cases()
    # 1D: the reference draws np.random.normal(0,1,(nx,nt)) per trial (gpcsd1d.py:306-308)
    c = cs["cfg1_1d_24x100x1"]          # SE + Matern keeps Kt numerically positive definite
    m = build_ref_model(c, C.case_lfp(c))
    np.random.seed(77)
    out["sp1_csd"] = m.sample_prior(3)
    np.random.seed(77)
    out["sp1_normals"] = np.stack([np.random.normal(0, 1, (24, 100)) for _ in range(3)], axis=2)
    # 2D: np.random.seed(seed); normal(0,1,(nx,nt,ntrials)) (gpcsd2d.py:338,354)
    c = cs["2d_grid_48x40x2"]
    m = build_ref_model(c, C.case_lfp(c))
    out["sp2_temporal_sigma2"] = np.array([tc.params["sigma2"]["value"] for tc in m.temporal_cov_list])
    csd, lfp = m.sample_prior(2, type="csd", seed=5)
    out["sp2_csd"] = csd
    out["sp2_lfp_isnan"] = np.array(bool(np.all(np.isnan(lfp))))
    np.random.seed(5)
    out["sp2_normals"] = np.random.normal(0, 1, (48, 40, 2))
    np.savez_compressed(os.path.join(HERE, "sample_prior.npz"), **out)
    print("wrote sample_prior")


def gen_shift_objective():
    """Fixture for the downstream per-trial consumer of (Qs, Qt, Dvec): the trial-shift objective of
    auditory_lfp/fit_mean_function.py.  That script cannot be imported (it loads Zenodo data at module level), so the
    pieces it takes from the package are run here exactly as it runs them -- the model with a per-electrode noise list,
    Ks + 1e-8 I, Kt summed over the components and the reference's own comp_eig_D (lines 87-106) -- and its objective
    (lines 311-321: background + time-shifted component means, residual whitened by Qs / Qt / Dvec, Gaussian prior on the
    shifts) is evaluated on synthetic component means."""
    import scipy.interpolate
    out = dict(META)
    rs = np.random.RandomState(41)
    nx, nt, ntrials, nseg = 12, 40, 5, 2
    x = np.linspace(0, 1100, nx)[:, None]
    t = np.linspace(0, 39, nt)[:, None]
    lfp = rs.standard_normal((nx, nt, ntrials))
    np.random.seed(0)
    spatial_cov = rcov.GPCSD1DSpatialCovSE(x, a=-100.0, b=1200.0)
    matern_cov = rcov.GPCSDTemporalCovMatern(t)
    se_cov = rcov.GPCSDTemporalCovSE(t)
    sig2n_prior = [rpr.GPCSDHalfNormalPrior(0.1) for _ in range(nx)]
    m = GPCSD1D(lfp, x, t, sig2n_prior=sig2n_prior, spatial_cov=spatial_cov, temporal_cov_list=[se_cov, matern_cov],
                a=-100.0, b=1200.0)
    m.R["value"] = 90.0
    m.spatial_cov.params["ell"]["value"] = 70.0
    se_cov.params["ell"]["value"], se_cov.params["sigma2"]["value"] = 6.0, 0.8
    matern_cov.params["ell"]["value"], matern_cov.params["sigma2"]["value"] = 3.0, 0.4
    m.sig2n["value"] = np.linspace(0.02, 0.3, nx)
    Kt = m.temporal_cov_list[0].compute_Kt() + m.temporal_cov_list[1].compute_Kt()
    Ks = m.spatial_cov.compKphi_1d(m.R["value"]) + 1e-8 * np.eye(nx)
    Qs, Qt, Dvec = ruf.comp_eig_D(Ks, Kt, m.sig2n["value"])
    # synthetic background + two evoked components (smooth bumps), then the shifted-mean objective
    tt = t.squeeze()
    mu_lfp = np.zeros((nx, nt, nseg + 1))
    mu_lfp[:, :, 0] = 0.1 * rs.standard_normal((nx, 1)) * np.ones((1, nt))
    for i in range(1, nseg + 1):
        prof = np.exp(-0.5 * np.square((np.arange(nx) - 3.0 * i) / 2.0))[:, None]
        mu_lfp[:, :, i] = prof * np.exp(-0.5 * np.square((tt - 10.0 * i) / 3.0))[None, :]
    mu_f = {i: scipy.interpolate.interp1d(tt, mu_lfp[:, :, i], axis=1, fill_value="extrapolate") for i in range(1, nseg + 1)}
    mutau, sigtau = 0.0, 10.0
    taus = np.array([[0.0, 0.0], [1.5, -2.0], [-3.25, 0.75], [6.0, 4.0]])
    nll = np.zeros((ntrials, taus.shape[0]))
    resid = np.zeros((nx, nt, ntrials, taus.shape[0]))
    for ti in range(ntrials):
        for k, tau in enumerate(taus):
            mu_new = np.copy(mu_lfp[:, :, 0])
            for i in range(1, nseg + 1):
                mu_new += mu_f[i](tt + tau[i - 1])
            r = lfp[:, :, ti] - mu_new
            alpha = np.reshape(np.linalg.multi_dot([Qs.T, r, Qt]), (nx * nt))
            nll[ti, k] = 0.5 * np.sum(alpha ** 2 / Dvec) + 0.5 * np.sum(np.square((tau - mutau) / sigtau))
            resid[:, :, ti, k] = r
    out.update(x=x, t=t, lfp=lfp, mu_lfp=mu_lfp, taus=taus, mutau=mutau, sigtau=sigtau, Ks=Ks, Kt=Kt,
               sig2n=m.sig2n["value"], Qs=Qs, Qt=Qt, Dvec=Dvec, nll=nll, resid=resid)
    np.savez_compressed(os.path.join(HERE, "shift_objective.npz"), **out)
    print("wrote shift_objective")


def gen_trad_csd():
    """Traditional second-difference CSD estimators (predict_csd.py), the comparison baseline of the simulation studies
    (sim_from_gp_1D.py:88, sim_from_gp_2D.py:141) and of the auditory analysis (fit_gpcsd_baseline.py:106,147)."""
    from gpcsd import predict_csd as rpc
    out = dict(META)
    rs = np.random.RandomState(7)
    lfp1 = rs.standard_normal((24, 50, 6))
    lfp1e = rs.standard_normal((2, 5, 3))                    # no interior electrode at all
    lfp2 = rs.standard_normal((5, 9, 20, 4))
    out.update(lfp1=lfp1, csd1=rpc.predictcsd_trad_1d(lfp1), lfp1e=lfp1e, csd1e=rpc.predictcsd_trad_1d(lfp1e),
               lfp2=lfp2, csd2=rpc.predictcsd_trad_2d(lfp2))
    np.savez_compressed(os.path.join(HERE, "trad_csd.npz"), **out)
    print("wrote trad_csd")


def gen_recipe_1d():
    """The reference's own script recipe end to end (BASELINE cfg1's "plumbing"), in the order of
    simulation_studies/sim_from_gp_1D.py: generator model on a dense grid (:49-56) -> sample_prior (:59) -> interpolation to the
    interior electrodes (:60-63) -> fwd_model_1d per trial (:66-68) -> white noise + normalize (:69-70) -> new model on the
    held-out half with the generating hyper-parameters (:100-107) -> predict(xshort, t) (:110) -> MSE / R^2 against the ground
    truth (:151-156).  kCSD and the plots are left out; 4 + 4 trials instead of 50 + 50 (fixture size).  Every array a drop-in
    backend must reproduce is stored: the sampled CSD (RNG stream included), the simulated LFP, the prediction, the two metrics."""
    import scipy.interpolate
    out = dict(META)
    np.random.seed(1)
    ntrials = 4
    a, b, nt, nx, nz = 0, 2300, 60, 24, 100
    t = np.linspace(0, nt, nt)[:, None]
    x = np.linspace(a, b, nx)[:, None]
    xshort = x[1:-1]
    z = np.linspace(a, b, nz)[:, None]
    hyp = dict(R=100.0, ellSE=200.0, sig2tM=0.7, elltM=5.0, sig2tSE=0.5, elltSE=20.0, sig2n=0.0001)

    def set_true(m):
        m.R['value'] = hyp["R"]
        m.sig2n['value'] = hyp["sig2n"]
        m.spatial_cov.params['ell']['value'] = hyp["ellSE"]
        m.temporal_cov_list[0].params['ell']['value'] = hyp["elltSE"]
        m.temporal_cov_list[0].params['sigma2']['value'] = hyp["sig2tSE"]
        m.temporal_cov_list[1].params['ell']['value'] = hyp["elltM"]
        m.temporal_cov_list[1].params['sigma2']['value'] = hyp["sig2tM"]
    gen = GPCSD1D(np.zeros((nz, nt)), z, t, temporal_cov_list=[rcov.GPCSDTemporalCovSE(t), rcov.GPCSDTemporalCovMatern(t)])
    set_true(gen)
    st = np.random.get_state()                            # (the constructor has drawn from the global stream: the position is data)
    out["rng_key_before_sample_prior"], out["rng_pos_before_sample_prior"] = st[1], np.array([st[2], st[3]])
    out["rng_gauss_before_sample_prior"] = np.array(st[4])
    csd = gen.sample_prior(2 * ntrials)
    csd_interior = np.zeros((nx - 2, nt, 2 * ntrials))
    for trial in range(2 * ntrials):
        interp = scipy.interpolate.RectBivariateSpline(z, t, csd[:, :, trial])
        csd_interior[:, :, trial] = interp(xshort, t)
    lfp = np.zeros((nx, nt, 2 * ntrials))
    for trial in range(2 * ntrials):
        lfp[:, :, trial] = rfm.fwd_model_1d(csd[:, :, trial], z, x, hyp["R"])
    lfp_clean = lfp.copy()
    noise = np.random.normal(0, np.sqrt(hyp["sig2n"]), size=(nx, nt, 2 * ntrials))
    lfp = ruf.normalize(lfp + noise)
    model = GPCSD1D(lfp[:, :, ntrials:], x, t)
    set_true(model)
    model.predict(xshort, t)
    truth = ruf.normalize(csd_interior[1:-1, :, ntrials:])
    pred = ruf.normalize(model.csd_pred[1:-1, :, :])
    out.update(ntrials=ntrials, t=t, x=x, z=z, hyp=np.array([hyp[k] for k in ("R", "ellSE", "sig2tM", "elltM", "sig2tSE", "elltSE", "sig2n")]),
               csd=csd, csd_interior=csd_interior, lfp_forward=lfp_clean, noise=noise, lfp=lfp, csd_pred=model.csd_pred,
               csd_pred_list=np.stack(model.csd_pred_list), loglik=np.array(model.loglik()),
               mse=np.nanmean(np.square(pred - truth), axis=(0, 1)),
               rsq=1 - np.sum(np.square(pred - truth), axis=(0, 1)) / np.sum(np.square(truth), axis=(0, 1)))
    np.savez_compressed(os.path.join(HERE, "recipe_1d.npz"), **out)
    print("wrote recipe_1d: mse %s  rsq %s" % (out["mse"], out["rsq"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["shift_objective", "ops", "sample_prior", "models", "trad_csd", "recipe_1d"]
    for name in which:                                       # a subset regenerates only those fixtures
        {"shift_objective": gen_shift_objective, "ops": gen_ops, "sample_prior": gen_sample_prior, "models": gen_models,
         "trad_csd": gen_trad_csd, "recipe_1d": gen_recipe_1d}[name]()
