"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz).

CPU-only.  Tolerances: the oracle calls the same LAPACK/BLAS as the reference, but
contracts trials in a different order, so agreement is to rounding (1e-10 relative
on well-scaled cases), far inside the 1e-6 gate of BASELINE.json.
"""
import numpy as np
import pytest

import cases as C
from helpers import golden, load_model_case, relerr, with_jitter
from oracle import gpcsd_oracle as O

MODEL_CASES = list(C.model_cases().keys())


def test_b_fwd():
    g = golden("ops")
    assert relerr(O.b_fwd_1d(g["bf1_r"], 100.0), g["bf1_out0"]) < 1e-15
    assert relerr(O.b_fwd_1d(g["bf1_r"], 37.5), g["bf1_out1"]) < 1e-15
    assert relerr(O.b_fwd_2d(g["bf2_d1"], g["bf2_d2"], 100.0, 80.0), g["bf2_out_R100_e80"]) < 1e-15
    assert relerr(O.b_fwd_2d(g["bf2_d1"], g["bf2_d2"], 30.0, 5.0), g["bf2_out_R30_e5"]) < 1e-15
    assert relerr(O.b_fwd_2d(None, None, 100.0, 80.0, w=g["bf2_w"]), g["bf2_out_w"]) < 1e-15


@pytest.mark.parametrize("n", [20, 30, 60, 100, 120])
def test_gauss_legendre(n):
    g = golden("ops")
    gx, gw = O.gauss_legendre(-200.0, 2600.0, n)
    assert np.array_equal(gx, g["gl_x_%d" % n])
    assert np.array_equal(gw, g["gl_w_%d" % n])


def test_temporal_gram():
    g = golden("ops")
    t, tp = g["kt_t"], g["kt_tp"]
    assert relerr(O.temporal_gram(O.SE, t, t, 4.5, 1.7), g["kt_se_default"]) < 1e-15
    assert relerr(O.temporal_gram(O.SE, tp, t, 4.5, 1.7), g["kt_se_t_tp"]) < 1e-15
    assert relerr(O.temporal_gram(O.SE, t, tp, 4.5, 1.7), g["kt_se_tp_only"]) < 1e-15
    assert relerr(O.temporal_gram(O.MATERN, t, t, 2.5, 0.6), g["kt_ma_default"]) < 1e-15
    assert relerr(O.temporal_gram(O.MATERN, tp, t, 2.5, 0.6), g["kt_ma_t_tp"]) < 1e-15


@pytest.mark.parametrize("tag,a,b,ngl", [("a", 0.0, 2300.0, 100), ("b", -200.0, 2600.0, 30)])
def test_spatial_1d(tag, a, b, ngl):
    g = golden("ops")
    x, z, xp = g["s1_x"], g["s1_z"], g["s1_xp"]
    gx, gw = O.gauss_legendre(a, b, ngl)
    assert relerr(O.ks_csd_1d(x, 200.0), g["s1%s_Ks" % tag]) < 1e-15
    assert relerr(O.kphi_1d(x, gx, gw, 100.0, 200.0), g["s1%s_Kphi" % tag]) < 1e-13
    assert relerr(O.kphi_1d(x, gx, gw, 100.0, 200.0, xp=xp), g["s1%s_Kphi_xp" % tag]) < 1e-13
    assert relerr(O.kphig_1d(x, gx, gw, z, 100.0, 200.0), g["s1%s_Kphig" % tag]) < 1e-13


def test_spatial_2d():
    g = golden("ops")
    x, z, xp = g["s2_x"], g["s2_z"], g["s2_xp"]
    gx1, gw1 = O.gauss_legendre(0.0, 48.0, 10)
    gx2, gw2 = O.gauss_legendre(0.0, 440.0, 24)
    assert np.array_equal(O.expand_grid(gx1, gx2), g["s2_gl_x_grid"])
    assert relerr(np.prod(O.expand_grid(gw1, gw2), axis=1, keepdims=True), g["s2_gl_w_prod"]) < 1e-15
    assert relerr(O.ks_csd_2d(x, 30.0, 100.0), g["s2_Ks"]) < 1e-15
    assert relerr(O.kphi_2d(x, gx1, gw1, gx2, gw2, 60.0, 20.0, 30.0, 100.0), g["s2_Kphi"]) < 1e-13
    assert relerr(O.kphi_2d(x, gx1, gw1, gx2, gw2, 60.0, 20.0, 30.0, 100.0, xp=xp), g["s2_Kphi_xp"]) < 1e-13
    assert relerr(O.kphig_2d(x, gx1, gw1, gx2, gw2, z, 60.0, 20.0, 30.0, 100.0), g["s2_Kphig"]) < 1e-13
    assert relerr(O.kphi_2d(xp, gx1, gw1, gx2, gw2, 60.0, 20.0, 30.0, 100.0), g["s2_Kphi_after_reset"]) < 1e-13


def test_eig_D_and_utils():
    g = golden("ops")
    Qs, Qt, D = O.eig_D(g["eig_Ks"], g["eig_Kt"], 0.3)
    assert relerr(D, g["eig_D_scalar"]) < 1e-14
    assert relerr(O.eig_D(g["eig_Ks"], g["eig_Kt"], g["eig_siglist"])[2], g["eig_D_list"]) < 1e-14
    assert relerr(Qs @ np.diag(g["eig_es"]) @ Qs.T, g["eig_Ks"]) < 1e-13
    assert relerr(Qt.T @ Qt, np.eye(14)) < 1e-13
    assert np.array_equal(O.mykron(g["kron_A"], g["kron_B"]), g["kron_out"])
    assert np.array_equal(O.expand_grid([1.0, 2.0, 3.5], [-1.0, 0.5]), g["expand_grid_out"])


def test_priors():
    g = golden("ops")
    a, b = O.invgamma_params(1.2, 80.0)
    assert relerr([a, b], g["ig_alpha_beta"]) < 1e-15
    for v, ref_ig, ref_hn in zip(g["prior_x"], g["ig_lpdf"], g["hn_lpdf"]):
        assert O.invgamma_lpdf(v, a, b) == pytest.approx(ref_ig, rel=1e-15) or (np.isinf(ref_ig) and np.isinf(O.invgamma_lpdf(v, a, b)))
        assert O.halfnormal_lpdf(v, 0.1) == pytest.approx(ref_hn, rel=1e-15) or (np.isinf(ref_hn) and np.isinf(O.halfnormal_lpdf(v, 0.1)))


def test_fwd_models():
    g = golden("ops")
    assert relerr(O.fwd_model_1d(g["fm1_arr"], g["fm1_x"], g["fm1_z"], 150.0), g["fm1_out"]) < 1e-13
    assert relerr(O.fwd_model_2d(g["fm2_arr"], g["fm2_x1"], g["fm2_x2"], g["fm2_z"], 60.0, 20.0), g["fm2_out"]) < 1e-13


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_loglik(name):
    c, g, geom, hp, lfp = load_model_case(name)
    Ks = O.spatial_kphi(geom, hp)
    assert relerr(np.diag(Ks), g["Ks_diag"]) < 1e-12
    assert relerr(Ks[0], g["Ks_row0"]) < 1e-12
    Kt = O.temporal_sum(hp["temporal"], geom.t)
    assert relerr(Kt[0], g["Kt_row0"]) < 1e-14
    ll = O.loglik(geom, with_jitter(hp, float(g["jitter"])), lfp)
    assert abs(ll - float(g["loglik"])) / abs(float(g["loglik"])) < 1e-9
    if "Dvec" in g.files:
        Qs, Qt, D = O.eig_D(Ks + float(g["jitter"]) * np.eye(Ks.shape[0]), Kt, hp["sig2n"])
        assert relerr(D, g["Dvec"]) < 1e-10


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if np.ndim(C.model_cases()[n]["sig2n"]) == 0
                                  and C.model_cases()[n]["x"].shape[0] * C.model_cases()[n]["t"].shape[0] <= 24 * 500])
def test_model_loglik_tridiagonal_form(name):
    """The log-likelihood without the temporal eigenvectors -- Ks decomposed, Kt only tridiagonalised, shifted tridiagonal
    LDL^T per spatial eigen-row (what the device's staged path evaluates) -- restated in NumPy/SciPy: equal to
    the eigen form (gpcsd1d.py:113-128) to the rounding both carry and to the reference's golden value."""
    c, g, geom, hp, lfp = load_model_case(name)
    hpj = with_jitter(hp, float(g["jitter"]))
    ll_tri = O.loglik_tridiagonal(geom, hpj, lfp)
    ll = O.loglik(geom, hpj, lfp)
    assert abs(ll_tri - ll) <= 2e-9 * abs(ll)          # (both forms carry the conditioning of a numerically singular Ks)
    assert abs(ll_tri - float(g["loglik"])) / abs(float(g["loglik"])) < 1e-9


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if not C.model_cases()[n].get("loglik_only")])
def test_model_predict(name):
    c, g, geom, hp, lfp = load_model_case(name)
    out = O.predict(geom, hp, lfp, c["x"], c["t"], type="both")
    tol = 1e-7   # the reference inverts a dense (nx*nt)^2 matrix; its own rounding is ~1e-9 of max|pred|
    assert relerr(out["csd"], g["csd_pred"]) < tol
    if c.get("predict_light"):
        assert relerr(out["lfp"][:, :, :1], g["lfp_pred_trial0"]) < tol
        return
    assert relerr(out["lfp"], g["lfp_pred"]) < tol
    for i in range(len(hp["temporal"])):
        assert relerr(out["csd_list"][i], g["csd_pred_%d" % i]) < tol
        assert relerr(out["lfp_list"][i], g["lfp_pred_%d" % i]) < tol
    o2 = O.predict(geom, hp, lfp, g["z2"], c["t"], type="csd")
    assert set(o2.keys()) == {"csd", "csd_list"}
    assert relerr(o2["csd"], g["csd_pred_z2"]) < tol
    o3 = O.predict(geom, hp, lfp, g["z2"], g["tq"], type="lfp")
    assert relerr(o3["lfp"], g["lfp_pred_z2_tq"]) < tol
    with pytest.raises(ValueError):
        O.predict(geom, hp, lfp, g["z2"], c["t"][:-1], type="csd")


def test_sample_prior():
    g = golden("sample_prior")
    c, gm, geom, hp, lfp = load_model_case("cfg1_1d_24x100x1")
    out, Ls, Lt = O.sample_prior_from_normals(geom, hp, g["sp1_normals"], which="csd", jitter=1e-8)
    assert relerr(out, g["sp1_csd"]) < 1e-10
    c, gm, geom, hp, lfp = load_model_case("2d_grid_48x40x2")
    out, Ls, Lt = O.sample_prior_from_normals(geom, hp, g["sp2_normals"], which="csd", jitter=1e-7)
    assert relerr(out, g["sp2_csd"]) < 1e-10
    assert bool(g["sp2_lfp_isnan"])


def test_dense_cholesky_crosscheck():
    """Kronecker-eigen loglik == dense Cholesky loglik for scalar sig2n (SURVEY 'three facts' item 1)."""
    c, g, geom, hp, lfp = load_model_case("1d_odd_17x37x5")
    hpj = with_jitter(hp, 1e-8)
    Ks = O.spatial_kphi(geom, hpj) + 1e-8 * np.eye(17)
    Kt = O.temporal_sum(hp["temporal"], geom.t)
    a = O.loglik_from_K(lfp, Ks, Kt, hp["sig2n"])
    b = O.loglik_dense_cholesky(lfp, Ks, Kt, hp["sig2n"])
    assert abs(a - b) / abs(b) < 1e-9


def test_shift_objective_vs_reference_pieces():
    """N4: the trial-shift objective (auditory_lfp/fit_mean_function.py:311-321) on the decomposition the reference's own
    comp_eig_D produced with a per-electrode noise list."""
    g = golden("shift_objective")
    Qs, Qt, D = O.eig_D(g["Ks"], g["Kt"], g["sig2n"])
    assert relerr(D, g["Dvec"]) < 1e-10
    for ti in range(g["lfp"].shape[2]):
        for k, tau in enumerate(g["taus"]):
            got = O.shift_objective(g["Qs"], g["Qt"], g["Dvec"], g["lfp"][:, :, ti], g["mu_lfp"], g["t"], tau,
                                    float(g["mutau"]), float(g["sigtau"]))
            assert abs(got - g["nll"][ti, k]) <= 1e-11 * abs(g["nll"][ti, k])
            # with the oracle's own eigenvectors (sign / ordering conventions of this LAPACK build) as well
            got2 = O.shift_objective(Qs, Qt, D, g["lfp"][:, :, ti], g["mu_lfp"], g["t"], tau, float(g["mutau"]),
                                     float(g["sigtau"]))
            assert abs(got2 - g["nll"][ti, k]) <= 1e-8 * abs(g["nll"][ti, k])
    q = O.whitened_quad(g["Qs"], g["Qt"], g["Dvec"], g["resid"][:, :, :, 1])
    prior = 0.5 * np.sum(np.square((g["taus"][1] - g["mutau"]) / g["sigtau"]))
    assert np.allclose(0.5 * q + prior, g["nll"][:, 1], rtol=1e-11, atol=0)


def test_traditional_csd_estimators_vs_reference():
    g = golden("trad_csd")
    np.testing.assert_array_equal(O.trad_csd_1d(g["lfp1"]), g["csd1"])
    np.testing.assert_array_equal(O.trad_csd_1d(g["lfp1e"]), g["csd1e"])
    np.testing.assert_array_equal(O.trad_csd_2d(g["lfp2"]), g["csd2"])


def _tparams_of(hp):
    tp = [np.log(hp["R"] / 100.0)] + [np.log(e / 100.0) for e in hp["ell_s"]]
    for _, ell, s2 in hp["temporal"]:
        tp += [np.log(ell), np.log(s2)]
    return np.array(tp + list(np.log(np.atleast_1d(hp["sig2n"]))))


@pytest.mark.parametrize("name", ["cfg1_1d_24x100x1", "1d_siglist_12x40x4", "1d_odd_17x37x5", "2d_grid_48x40x2"])
def test_closed_form_gradient_vs_central_differences_of_the_pinned_loglik(name):
    """O.loglik_and_grad (the closed form of what gpcsd1d.py:211 / gpcsd2d.py:250 `jac=grad(obj_fun)` differentiates) against
    4th-order central differences of O.loglik, which the fixtures above pin to the reference: scalar noise, a per-electrode
    noise list (eigenvector-rotation term), 1D and 2D forward models, SE and Matern components, R held fixed."""
    c, g, geom, hp, lfp = load_model_case(name)
    jit = float(g["jitter"])
    kinds = [k for k, _, _ in hp["temporal"]]
    n_sig = int(np.size(hp["sig2n"]))
    tp = _tparams_of(hp)
    tp = tp + 0.05 * np.cos(np.arange(tp.size))          # away from the fixture's own point, every parameter moved
    ll, grad = O.loglik_and_grad(geom, lfp, tp, kinds, n_sig, eps=c["eps"], jitter=jit)
    f = lambda v: O.loglik(geom, O.hparams_from_tparams(v, geom.dim, kinds, n_sig, c["eps"], jit), lfp)
    assert ll == f(tp)
    h, fd = 1e-3, np.zeros_like(tp)
    for i in range(tp.size):
        e = np.zeros_like(tp)
        e[i] = h
        fd[i] = (8.0 * (f(tp + e) - f(tp - e)) - (f(tp + 2 * e) - f(tp - 2 * e))) / (12.0 * h)
    assert np.max(np.abs(grad - fd)) / np.max(np.abs(fd)) < 1e-7, (grad, fd)
    ll_fix, grad_fix = O.loglik_and_grad(geom, lfp, tp, kinds, n_sig, eps=c["eps"], jitter=jit, R_fixed=hp["R"])
    assert grad_fix[0] == 0.0 and ll_fix == f(np.concatenate([[np.log(hp["R"] / 100.0)], tp[1:]]))
    # prior derivatives used beside it by the fit checkers
    for x in (0.3, 2.0, 40.0):
        a, b = O.invgamma_params(1.0, 20.0)
        assert abs(O.invgamma_dlpdf(x, a, b) - (O.invgamma_lpdf(x + 1e-6, a, b) - O.invgamma_lpdf(x - 1e-6, a, b)) / 2e-6) < 1e-6 * (1 + abs(O.invgamma_dlpdf(x, a, b)))
        assert abs(O.halfnormal_dlpdf(x, 1.5) - (O.halfnormal_lpdf(x + 1e-6, 1.5) - O.halfnormal_lpdf(x - 1e-6, 1.5)) / 2e-6) < 1e-6 * (1 + x)


def test_the_reference_script_recipe_end_to_end_with_the_oracle():
    """tests/golden/recipe_1d.npz is the reference run through its own script recipe (simulation_studies/sim_from_gp_1D.py:49-70,
    100-110, 151-156: sample_prior -> fwd_model_1d -> noise + normalize -> new model -> predict -> MSE / R^2).  The oracle's pieces
    chained the same way reproduce every stored stage (the RNG stream is NumPy's: the normals are re-drawn from the stored generator
    state -- the reference's constructors draw initial hyper-parameters from the same global stream first)."""
    g = golden("recipe_1d")
    n, t, x, z = int(g["ntrials"]), g["t"], g["x"], g["z"]
    R, ellSE, sig2tM, elltM, sig2tSE, elltSE, sig2n = (float(v) for v in g["hyp"])
    temporal = [(O.SE, elltSE, sig2tSE), (O.MATERN, elltM, sig2tM)]
    np.random.set_state(("MT19937", g["rng_key_before_sample_prior"], int(g["rng_pos_before_sample_prior"][0]),
                         int(g["rng_pos_before_sample_prior"][1]), float(g["rng_gauss_before_sample_prior"])))
    normals = np.stack([np.random.normal(0, 1, (z.shape[0], t.shape[0])) for _ in range(2 * n)], axis=2)
    gen = O.Geometry1D(z, t)
    hp_gen = O.make_hparams(R, (ellSE,), temporal, sig2n)
    csd, _, _ = O.sample_prior_from_normals(gen, hp_gen, normals, which="csd", jitter=1e-8)
    assert relerr(csd, g["csd"]) < 1e-10
    lfp = np.stack([O.fwd_model_1d(csd[:, :, r], z, x, R) for r in range(2 * n)], axis=2)
    assert relerr(lfp, g["lfp_forward"]) < 1e-12
    noise = np.random.normal(0, np.sqrt(sig2n), size=lfp.shape)
    assert np.array_equal(noise, g["noise"])
    lfp = lfp + noise
    lfp = lfp / np.max(np.abs(lfp), axis=(0, 1))
    assert relerr(lfp, g["lfp"]) < 1e-12
    geom = O.Geometry1D(x, t)
    hp = O.make_hparams(R, (ellSE,), temporal, sig2n)
    assert abs(O.loglik(geom, with_jitter(hp, 1e-8), lfp[:, :, n:]) - float(g["loglik"])) <= 1e-9 * abs(float(g["loglik"]))
    pred = O.predict(geom, hp, lfp[:, :, n:], x[1:-1], t, type="csd")
    # (sig2n = 1e-4 as the script has it: K = Ks (x) Kt + 1e-4 I is conditioned ~1e9, and the reference's dense (nx nt)^2 algebra
    # (gpcsd1d.py:262-265) and the structured form differ by their own rounding, ~1e-6 of the posterior mean)
    assert relerr(pred["csd"], g["csd_pred"]) < 5e-6 and relerr(pred["csd_list"], g["csd_pred_list"]) < 5e-6
    nrm = lambda a: a / np.max(np.abs(a), axis=(0, 1))
    truth, est = nrm(g["csd_interior"][1:-1, :, n:]), nrm(pred["csd"][1:-1])
    mse = np.nanmean(np.square(est - truth), axis=(0, 1))
    rsq = 1 - np.sum(np.square(est - truth), axis=(0, 1)) / np.sum(np.square(truth), axis=(0, 1))
    print("recipe: mse rel dev %.1e, R^2 rel dev %.1e" % (relerr(mse, g["mse"]), relerr(rsq, g["rsq"])))
    assert relerr(mse, g["mse"]) < 1e-4 and relerr(rsq, g["rsq"]) < 1e-7
