"""The reference's own 1D workload (auditory_lfp/fit_gpcsd_baseline.py:79-105): GPCSD1D, 24 electrodes, integration limits
(-200, 2600), an SE + a Matern temporal component and ONE NOISE PRIOR PER ELECTRODE (a 24-entry sig2n list -> 30 parameters),
fit(n_restarts) then predict(x, t).  The list ties noise variance x to EIGEN-RANK x of Ks (utility_functions.py:54-63), so
the evaluation runs in merged eigen-order (no folded GEMMs) and the spatial gradient carries the eigenvector-rotation term.
Round 5: gpcsd_loglik_grad_batch takes B sets with a list each, so the restarts of this fit advance in lock-step like cfg5's."""
import os
import sys

import numpy as np
import pytest

from oracle import gpcsd_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GATE = 1e-6


def _aud24(ntrials, seed=11):
    import bench
    w = bench.workload("aud24")
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, ntrials, seed=seed)
    m.update_lfp(lfp, w["t"])
    O_, geom, hp, hp0 = bench.oracle_setup(w, m)
    return w, m, lfp, geom, hp, hp0


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(np.asarray(b))))


def test_aud24_shape_loglik_gradient_and_predict_vs_oracle():
    """loglik (gpcsd1d.py:113-128), the 30-entry gradient against central differences of the oracle (incl. the 24 noise
    components), predict(type='both') at the 24 electrodes and at 100 depths (gpcsd1d.py:248-293).  With a noise list the
    objective depends on the ORDER of Ks's eigenvalues, which for the noise-level ones is decided by the eigensolver's rounding:
    the gates are north_star's 1e-6 or three times what LAPACK's own drivers disagree by on the same inputs, both printed."""
    w, m, lfp, geom, hp, hp0 = _aud24(12)
    ll = float(m.loglik())
    ll_ref = O.loglik(geom, hp, lfp)
    spread_ll = O.driver_spread(lambda: O.loglik(geom, hp, lfp))
    print("aud24 loglik rel err %.2e (driver spread %.2e)" % (abs(ll - ll_ref) / abs(ll_ref), spread_ll))
    assert abs(ll - ll_ref) / abs(ll_ref) <= max(1e-9, 3 * spread_ll)
    # gradient in log-parameters vs central differences of the oracle
    ll2, g_nat = m._loglik_and_grad_natural()
    assert g_nat.shape == (1 + 1 + 4 + 24,) and abs(ll2 - ll) <= 1e-12 * abs(ll)
    kinds = [k for k, _, _ in w["temporal"]]
    tp = m._current_tparams()
    vals = np.exp(tp) * np.array([100.0, 100.0] + [1.0] * 28)
    fd = O.loglik_grad_fd(geom, lfp, tp, kinds, 24, eps=0.0, jitter=1e-8, h=1e-5)
    err = float(np.max(np.abs(g_nat * vals - fd)) / np.max(np.abs(fd)))
    err_noise = float(np.max(np.abs((g_nat * vals - fd)[6:])) / np.max(np.abs(fd)))
    print("aud24 gradient vs oracle central differences: %.2e of the largest component (noise entries alone %.2e)" % (err, err_noise))
    assert err < 2e-5, (g_nat * vals, fd)
    for z, tag in ((w["x"], "24 electrodes"), (w["z100"], "100 depths")):
        m.predict(z, w["t"], type="both")
        ref = O.predict(geom, hp0, lfp, z, w["t"], type="both")
        spread = O.driver_spread(lambda: O.predict(geom, hp0, lfp, z, w["t"], type="csd")["csd"])
        e = {"csd": _rel(m.csd_pred, ref["csd"]), "lfp": _rel(m.lfp_pred, ref["lfp"]),
             "csd_list1": _rel(m.csd_pred_list[1], ref["csd_list"][1])}
        print("aud24 predict at %s:" % tag, {k: "%.2e" % v for k, v in e.items()}, "driver spread %.2e" % spread)
        assert m.csd_pred.shape == (z.shape[0], 500, 12)
        assert max(e.values()) <= max(GATE, 3 * spread), e


def test_loglik_grad_batch_with_noise_lists_is_bitwise_the_sequential_evaluation():
    """Five hyper-parameter sets, each with its own 24-entry noise list, through ONE chain of launches: every set gets exactly the
    bits of a gpcsd_loglik_grad call of its own -- value, the 6 leading entries (with the rotation term) and the 24 noise entries
    -- and the objective / gradient of the class API are the same alone and in a batch."""
    w, m, lfp, geom, hp, hp0 = _aud24(8)
    ctx = m._sync_device()
    rs = np.random.RandomState(2)
    tp0 = m._current_tparams()
    tps = [tp0 + 0.1 * rs.standard_normal(tp0.size) for _ in range(5)]
    ng = 1 + 1 + 4 + 24
    hps, keep, seq = [], [], []
    for tp in tps:
        m._set_from_tparams(tp, False)
        h, k = m._hparams(m.JITTER)
        assert h.n_sig2n == 24
        hps.append(h)
        keep.append(k)
        seq.append(ctx.loglik_grad(h, ng))
    sumlog, quad, grad, st = ctx.loglik_grad_batch(hps, ng)
    assert np.all(st == 0)
    for b in range(5):
        assert sumlog[b] == seq[b][0] and quad[b] == seq[b][1] and np.array_equal(grad[b], seq[b][2])
    assert not np.array_equal(grad[0], grad[1])
    # the class API: batch of one == scalar call == its slot in a batch of five
    assert m._batch_can_evaluate() and m._vector_glue_applies()
    one = [m._objective_and_grad(tp, False) for tp in tps]
    many = m._objective_and_grad_batch(list(enumerate(tps)), False)
    for b in range(5):
        assert many[b][0] == one[b][0] and np.array_equal(many[b][1], one[b][1])
    # ... and the vectorised host glue agrees with the dict-walking one (log-prior, chain rule) to rounding
    m._set_from_tparams(tps[3], False)
    lp = m._log_prior()
    llg = m._loglik_and_grad_natural()
    f_ref, g_ref = -(llg[0] + lp), m._chain_rule(tps[3], llg[1], False)
    assert abs(f_ref - one[3][0]) <= 1e-12 * abs(f_ref) and np.max(np.abs(g_ref - one[3][1])) <= 1e-10 * np.max(np.abs(g_ref))


def test_aud24_fit_restarts_run_in_lockstep_and_equal_the_sequential_loop():
    """fit(n_restarts=6) with the noise list: all restarts in one lock-step batch (far fewer device calls than evaluations), and
    the optima are those of the one-after-the-other loop of the reference (gpcsd1d.py:193-220), bit for bit."""
    w, m, lfp, geom, hp, hp0 = _aud24(8)
    np.random.seed(5)
    starts = [m._sample_start(False) for _ in range(6)]
    opts = {"maxiter": 5, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}
    m.fit(n_restarts=6, options=opts, starts=starts)
    nb, npts = m.fit_batches_
    nll_batched = np.array(m.fit_nll_values_, dtype=float)
    # (a tick of the lock-step driver serves every live chain once, so the number of batches is the LONGEST chain's evaluation
    # count: with these wild starts -- objective values of 1e8 -- one chain spends 32 evaluations in its line searches, the six
    # together 79)
    assert m.fit_driver_used_ == "setulb" and nb <= npts / 2
    w2, m2, *_ = _aud24(8)
    m2.fit(n_restarts=6, options=opts, starts=starts, batch=1)
    assert np.array_equal(nll_batched, np.array(m2.fit_nll_values_, dtype=float))
    assert np.array_equal(np.asarray(m.sig2n["value"]), np.asarray(m2.sig2n["value"]))
