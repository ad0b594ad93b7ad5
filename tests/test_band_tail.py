"""Band tail (round 5): the temporal side of a fused call reduced to HALF-BANDWIDTH 4 instead of a tridiagonal matrix
(gpcsd_amd/csrc/sytrd_bandtail.hpp), the shifted banded systems of the log-likelihood (gpcsd1d.py:113-128) and of the prediction
(gpcsd1d.py:248-293) in gpcsd_amd/csrc/band.hip.  The form is switchable (gpcsd_band_tail) and off by default: these tests hold it to
the tridiagonal form, the eigenvector form and the oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import gpcsd_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _kt(n):
    t = np.arange(n) * 0.4
    return 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)


@pytest.mark.parametrize("n", [9, 13, 64, 188, 192, 193, 231, 250])
def test_band_reduction_on_the_device_vs_numpy(n):
    """gpcsd_debug_sybrd: A = Q B Q^T with B of half-bandwidth 4 (register block alone up to 192 rows, LDS strip beyond), Q from
    the reflectors orthogonal, the spectrum that of A -- for the temporal Gram matrices of the path, random symmetric matrices
    and a numerically rank-one matrix (reflectors that are the identity)."""
    from gpcsd_amd import _hip
    ctx = _hip.default_context()
    rs = np.random.RandomState(n)
    G = rs.standard_normal((n, n))
    for A, tol in ((_kt(n), 2e-13), (G + G.T, 2e-13), (np.ones((n, n)) + 1e-3 * np.eye(n), 1e-11)):
        band, V, tau = ctx.debug_sybrd(A)
        Q, B = O.band_q(V, tau), O.band_dense(band)
        sc = np.max(np.abs(A))
        assert np.max(np.abs(Q.T @ Q - np.eye(n))) < 1e-13 * n
        assert np.max(np.abs(Q.T @ A @ Q - B)) <= tol * n * sc
        assert np.max(np.abs(np.linalg.eigvalsh(B) - np.linalg.eigvalsh(A))) <= tol * n * sc
        # reflector k has its support from row k + 4 on (the layout the compact-WY back-transformation takes)
        assert np.all(np.triu(np.ones((n, n)), 4) * V == V)


def _step_model(R, name="cfg3"):
    import bench
    w = bench.workload(name)
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, R, seed=7)
    m.update_lfp(lfp, w["t"])
    return w, m, lfp


@pytest.mark.parametrize("name,R", [("cfg3", 16), ("cfg2", 24), ("npx69", 20)])
def test_fused_calls_in_the_band_form_vs_tridiagonal_form_and_oracle(name, R):
    """loglik() and predict() with the band form on: equal to the tridiagonal form (same basis, another reduction of the temporal
    side) to rounding, and to the oracle within the gates of the tridiagonal form's own tests; the counter shows the chains took it."""
    import bench
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(R, name)
    O_, geom, hp, hp0 = bench.oracle_setup(w, m)
    ctx = m._sync_device()
    z = w.get("z", w["x"])
    ctx.decomposition_cache(False)
    out = {}
    for on in (False, True):
        ctx.band_tail(on)
        n0 = ctx.band_tail()
        ll = float(m.loglik())
        m.predict(z, w["t"], type="csd")
        out[on] = (ll, np.array(m.csd_pred), np.array(m.csd_pred_list[1]), ctx.band_tail() - n0)
    ctx.band_tail(False)
    assert out[False][3] == 0 and out[True][3] >= 2            # both calls' temporal chains took the band form
    ll_ref = O.loglik(geom, hp, lfp)
    ref = O.predict(geom, hp0, lfp, z, w["t"], type="csd")
    sc = np.max(np.abs(ref["csd"]))
    print(name, "loglik: band vs tridiagonal %.1e, band vs oracle %.1e; predict: band vs tridiagonal %.1e, band vs oracle %.1e" % (
        abs(out[True][0] - out[False][0]) / abs(ll_ref), abs(out[True][0] - ll_ref) / abs(ll_ref),
        np.max(np.abs(out[True][1] - out[False][1])) / sc, np.max(np.abs(out[True][1] - ref["csd"])) / sc))
    assert abs(out[True][0] - out[False][0]) <= 1e-11 * abs(ll_ref)
    assert abs(out[True][0] - ll_ref) <= 1e-9 * abs(ll_ref)
    assert np.max(np.abs(out[True][1] - out[False][1])) <= 1e-9 * sc
    assert np.max(np.abs(out[True][1] - ref["csd"])) <= 1e-8 * sc
    assert np.max(np.abs(out[True][2] - ref["csd_list"][1])) <= 1e-8 * np.max(np.abs(ref["csd_list"][1]))


def test_paired_call_in_the_band_form_is_bitwise_its_fenced_calls():
    """gpcsd_loglik_predict_async with the band form: the queued pair gives the bits of the two calls fenced one by one (the band
    form changes the reduction, not the determinism), over a few steps with changing hyper-parameters."""
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(12)
    ctx = m._sync_device()
    ctx.pair_share_s(False)        # (bit-for-bit against the fenced calls: the pair decomposes both spatial matrices, as they do)
    ctx.decomposition_cache(False)
    ctx.band_tail(True)
    try:
        z = w["x"]
        fenced, queued = [], []
        for step in range(3):
            m.temporal_cov_list[0].params["ell"]["value"] = 20.0 + step
            hp, k1 = m._hparams(m.JITTER)
            hp0, k0 = m._hparams(0.0)
            sl, qd = ctx.loglik_parts(hp)
            ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            ctx.synchronize()
            fenced.append((sl, qd, ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], 12)).copy()))
        for step in range(3):
            m.temporal_cov_list[0].params["ell"]["value"] = 20.0 + step
            hp, k1 = m._hparams(m.JITTER)
            hp0, k0 = m._hparams(0.0)
            ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            sl, qd = ctx.loglik_parts_wait()
            ctx.synchronize()
            queued.append((sl, qd, ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], 12)).copy()))
        for f, q in zip(fenced, queued):
            assert f[0] == q[0] and f[1] == q[1] and np.array_equal(f[2], q[2])
        assert not np.array_equal(fenced[0][2], fenced[1][2])
    finally:
        ctx.band_tail(False)
