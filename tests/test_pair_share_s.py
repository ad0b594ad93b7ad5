"""The paired call's shared spatial side (round 5): loglik adds jitter * I to Ks (gpcsd1d.py:116, gpcsd2d.py:139), predict does not
(gpcsd1d.py:258, gpcsd2d.py:296) -- eigh(Ks + j I) has the eigenvectors of eigh(Ks) and its spectrum shifted by j
(utility_functions.py:58-59), so a pair with equal spatial hyper-parameters decomposes ONE matrix (gpcsd_pair_share_s).  Exact
algebra, and off by default (it measured slower end to end); these tests bound what it does in floating point: against
the pair with two decompositions, the fenced calls and the oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import gpcsd_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _step_model(R, name):
    import bench
    w = bench.workload(name)
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, R, seed=5)
    m.update_lfp(lfp, w["t"])
    return w, m, lfp


@pytest.mark.parametrize("name,R", [("cfg3", 16), ("cfg3", 12), ("cfg3", 80), ("cfg2", 24), ("npx69", 20), ("npx72sym", 20)])
def test_pair_with_one_spatial_decomposition_vs_two_and_oracle(name, R):
    """The queued pair with the spatial side shared and not: the log-likelihood is the same bits (its own decomposition either way),
    the prediction agrees to the eigensolver's backward error and with the oracle within the usual gate.  Folded and unfolded spatial
    sides (npx69's sites do not share the electrodes' symmetry), tridiagonal and eigenvector form of the prediction (12 trials), both
    sets in the eigenvector form (80 trials: past the size where the tridiagonal forms switch themselves off; W is shared there too)."""
    import bench
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(R, name)
    O_, geom, hp_o, hp0_o = bench.oracle_setup(w, m)
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    z = np.ascontiguousarray(w.get("z", w["x"]))
    hp, k1 = m._hparams(m.JITTER)
    hp0, k0 = m._hparams(0.0)
    out = {}
    try:
        for on in (False, True):
            ctx.pair_share_s(on)
            n0 = ctx.pair_share_s()
            ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            sl, qd = ctx.loglik_parts_wait()
            ctx.synchronize()
            out[on] = (sl, qd, ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], R)).copy(), ctx.pair_share_s() - n0)
    finally:
        ctx.pair_share_s(False)
    assert out[False][3] == 0 and out[True][3] == 1
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]          # the log-likelihood: bit for bit
    ref = O.predict(geom, hp0_o, lfp, z, w["t"], type="csd")["csd"]
    sc = np.max(np.abs(ref))
    d_pair = np.max(np.abs(out[True][2] - out[False][2])) / sc
    d_or = np.max(np.abs(out[True][2] - ref)) / sc
    d_or2 = np.max(np.abs(out[False][2] - ref)) / sc
    print(name, R, "prediction: shared vs two decompositions %.1e; vs oracle %.1e (two decompositions: %.1e)" % (d_pair, d_or, d_or2))
    assert d_pair <= 1e-10
    assert d_or <= 1e-8


def test_shared_spatial_side_needs_equal_spatial_hyperparameters():
    """Different spatial length scales in the two sets: two decompositions, as ever (the counter does not move), and the pair equals
    its fenced calls bit for bit."""
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(16, "cfg3")
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    z = np.ascontiguousarray(w["x"])
    hp, k1 = m._hparams(m.JITTER)
    m.spatial_cov.params["ell1"]["value"] *= 1.1
    hp0, k0 = m._hparams(0.0)
    ctx.pair_share_s(True)
    n0 = ctx.pair_share_s()
    ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    sl, qd = ctx.loglik_parts_wait()
    ctx.synchronize()
    q = ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], 16)).copy()
    assert ctx.pair_share_s() == n0
    sl2, qd2 = ctx.loglik_parts(hp)
    ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    ctx.synchronize()
    f = ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], 16)).copy()
    assert sl == sl2 and qd == qd2 and np.array_equal(q, f)
