"""Pipelined orthogonal factor (round 5): the register tail of a staged temporal chain publishes its progress every 64
reflectors, and the T factor, the finished columns of Q (utility_functions.py:58: Kt = Q T Q^T) and the matching columns of
X = Y~ Q follow on another stream while the tail is still reducing the next panel (gpcsd_amd/csrc/wy.hip: wy_q_pipeline).  Switchable
(gpcsd_q_pipeline); these tests hold it to the unpipelined form, to the oracle and to itself across call forms."""
import os
import sys

import numpy as np
import pytest

from oracle import gpcsd_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _step_model(R, name="cfg3"):
    import bench
    w = bench.workload(name)
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, R, seed=11)
    m.update_lfp(lfp, w["t"])
    return w, m, lfp


@pytest.mark.parametrize("name,R", [("cfg3", 16), ("cfg3", 50), ("cfg2", 24), ("npx69", 20)])
def test_fused_calls_with_the_pipeline_vs_without_and_oracle(name, R):
    """loglik() and predict() with stage 5 on and off: the same reflectors and T factors, Q accumulated forward instead of backward --
    equal to rounding; both within the oracle gates of the tridiagonal form's own tests; the counter shows the chains took it.
    Four panels at 250-row halves (cfg3, cfg2), three at 188 (the reference's 2D shape)."""
    import bench
    w, m, lfp = _step_model(R, name)
    O_, geom, hp, hp0 = bench.oracle_setup(w, m)
    ctx = m._sync_device()
    z = w.get("z", w["x"])
    ctx.decomposition_cache(False)
    out = {}
    try:
        for on in (False, True):
            ctx.q_pipeline(on)
            n0 = ctx.q_pipeline()
            ll = float(m.loglik())
            m.predict(z, w["t"], type="csd")
            out[on] = (ll, np.array(m.csd_pred), np.array(m.csd_pred_list[1]), ctx.q_pipeline() - n0)
    finally:
        ctx.q_pipeline(True)
    assert out[False][3] == 0 and out[True][3] >= 2            # both calls' temporal chains took it
    ll_ref = O.loglik(geom, hp, lfp)
    ref = O.predict(geom, hp0, lfp, z, w["t"], type="csd")
    sc = np.max(np.abs(ref["csd"]))
    print(name, R, "loglik: pipelined vs not %.1e, vs oracle %.1e; predict: pipelined vs not %.1e, vs oracle %.1e" % (
        abs(out[True][0] - out[False][0]) / abs(ll_ref), abs(out[True][0] - ll_ref) / abs(ll_ref),
        np.max(np.abs(out[True][1] - out[False][1])) / sc, np.max(np.abs(out[True][1] - ref["csd"])) / sc))
    assert abs(out[True][0] - out[False][0]) <= 1e-11 * abs(ll_ref)
    assert abs(out[True][0] - ll_ref) <= 1e-9 * abs(ll_ref)
    assert np.max(np.abs(out[True][1] - out[False][1])) <= 1e-9 * sc
    assert np.max(np.abs(out[True][1] - ref["csd"])) <= 1e-8 * sc
    assert np.max(np.abs(out[True][2] - ref["csd_list"][1])) <= 1e-8 * np.max(np.abs(ref["csd_list"][1]))


@pytest.mark.parametrize("R,calls", [(12, 0), (16, 18)])
def test_paired_call_with_the_pipeline_is_bitwise_its_fenced_calls_and_replays(R, calls):
    """gpcsd_loglik_predict_async with stage 5: the queued pair gives the bits of the two calls fenced one by one (each of which
    pipelines its own chain), step after step with changing hyper-parameters and again when the captured graphs are replayed;
    no gate ran out of time (status 7 would raise).  With 12 trials the prediction takes the eigenvector form and nobody pipelines (a pair
    whose chain runs all four stages keeps stage 3, and a log-likelihood gives the same bits alone and in a pair); with 16 the
    tridiagonal form, sharing the log-likelihood's X."""
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(R)
    ctx = m._sync_device()
    ctx.pair_share_s(False)        # (bit-for-bit against the fenced calls: the pair decomposes both spatial matrices, as they do)
    ctx.decomposition_cache(False)
    z = w["x"]
    n0 = ctx.q_pipeline()

    def run(queued):
        res = []
        for step in range(6):
            m.temporal_cov_list[0].params["ell"]["value"] = 20.0 + (step % 3)
            hp, k1 = m._hparams(m.JITTER)
            hp0, k0 = m._hparams(0.0)
            if queued:
                ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
                sl, qd = ctx.loglik_parts_wait()
            else:
                sl, qd = ctx.loglik_parts(hp)
                ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            ctx.synchronize()
            res.append((sl, qd, ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], R)).copy()))
        return res

    fenced, queued = run(False), run(True)
    assert ctx.q_pipeline() - n0 == calls
    for f, q in zip(fenced, queued):
        assert f[0] == q[0] and f[1] == q[1] and np.array_equal(f[2], q[2])
    for k in range(3):                                        # the same hyper-parameters three steps later: replayed graphs, same bits
        assert queued[k][0] == queued[k + 3][0] and queued[k][1] == queued[k + 3][1] and np.array_equal(queued[k][2], queued[k + 3][2])
    assert not np.array_equal(fenced[0][2], fenced[1][2])


@pytest.mark.parametrize("nt", [131, 258, 387, 512])
def test_pipeline_over_panel_counts_and_odd_grids(nt):
    """Halves of 66 / 65 rows (one panel and the two-column remainder), 129 / 129 (two panels), 194 / 193 (the halves differ in their
    number of panels) and 256 / 256 (a full 64-row strip: the first panel is published by the strip phase), 1D model, against the
    oracle and the unpipelined form."""
    import bench
    w = bench.workload("cfg2")
    w["nt"] = nt
    w["t"] = 0.5 * np.arange(float(nt))[:, None]
    R = 16                                   # (the pipeline applies where the prediction takes the tridiagonal form too: >= 16 trials)
    m = bench.build_model(w, np.zeros((w["nx"], nt, 1)))
    lfp = bench.synth_data(w, m, R, seed=nt)
    m.update_lfp(lfp, w["t"])
    O_, geom, hp, hp0 = bench.oracle_setup(w, m)
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    ctx.ll_tridiag(1)
    vals = {}
    try:
        for on in (False, True):
            ctx.q_pipeline(on)
            n0 = ctx.q_pipeline()
            ll = float(m.loglik())
            m.predict(w["x"], w["t"], type="csd")
            vals[on] = (ll, np.array(m.csd_pred), ctx.q_pipeline() - n0)
    finally:
        ctx.q_pipeline(True)
        ctx.ll_tridiag(2)
    assert vals[True][2] >= 1 and vals[False][2] == 0
    ll_ref = O.loglik(geom, hp, lfp)
    ref = O.predict(geom, hp0, lfp, w["x"], w["t"], type="csd")
    sc = np.max(np.abs(ref["csd"]))
    print(nt, "loglik: pipelined vs not %.1e, vs oracle %.1e; predict: pipelined vs not %.1e, vs oracle %.1e" % (
        abs(vals[True][0] - vals[False][0]) / abs(ll_ref), abs(vals[True][0] - ll_ref) / abs(ll_ref),
        np.max(np.abs(vals[True][1] - vals[False][1])) / sc, np.max(np.abs(vals[True][1] - ref["csd"])) / sc))
    assert abs(vals[True][0] - vals[False][0]) <= 1e-11 * abs(ll_ref)
    assert abs(vals[True][0] - ll_ref) <= 1e-9 * abs(ll_ref)
    assert np.max(np.abs(vals[True][1] - vals[False][1])) <= 1e-9 * sc
    assert np.max(np.abs(vals[True][1] - ref["csd"])) <= 1e-8 * sc


@pytest.mark.parametrize("form", ["fenced", "paired", "resident", "class_api"])
def test_a_gate_that_gives_up_is_a_scheduling_miss_and_the_call_is_evaluated_again(form):
    """ADVICE r5 (wy.hip gate): when the tridiagonalisation is not running beside the launch that waits for it (kernels serialised by
    a profiler, an oversubscribed card) the gate's bounded wait runs out.  That is not a numerical failure: the stage launch leaves
    the unfinished reflectors alone, and the collecting call switches the pipeline off for the context (latched), counts the miss
    and evaluates again unpipelined -- the caller sees the unpipelined result, no LinAlgError.  Forced here with a patience of zero
    ticks (gpcsd_q_pipeline_stats), through every call form that can carry a pipelined chain."""
    from gpcsd_amd import _hip
    R = 16
    w, m, lfp = _step_model(R)
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    ctx.pair_share_s(False)
    z = w["x"]
    hp, k1 = m._hparams(m.JITTER)
    hp0, k0 = m._hparams(0.0)
    ctx.q_pipeline(False)                                     # the answer: the unpipelined form
    want_ll = ctx.loglik_parts(hp)
    ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    ctx.synchronize()
    want_pr = ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], R)).copy()
    ctx.q_pipeline(True)
    try:
        ctx.q_pipeline_stats(gate_ticks=0)                    # every gate gives up at once
        on0, miss0 = ctx.q_pipeline_stats()
        assert on0 and miss0 == 0
        if form == "fenced":
            got_ll = ctx.loglik_parts(hp)
            ctx.q_pipeline(True)                              # (latched off by the miss: on again for the second call form)
            ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            ctx.synchronize()                                 # the deferred status of the queued prediction is collected here
        elif form == "paired":
            ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            got_ll = ctx.loglik_parts_wait()
            ctx.synchronize()
        elif form == "resident":
            ctx.loglik_parts_async(hp)
            got_ll = ctx.loglik_parts_wait()
            ctx.q_pipeline(True)
            ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            # (no synchronize: gpcsd_fetch itself collects the queued prediction's status -- and evaluates it again -- before it copies)
        else:
            got_ll = None
            ll_class = float(m.loglik())
            ctx.q_pipeline(True)
            m.predict(z, w["t"], type="csd")
        got_pr = ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], R)).copy()     # (collects a queued prediction's status first)
        on1, miss1 = ctx.q_pipeline_stats()
        assert not on1 and miss1 >= 1, (on1, miss1)           # latched off, counted
        if form == "class_api":
            assert ll_class == -0.5 * R * want_ll[0] - 0.5 * want_ll[1]
            assert np.array_equal(np.asarray(m.csd_pred), want_pr)
        else:
            assert tuple(got_ll) == tuple(want_ll)
        assert np.array_equal(got_pr, want_pr)                # bit for bit the unpipelined evaluation
        # ... and the context works on (pipeline off) with a normal patience
        ctx.q_pipeline_stats(gate_ticks=20000000)
        assert tuple(ctx.loglik_parts(hp)) == tuple(want_ll)
    finally:
        ctx.q_pipeline_stats(gate_ticks=20000000)
        ctx.q_pipeline(True)
