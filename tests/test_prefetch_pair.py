"""gpcsd_prefetch_pair (round 5): the decomposition chains of the NEXT paired call (utility_functions.py:58-59 for Kt and Ks of
gpcsd2d.py:136-151 and :289-334) queued ahead of it.  Same launches on the same buffers: the results are the bits of the
unannounced calls; an announcement that does not come true is dropped."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _step_model(R, name):
    import bench
    w = bench.workload(name)
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, R, seed=3)
    m.update_lfp(lfp, w["t"])
    return w, m, lfp


@pytest.mark.parametrize("name,R", [("cfg3", 16), ("cfg3", 12), ("cfg2", 24), ("npx69", 20)])
def test_announced_pairs_give_the_bits_of_unannounced_ones(name, R):
    """A sequence of paired calls with changing hyper-parameters, each announcing its successor, against the same sequence without
    announcements: every log-likelihood and every prediction bit for bit; all announcements but the last are taken over."""
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(R, name)
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    z = np.ascontiguousarray(w.get("z", w["x"]))
    tc = m.temporal_cov_list[0]

    def hps(step):
        tc.params["ell"]["value"] = 20.0 + 0.5 * (step % 4)
        return m._hparams(m.JITTER), m._hparams(0.0)

    def run(announce, steps=7):
        out = []
        cur = hps(0)
        for step in range(steps):
            (hp, k1), (hp0, k0) = cur
            ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            nxt = hps(step + 1)
            if announce:
                assert ctx.prefetch_pair(nxt[0][0], nxt[1][0], z, w["t"])
            sl, qd = ctx.loglik_parts_wait()
            if step % 3 == 2:                 # (now and then the prediction is looked at, with the next chains already running)
                ctx.synchronize()
                out.append((sl, qd, ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], R)).copy()))
            else:
                out.append((sl, qd, None))
            cur = nxt
        ctx.synchronize()
        return out

    q0, t0 = ctx.prefetch_stats()
    plain = run(False)
    assert ctx.prefetch_stats() == (q0, t0)
    ann = run(True)
    q1, t1 = ctx.prefetch_stats()
    assert q1 - q0 == 7 and t1 - t0 == 6                      # the first call had no announcement, the last announcement no call
    for a, b in zip(plain, ann):
        assert a[0] == b[0] and a[1] == b[1]
        assert (a[2] is None) == (b[2] is None) and (a[2] is None or np.array_equal(a[2], b[2]))
    assert plain[0][0:2] != plain[1][0:2]


def test_an_announcement_that_does_not_come_true_is_dropped():
    """Announce one set of hyper-parameters, call with another (and: announce, then make a plain loglik call in between): the call
    queues its own chains and gives the bits of an unannounced call."""
    from gpcsd_amd import _hip
    w, m, lfp = _step_model(16, "cfg3")
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    z = np.ascontiguousarray(w["x"])
    tc = m.temporal_cov_list[0]
    (hpA, kA), (hpA0, kA0) = m._hparams(m.JITTER), m._hparams(0.0)
    tc.params["ell"]["value"] = 23.0
    (hpB, kB), (hpB0, kB0) = m._hparams(m.JITTER), m._hparams(0.0)

    def pair(hp, hp0):
        ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        sl, qd = ctx.loglik_parts_wait()
        ctx.synchronize()
        return sl, qd, ctx.fetch("pred_out_csd", (z.shape[0], w["nt"], 16)).copy()

    refB = pair(hpB, hpB0)
    q0, t0 = ctx.prefetch_stats()
    assert ctx.prefetch_pair(hpA, hpA0, z, w["t"])
    gotB = pair(hpB, hpB0)                                     # announced A, called B
    assert ctx.prefetch_stats() == (q0 + 1, t0)
    assert gotB[0] == refB[0] and gotB[1] == refB[1] and np.array_equal(gotB[2], refB[2])
    assert ctx.prefetch_pair(hpB, hpB0, z, w["t"])
    sl, qd = ctx.loglik_parts(hpA)                             # a plain call in between overtakes the announcement
    gotB = pair(hpB, hpB0)
    assert ctx.prefetch_stats() == (q0 + 2, t0)
    assert gotB[0] == refB[0] and gotB[1] == refB[1] and np.array_equal(gotB[2], refB[2])
