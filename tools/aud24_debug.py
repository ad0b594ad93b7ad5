"""Which starts of the aud24 fit fail in the eigensolver, alone and in a batch (round 5 debugging aid)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

w = bench.workload("aud24")
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, 8, seed=11)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
np.random.seed(5)
starts = [m._sample_start(False) for _ in range(6)]
ng = 1 + 1 + 4 + 24
hps, keep = [], []
for k, tp in enumerate(starts):
    m._set_from_tparams(tp, False)
    h, kk = m._hparams(m.JITTER)
    hps.append(h)
    keep.append(kk)
    try:
        r = ctx.loglik_grad(h, ng)
        print("start", k, "alone ok", r[0], r[1])
    except Exception as e:
        print("start", k, "alone FAILED", repr(e))
        print("   nat:", np.exp(tp[:6]) * np.array([100, 100, 1, 1, 1, 1]), "sig2n min/max", np.exp(tp[6:]).min(), np.exp(tp[6:]).max())
    try:
        print("   loglik_parts", ctx.loglik_parts(h))
    except Exception as e:
        print("   loglik_parts FAILED", repr(e))
sumlog, quad, grad, st = ctx.loglik_grad_batch(hps, ng)
print("batch status", st, "sumlog", sumlog)
opts = {"maxiter": 5, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}
m.fit(n_restarts=6, options=opts, starts=starts)
print("fit batches", m.fit_batches_, "nll", m.fit_nll_values_)
