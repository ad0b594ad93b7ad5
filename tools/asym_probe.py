"""cfg3 geometry, predict at four off-grid sites (no mirror symmetry): ms per paired step and per fenced predict."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from gpcsd_amd import _hip
w = bench.workload("cfg3"); m = bench.build_model(w, np.zeros((384, 500, 1)))
lfp = bench.synth_data(w, m, 50, seed=1000); m.update_lfp(lfp, w["t"])
ctx = m._sync_device(); ctx.decomposition_cache(False)
z = np.stack([24.0 * np.ones(4), np.array([2260.0, 2450.0, 2650.0, 2785.0])]).T
h1, k1 = m._hparams(m.JITTER); h0, k0 = m._hparams(0.0)
def step():
    ctx.loglik_predict_async(h1, h0, z, w["t"], _hip.PRED_CSD, want_lists=True); return ctx.loglik_parts_wait()
for _ in range(100): step()
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
ctx.synchronize(); print("paired step, 4 off-grid sites: %.4f ms" % (1e3 * (time.perf_counter() - t0) / 200))
t0 = time.perf_counter()
for _ in range(50):
    ctx.predict_resident(h0, z, w["t"], _hip.PRED_CSD, want_lists=True); ctx.synchronize()
print("fenced predict: %.4f ms" % (1e3 * (time.perf_counter() - t0) / 50))
