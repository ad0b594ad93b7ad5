import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, bench
from gpcsd_amd import _hip
w = bench.workload("cfg3"); m = bench.build_model(w, np.zeros((384, 500, 1)))
lfp = bench.synth_data(w, m, 50, seed=1000); m.update_lfp(lfp, w["t"])
ctx = m._sync_device(); ctx.decomposition_cache(False)
h0, k0 = m._hparams(0.0)
for _ in range(3):
    ctx.predict_resident(h0, w["x"], w["t"], _hip.PRED_CSD, want_lists=True); ctx.synchronize()
