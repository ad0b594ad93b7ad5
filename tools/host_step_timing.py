"""Host-side split of bench.py's paired step: marshalling the two hyper-parameter sets, queueing the paired call, waiting for the
log-likelihood (medians over N steps).   python tools/host_step_timing.py [cfg3|cfg2] [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpcsd_amd import _hip

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w = bench.workload(name)
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
ctx.decomposition_cache(False)
z = w["x"]
T = np.zeros((N, 4))
for k in range(N + 100):
    t0 = time.perf_counter()
    hp, keep = m._hparams(m.JITTER)
    hp0, keep0 = m._hparams(0.0)
    t1 = time.perf_counter()
    ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    t2 = time.perf_counter()
    ctx.loglik_parts_wait()
    t3 = time.perf_counter()
    if k >= 100:
        T[k - 100] = (t1 - t0, t2 - t1, t3 - t2, t3 - t0)
ctx.synchronize()
med = np.median(T, axis=0) * 1e6
print("%s: hparams x2 %.1f us | queue paired call %.1f us | wait %.1f us | step %.1f us" % (name, *med))
