"""Host-side split of the step `value` is timed on (announced, one spatial decomposition): marshalling, queueing the paired call,
announcing the next one, waiting for the log-likelihood -- medians over N steps; with GPCSD_TIMELINE=1 the device-side phase marks
of the last steps follow on stderr when the context closes.   python tools/step_phases.py [cfg3|cfg2] [N]"""
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
ctx.pair_share_s(True)
z = w.get("z", w["x"])
WARM = 700
T = np.zeros((N, 5))
for k in range(N + WARM):
    t0 = time.perf_counter()
    hp, keep = m._hparams(m.JITTER)
    hp0, keep0 = m._hparams(0.0)
    t1 = time.perf_counter()
    ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    t2 = time.perf_counter()
    ctx.prefetch_pair(hp, hp0, z, w["t"])
    t3 = time.perf_counter()
    ctx.loglik_parts_wait()
    t4 = time.perf_counter()
    if k >= WARM:
        T[k - WARM] = (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)
ctx.synchronize()
med = np.median(T, axis=0) * 1e6
print("%s: hparams x2 %.1f us | queue paired call %.1f us | announce next %.1f us | wait %.1f us | step %.1f us" % (name, *med))
ctx.close()
