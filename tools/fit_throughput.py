"""Restart throughput of fit() on one GPU: sequential vs concurrent restarts (cfg5-like: GPCSD1D 24 x 500 x 200 trials)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.workload("cfg2")
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, 200, seed=1)
m.update_lfp(lfp, w["t"])
np.random.seed(0)
starts = [m._sample_start(False) for _ in range(8)]
opts = {"maxiter": 15, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}
best = {}
for bits in (64, 32):                    # 32 = the "fp32 kernel build + fp64 factor" variant of BASELINE cfg5
    for workers in (1, 2, 4, 8):
        mm = bench.build_model(w, lfp)
        mm.gram_precision = bits
        t0 = time.perf_counter()
        mm.fit(n_restarts=8, options=opts, starts=starts, workers=workers)
        dt = time.perf_counter() - t0
        best[bits] = float(np.nanmin(mm.fit_nll_values_))
        print("gram fp%d workers=%d: 8 restarts x <=15 iterations in %.2f s (%.2f restarts/s), best nll %.6f"
              % (bits, workers, dt, 8 / dt, best[bits]), flush=True)
print("best nll, fp32 Gram build vs fp64: relative deviation %.2e" % (abs(best[32] - best[64]) / abs(best[64])))
