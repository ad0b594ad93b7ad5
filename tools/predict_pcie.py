"""Class-API predict() with host outputs at cfg3: chunked copy-out on / off, decomposition cache off / on (round 5)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

w = bench.workload("cfg3")
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
R = w["trials_per_gpu"]
lfp = bench.synth_data(w, m, R, seed=1000)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
out_bytes = 3 * w["nx"] * w["nt"] * R * 8
for cache in (False, True):
    ctx.decomposition_cache(cache)
    for rep in range(2):
        for on in (False, True):
            ctx.predict_chunked_copy(on)
            for _ in range(3):
                m.predict(w["x"], w["t"], type="csd")
            t0 = time.perf_counter()
            n = 6
            for _ in range(n):
                m.predict(w["x"], w["t"], type="csd")
            dt = (time.perf_counter() - t0) / n
            print("cache %d chunked %d: %.3f ms per predict = %.0f trials/s, %.1f GB/s to the host" % (cache, on, 1e3 * dt, R / dt, out_bytes / dt / 1e9))
