"""Throughput of k concurrent loglik+gradient evaluation streams on one GPU (one context per thread)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = bench.workload(name)
m0 = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m0, w["trials_per_gpu"], seed=1)
for k in (1, 2, 4, 8):
    models = []
    for i in range(k):
        m = bench.build_model(w, lfp)
        for tc, tc0 in zip(m.temporal_cov_list, m0.temporal_cov_list):
            tc.params["sigma2"]["value"] = tc0.params["sigma2"]["value"]
        m._loglik_and_grad_natural(); m._loglik_and_grad_natural(); m._loglik_and_grad_natural()
        models.append(m)
    n = 30
    def run(m):
        for _ in range(n):
            m._loglik_and_grad_natural()
    ths = [threading.Thread(target=run, args=(m,)) for m in models]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("%s: %d concurrent streams: %.1f evaluations/s (%.3f ms each per stream)" % (name, k, k * n / dt, dt / n * 1e3), flush=True)
    del models
