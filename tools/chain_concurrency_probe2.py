"""k contexts (one host thread each), each evaluating loglik at the cfg3 geometry with ONE resident trial: nothing but the two
eigen-chains and a few tiny GEMMs per call, no host<->device copies besides the 528-byte result."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.workload("cfg3")
reps = 60
for k in (1, 2, 3, 4, 6):
    ms = []
    for i in range(k):
        m = bench.build_model(w, np.random.RandomState(i).standard_normal((w["nx"], w["nt"], 1)))
        ctx = m._sync_device()
        ctx.decomposition_cache(False)
        hp, keep = m._hparams(m.JITTER)
        for _ in range(5):
            ctx.loglik_parts(hp)
        ms.append((m, ctx, hp, keep))

    def run(ctx, hp):
        for _ in range(reps):
            ctx.loglik_parts(hp)
    ths = [threading.Thread(target=run, args=(c, h)) for _, c, h, _ in ms]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    print("%d contexts: %.3f ms per loglik per context, %.0f calls/s in total" % (k, 1e3 * dt / reps, k * reps / dt), flush=True)
    del ms
