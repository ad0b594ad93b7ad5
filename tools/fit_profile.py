"""cProfile of a truncated cfg5 fit (32 restarts x 15 iterations, lock-step): where the host time between the batched device
evaluations goes.   python tools/fit_profile.py [threads|auto]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402


def main():
    driver = sys.argv[1] if len(sys.argv) > 1 else "auto"
    w = bench.workload("cfg5")
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000)
    m.update_lfp(lfp, w["t"])
    starts = []
    for k in range(32):
        np.random.seed(k)
        starts.append(m._sample_start(False))
    opts = {"maxiter": 15, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}
    m.fit_driver = driver
    m.fit(n_restarts=32, options=opts, starts=starts, batch=32)
    t0 = time.perf_counter()
    m.fit(n_restarts=32, options=opts, starts=starts, batch=32)
    print("fit %.2f ms, batches %s" % (1e3 * (time.perf_counter() - t0), m.fit_batches_))
    pr = cProfile.Profile()
    pr.enable()
    m.fit(n_restarts=32, options=opts, starts=starts, batch=32)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
